// REHEARSAL STAND-IN (oracle/pin_dryrun/README.md): voxblox/src/utils/camera_model.cc has nothing to add to the header-only stand-ins.

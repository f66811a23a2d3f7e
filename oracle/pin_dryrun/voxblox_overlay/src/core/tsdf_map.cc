// REHEARSAL STAND-IN (oracle/pin_dryrun/README.md): voxblox/src/core/tsdf_map.cc has nothing to add to the header-only stand-ins.

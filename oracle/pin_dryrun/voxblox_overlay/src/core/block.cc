// REHEARSAL STAND-IN (oracle/pin_dryrun/README.md): voxblox/src/core/block.cc has nothing to add to the header-only stand-ins.

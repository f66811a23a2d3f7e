// REHEARSAL STAND-IN (oracle/pin_dryrun/README.md) for voxblox/src/integrator/integrator_utils.cc (RayCaster): here the
// oracle itself -- oracle/tsdf_oracle.c builds as C++ -- so that the fake voxblox links without liboracle.
extern "C" {
#include "tsdf_oracle.c"
}

// Stand-in for cblox/core/tsdf_submap.h: included by voxgraph's headers, nothing of it is used on the paths compiled
// here.  TEST INFRASTRUCTURE.
#ifndef ORACLE_REF_SHIMS_CBLOX_CORE_TSDF_SUBMAP_H_
#define ORACLE_REF_SHIMS_CBLOX_CORE_TSDF_SUBMAP_H_
#include "cblox/core/common.h"
#endif

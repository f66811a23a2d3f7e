// Stand-in for cblox/core/common.h ([recalled]): the two names voxgraph's backend uses.  TEST INFRASTRUCTURE.
#ifndef ORACLE_REF_SHIMS_CBLOX_CORE_COMMON_H_
#define ORACLE_REF_SHIMS_CBLOX_CORE_COMMON_H_
#include "voxblox/core/common.h"
namespace cblox {
typedef unsigned int SubmapID;
using voxblox::Transformation;
}  // namespace cblox
#endif

// TF publishing is visualisation only: no-op stand-in.  TEST INFRASTRUCTURE.
#ifndef ORACLE_REF_SHIMS_VOXGRAPH_TOOLS_TF_HELPER_H_
#define ORACLE_REF_SHIMS_VOXGRAPH_TOOLS_TF_HELPER_H_
#include <string>

#include "voxblox/core/common.h"
namespace voxgraph {
class TfHelper {
 public:
  static void publishTransform(const voxblox::Transformation&, const std::string&, const std::string&,
                               bool = false) {}
};
}  // namespace voxgraph
#endif

// RViz residual / gradient clouds are visualisation only: no-op stand-in.  TEST INFRASTRUCTURE.
#ifndef ORACLE_REF_SHIMS_VOXGRAPH_TOOLS_VISUALIZATION_COST_FUNCTION_VISUALS_H_
#define ORACLE_REF_SHIMS_VOXGRAPH_TOOLS_VISUALIZATION_COST_FUNCTION_VISUALS_H_
#include "voxblox/core/common.h"
namespace voxgraph {
class CostFunctionVisuals {
 public:
  void addResidual(const voxblox::Point&, const double&) {}
  void addJacobian(const voxblox::Point&, const voxblox::Point&) {}
  void scaleAndPublish(const double) {}
  void reset() {}
};
}  // namespace voxgraph
#endif

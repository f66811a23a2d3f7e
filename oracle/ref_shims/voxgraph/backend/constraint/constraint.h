// The reference's constraint.h drags in the node collection and ceres::Problem; the cost functor
// compiled here only needs Constraint::InformationMatrix.  TEST INFRASTRUCTURE (oracle/ref_shims).
#ifndef ORACLE_REF_SHIMS_VOXGRAPH_BACKEND_CONSTRAINT_CONSTRAINT_H_
#define ORACLE_REF_SHIMS_VOXGRAPH_BACKEND_CONSTRAINT_CONSTRAINT_H_
#include <Eigen/Core>

#include "voxblox/core/common.h"
namespace voxgraph {
class Constraint {
 public:
  typedef Eigen::Matrix<double, 4, 4> InformationMatrix;
};
}  // namespace voxgraph
#endif

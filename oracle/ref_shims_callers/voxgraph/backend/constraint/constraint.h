// oracle/ref_shims/voxgraph/backend/constraint/constraint.h is an InformationMatrix-only stand-in (the cost-function
// build must not drag in the node collection and ceres::Problem).  The CALLERS build needs the reference's real header:
// this directory comes first on its include path and forwards to it (the path is given on the command line, so no
// reference text is copied).  TEST INFRASTRUCTURE.
#ifndef VGX_REFERENCE_CONSTRAINT_H
#error "build through oracle/Makefile (_ref/callers_check_*): VGX_REFERENCE_CONSTRAINT_H names the reference's constraint.h"
#endif
#include VGX_REFERENCE_CONSTRAINT_H

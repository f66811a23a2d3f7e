// ceres::CostFunction interface only (Ceres is absent): shared with the adapter tests.
#include "../../../tests/stubs/ceres/ceres.h"

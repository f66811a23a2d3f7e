// Minimal stand-in for <ceres/ceres.h> (Ceres is not installed in this image): just the
// ceres::CostFunction surface the adapter in voxgraph_amd/cpp/ touches.  TEST ONLY.
#ifndef TESTS_STUBS_CERES_CERES_H_
#define TESTS_STUBS_CERES_CERES_H_
#include <cmath>
#include <cstdint>
#include <vector>
namespace ceres {
class CostFunction {
 public:
  CostFunction() : num_residuals_(0) {}
  virtual ~CostFunction() {}
  virtual bool Evaluate(double const* const* parameters, double* residuals,
                        double** jacobians) const = 0;
  const std::vector<int32_t>& parameter_block_sizes() const { return parameter_block_sizes_; }
  int num_residuals() const { return num_residuals_; }

 protected:
  std::vector<int32_t>* mutable_parameter_block_sizes() { return &parameter_block_sizes_; }
  void set_num_residuals(int n) { num_residuals_ = n; }

 private:
  std::vector<int32_t> parameter_block_sizes_;
  int num_residuals_;
};
template <int kNumResiduals, int N0, int N1>
class SizedCostFunction : public CostFunction {
 public:
  SizedCostFunction() {
    set_num_residuals(kNumResiduals);
    mutable_parameter_block_sizes()->push_back(N0);
    mutable_parameter_block_sizes()->push_back(N1);
  }
};
// the scalar overloads ceres::cos / sin / floor resolve to for T = double
inline double cos(double x) { return std::cos(x); }
inline double sin(double x) { return std::sin(x); }
inline double floor(double x) { return std::floor(x); }
// AutoDiffCostFunction: residuals through the functor with T = double; no Jets here, so
// Jacobians cannot be produced (Evaluate returns false if they are requested)
template <typename Functor, int kNumResiduals, int N0, int N1>
class AutoDiffCostFunction : public SizedCostFunction<kNumResiduals, N0, N1> {
 public:
  explicit AutoDiffCostFunction(Functor* functor) : functor_(functor) {}
  ~AutoDiffCostFunction() override { delete functor_; }
  bool Evaluate(double const* const* parameters, double* residuals, double** jacobians) const override {
    if (jacobians) return false;
    return (*functor_)(parameters[0], parameters[1], residuals);
  }

 private:
  Functor* functor_;
};
class EvaluationCallback {
 public:
  virtual ~EvaluationCallback() {}
  virtual void PrepareForEvaluation(bool evaluate_jacobians, bool new_evaluation_point) = 0;
};
}  // namespace ceres
#endif

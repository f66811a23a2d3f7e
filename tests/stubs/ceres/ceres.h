// Minimal stand-in for <ceres/ceres.h> (Ceres is not installed in this image): the
// ceres::CostFunction / EvaluationCallback surface the adapters in voxgraph_amd/cpp/ touch, and a
// small Problem / Solve with Ceres' calling conventions for end-to-end checks.  TEST ONLY.
// Round 5: enough more of the interface (Problem::Options, local parameterizations, residual-block queries,
// Covariance, NumericDiffCostFunction, Solver::Summary reports) for the reference's OWN callers -- pose_graph.cpp,
// registration_constraint.cpp, node*.cpp, submap_registration_helper.cpp -- to compile against it
// (oracle/ref_driver/callers_check.cpp).
// Round 6: Jet + a real AutoDiffCostFunction (the reference's relative_pose_cost_function_inl.h, i.e. its odometry /
// loop-closure / absolute-pose constraints, now compile and differentiate against it), and a solver loop that
// evaluates trial steps cost-only (`jacobians == nullptr`) as Ceres' trust-region minimizer does; ceres::Covariance
// computes (dense (J^T J)^-1 over the free blocks) instead of reporting failure: PoseGraph::getEdgeCovarianceMap runs.
#ifndef TESTS_STUBS_CERES_CERES_H_
#define TESTS_STUBS_CERES_CERES_H_
#include <cmath>
#include <cstdint>
#include <deque>
#include <ostream>
#include <string>
#include <type_traits>
#include <utility>
#include <vector>
namespace ceres {
enum Ownership { DO_NOT_TAKE_OWNERSHIP, TAKE_OWNERSHIP };
enum LinearSolverType { DENSE_NORMAL_CHOLESKY, DENSE_QR, SPARSE_NORMAL_CHOLESKY, DENSE_SCHUR, SPARSE_SCHUR, ITERATIVE_SCHUR, CGNR };
enum NumericDiffMethodType { CENTRAL, FORWARD, RIDDERS };
enum { DYNAMIC = -1 };
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

// ---- ceres::Jet: first-order forward-mode dual numbers (value a + N partial derivatives v), what
// AutoDiffCostFunction feeds a functor.  Only the operations voxgraph's functors use
// (relative_pose_cost_function_inl.h, normalize_angle.h, angle_local_parameterization.h): + - * /, comparisons on the
// value, cos, sin, floor (piecewise constant: zero derivative, as in Ceres), abs, streaming.
template <typename T, int N>
struct Jet {
  T a;
  T v[N];
  Jet() : a() {
    for (int k = 0; k < N; ++k) v[k] = T();
  }
  template <typename U, typename = typename std::enable_if<std::is_arithmetic<U>::value>::type>
  Jet(const U& value) : a(static_cast<T>(value)) {  // NOLINT (implicit: T(0), static_cast<T>(double), 2.0 * x)
    for (int k = 0; k < N; ++k) v[k] = T();
  }
  Jet(const T& value, int k_one) : a(value) {
    for (int k = 0; k < N; ++k) v[k] = k == k_one ? T(1) : T();
  }
  Jet& operator+=(const Jet& y) { return *this = *this + y; }
  Jet& operator-=(const Jet& y) { return *this = *this - y; }
  Jet& operator*=(const Jet& y) { return *this = *this * y; }
  Jet& operator/=(const Jet& y) { return *this = *this / y; }
};
template <typename T, int N>
Jet<T, N> operator+(const Jet<T, N>& x, const Jet<T, N>& y) {
  Jet<T, N> r;
  r.a = x.a + y.a;
  for (int k = 0; k < N; ++k) r.v[k] = x.v[k] + y.v[k];
  return r;
}
template <typename T, int N>
Jet<T, N> operator-(const Jet<T, N>& x, const Jet<T, N>& y) {
  Jet<T, N> r;
  r.a = x.a - y.a;
  for (int k = 0; k < N; ++k) r.v[k] = x.v[k] - y.v[k];
  return r;
}
template <typename T, int N>
Jet<T, N> operator-(const Jet<T, N>& x) {
  Jet<T, N> r;
  r.a = -x.a;
  for (int k = 0; k < N; ++k) r.v[k] = -x.v[k];
  return r;
}
template <typename T, int N>
Jet<T, N> operator*(const Jet<T, N>& x, const Jet<T, N>& y) {
  Jet<T, N> r;
  r.a = x.a * y.a;
  for (int k = 0; k < N; ++k) r.v[k] = x.a * y.v[k] + x.v[k] * y.a;
  return r;
}
template <typename T, int N>
Jet<T, N> operator/(const Jet<T, N>& x, const Jet<T, N>& y) {
  Jet<T, N> r;
  const T inv = T(1) / y.a;
  r.a = x.a * inv;
  for (int k = 0; k < N; ++k) r.v[k] = (x.v[k] - r.a * y.v[k]) * inv;
  return r;
}
#define VGX_JET_MIXED(op)                                                                                     \
  template <typename T, int N, typename U, typename = typename std::enable_if<std::is_arithmetic<U>::value>::type> \
  Jet<T, N> operator op(const Jet<T, N>& x, const U& s) { return x op Jet<T, N>(s); }                        \
  template <typename T, int N, typename U, typename = typename std::enable_if<std::is_arithmetic<U>::value>::type> \
  Jet<T, N> operator op(const U& s, const Jet<T, N>& y) { return Jet<T, N>(s) op y; }
VGX_JET_MIXED(+)
VGX_JET_MIXED(-)
VGX_JET_MIXED(*)
VGX_JET_MIXED(/)
#undef VGX_JET_MIXED
#define VGX_JET_COMPARE(op)                                                                       \
  template <typename T, int N>                                                                    \
  bool operator op(const Jet<T, N>& x, const Jet<T, N>& y) { return x.a op y.a; }                 \
  template <typename T, int N, typename U, typename = typename std::enable_if<std::is_arithmetic<U>::value>::type> \
  bool operator op(const Jet<T, N>& x, const U& s) { return x.a op static_cast<T>(s); }           \
  template <typename T, int N, typename U, typename = typename std::enable_if<std::is_arithmetic<U>::value>::type> \
  bool operator op(const U& s, const Jet<T, N>& y) { return static_cast<T>(s) op y.a; }
VGX_JET_COMPARE(<)
VGX_JET_COMPARE(<=)
VGX_JET_COMPARE(>)
VGX_JET_COMPARE(>=)
VGX_JET_COMPARE(==)
VGX_JET_COMPARE(!=)
#undef VGX_JET_COMPARE
template <typename T, int N>
Jet<T, N> cos(const Jet<T, N>& x) {
  Jet<T, N> r;
  r.a = std::cos(x.a);
  const T d = -std::sin(x.a);
  for (int k = 0; k < N; ++k) r.v[k] = d * x.v[k];
  return r;
}
template <typename T, int N>
Jet<T, N> sin(const Jet<T, N>& x) {
  Jet<T, N> r;
  r.a = std::sin(x.a);
  const T d = std::cos(x.a);
  for (int k = 0; k < N; ++k) r.v[k] = d * x.v[k];
  return r;
}
template <typename T, int N>
Jet<T, N> floor(const Jet<T, N>& x) {
  return Jet<T, N>(std::floor(x.a));
}
template <typename T, int N>
Jet<T, N> abs(const Jet<T, N>& x) {
  return x.a < T(0) ? -x : x;
}
template <typename T, int N>
std::ostream& operator<<(std::ostream& os, const Jet<T, N>& x) {
  os << "[" << x.a << " ;";
  for (int k = 0; k < N; ++k) os << " " << x.v[k];
  return os << "]";
}

// AutoDiffCostFunction<Functor, kNumResiduals, N0, N1>: the functor with T = double when no Jacobian is asked for, with
// T = Jet<double, N0 + N1> otherwise (parameter k of block b carries the unit derivative N0 * b + k): residual i's
// partial derivatives are row i of the two Jacobian blocks, row-major -- Ceres' contract.
template <typename Functor, int kNumResiduals, int N0, int N1>
class AutoDiffCostFunction : public SizedCostFunction<kNumResiduals, N0, N1> {
 public:
  explicit AutoDiffCostFunction(Functor* functor) : functor_(functor) {}
  ~AutoDiffCostFunction() override { delete functor_; }
  bool Evaluate(double const* const* parameters, double* residuals, double** jacobians) const override {
    if (!jacobians) return (*functor_)(parameters[0], parameters[1], residuals);
    typedef Jet<double, N0 + N1> J;
    J x0[N0], x1[N1], y[kNumResiduals];
    for (int k = 0; k < N0; ++k) x0[k] = J(parameters[0][k], k);
    for (int k = 0; k < N1; ++k) x1[k] = J(parameters[1][k], N0 + k);
    if (!(*functor_)(x0, x1, y)) return false;
    for (int i = 0; i < kNumResiduals; ++i) {
      residuals[i] = y[i].a;
      if (jacobians[0])
        for (int k = 0; k < N0; ++k) jacobians[0][i * N0 + k] = y[i].v[k];
      if (jacobians[1])
        for (int k = 0; k < N1; ++k) jacobians[1][i * N1 + k] = y[i].v[N0 + k];
    }
    return true;
  }

 private:
  Functor* functor_;
};
class EvaluationCallback {
 public:
  virtual ~EvaluationCallback() {}
  virtual void PrepareForEvaluation(bool evaluate_jacobians, bool new_evaluation_point) = 0;
};

// ---- a small Problem / Solver with Ceres' calling conventions (TEST ONLY) ---------------------------
// Enough of ceres::Problem and ceres::Solve to drive the GPU cost functions the way voxgraph does
// (pose_graph.cpp:74-106): residual blocks over 4-parameter pose blocks, constant blocks
// (SetParameterBlockConstant, pose_graph_interface.cpp:30-32), Solver::Options::evaluation_callback,
// and a dense Levenberg-Marquardt loop with Ceres' acceptance / radius rules and default tolerances.
// Jacobians of constant blocks are requested as nullptr, as Ceres does.
class LossFunction;
struct ResidualBlock;
typedef ResidualBlock* ResidualBlockId;   // (a pointer type, as in Ceres: voxgraph initialises its ids with nullptr)

// ---- local parameterizations: Plus() only (the stub solver's step: x <- Plus(x, delta)) -------------------------------
// voxgraph's are Identity(3) x AutoDiff<AngleLocalParameterization, 1, 1> (node_collection.cpp:7-11): the Jacobian of
// Plus at delta = 0 is the identity for both, so the solver needs no ComputeJacobian.
class LocalParameterization {
 public:
  virtual ~LocalParameterization() {}
  virtual bool Plus(const double* x, const double* delta, double* x_plus_delta) const = 0;
  virtual int GlobalSize() const = 0;
  virtual int LocalSize() const = 0;
};
class IdentityParameterization : public LocalParameterization {
 public:
  explicit IdentityParameterization(int size) : size_(size) {}
  bool Plus(const double* x, const double* delta, double* out) const override {
    for (int i = 0; i < size_; ++i) out[i] = x[i] + delta[i];
    return true;
  }
  int GlobalSize() const override { return size_; }
  int LocalSize() const override { return size_; }

 private:
  int size_;
};
template <typename Functor, int kGlobalSize, int kLocalSize>
class AutoDiffLocalParameterization : public LocalParameterization {
 public:
  AutoDiffLocalParameterization() : functor_(new Functor()) {}
  explicit AutoDiffLocalParameterization(Functor* functor) : functor_(functor) {}
  ~AutoDiffLocalParameterization() override { delete functor_; }
  bool Plus(const double* x, const double* delta, double* out) const override { return (*functor_)(x, delta, out); }
  int GlobalSize() const override { return kGlobalSize; }
  int LocalSize() const override { return kLocalSize; }

 private:
  Functor* functor_;
};
class ProductParameterization : public LocalParameterization {
 public:
  ProductParameterization(LocalParameterization* a, LocalParameterization* b) : a_(a), b_(b) {}
  ~ProductParameterization() override {
    delete a_;
    delete b_;
  }
  bool Plus(const double* x, const double* delta, double* out) const override {
    return a_->Plus(x, delta, out) && b_->Plus(x + a_->GlobalSize(), delta + a_->LocalSize(), out + a_->GlobalSize());
  }
  int GlobalSize() const override { return a_->GlobalSize() + b_->GlobalSize(); }
  int LocalSize() const override { return a_->LocalSize() + b_->LocalSize(); }

 private:
  LocalParameterization *a_, *b_;
};

struct CRSMatrix {};

struct ResidualBlock {
  CostFunction* cost;
  int param[2];
};

class Problem {
 public:
  struct Options {
    Ownership cost_function_ownership = TAKE_OWNERSHIP;
    Ownership loss_function_ownership = TAKE_OWNERSHIP;
    Ownership local_parameterization_ownership = TAKE_OWNERSHIP;
  };
  struct EvaluateOptions {};
  Problem() {}
  explicit Problem(const Options& options) : options_(options) {}
  Problem(const Problem&) = delete;
  Problem& operator=(const Problem&) = delete;
  ~Problem() {
    if (options_.cost_function_ownership == TAKE_OWNERSHIP)
      for (Block& b : blocks_) delete b.cost;  // Ceres' default
    // (local parameterizations: voxgraph keeps ownership, pose_graph.cpp:76-77; the stub never deletes them)
  }
  void AddParameterBlock(double* values, int size) { Index(values, size); }
  void SetParameterBlockConstant(double* values) { constant_[Index(values, 4)] = true; }
  void SetParameterization(double* values, LocalParameterization* lp) { parameterization_[Index(values, lp->GlobalSize())] = lp; }
  ResidualBlockId AddResidualBlock(CostFunction* cost, LossFunction* /*loss*/, double* x0, double* x1) {
    Block b;
    b.cost = cost;
    b.param[0] = Index(x0, cost->parameter_block_sizes()[0]);
    b.param[1] = Index(x1, cost->parameter_block_sizes()[1]);
    blocks_.push_back(b);
    return &blocks_.back();
  }
  int NumResidualBlocks() const { return static_cast<int>(blocks_.size()); }
  void GetResidualBlocks(std::vector<ResidualBlockId>* ids) const {
    ids->clear();
    for (const Block& b : blocks_) ids->push_back(const_cast<Block*>(&b));
  }
  void GetParameterBlocksForResidualBlock(const ResidualBlockId id, std::vector<double*>* out) const {
    out->clear();
    out->push_back(params_[static_cast<size_t>(id->param[0])]);
    out->push_back(params_[static_cast<size_t>(id->param[1])]);
  }
  const CostFunction* GetCostFunctionForResidualBlock(const ResidualBlockId id) const { return id->cost; }
  // cost and residuals at the current parameter values (gradient / jacobian: not provided by the stand-in)
  bool Evaluate(const EvaluateOptions&, double* cost, std::vector<double>* residuals, std::vector<double>* gradient,
                CRSMatrix* jacobian) {
    if (gradient || jacobian) return false;
    double c = 0;
    if (residuals) residuals->clear();
    for (Block& b : blocks_) {
      std::vector<double> r(static_cast<size_t>(b.cost->num_residuals()));
      double* params[2] = {params_[static_cast<size_t>(b.param[0])], params_[static_cast<size_t>(b.param[1])]};
      if (!b.cost->Evaluate(params, r.data(), nullptr)) return false;
      for (double v : r) c += 0.5 * v * v;
      if (residuals) residuals->insert(residuals->end(), r.begin(), r.end());
    }
    if (cost) *cost = c;
    return true;
  }

  // (stub: the solver below reads these directly)
  typedef ResidualBlock Block;
  int Index(double* values, int size) {
    for (size_t k = 0; k < params_.size(); ++k)
      if (params_[k] == values) return static_cast<int>(k);
    params_.push_back(values);
    sizes_.push_back(size);
    constant_.push_back(false);
    parameterization_.push_back(nullptr);
    return static_cast<int>(params_.size()) - 1;
  }
  Options options_;
  std::deque<Block> blocks_;   // (stable addresses: ResidualBlockId points into it)
  std::vector<double*> params_;
  std::vector<int> sizes_;
  std::vector<bool> constant_;
  std::vector<LocalParameterization*> parameterization_;
};

// NumericDiffCostFunction<Functor, CENTRAL, DYNAMIC, 4, 4>(functor, ownership, num_residuals)
// (submap_registration_helper.cpp:54-57): residuals from functor->Evaluate, Jacobians by central differences
template <typename Functor, NumericDiffMethodType kMethod, int kNumResiduals, int N0, int N1>
class NumericDiffCostFunction : public CostFunction {
 public:
  NumericDiffCostFunction(Functor* functor, Ownership ownership, int num_residuals = kNumResiduals)
      : functor_(functor), ownership_(ownership) {
    set_num_residuals(num_residuals);
    mutable_parameter_block_sizes()->push_back(N0);
    mutable_parameter_block_sizes()->push_back(N1);
  }
  ~NumericDiffCostFunction() override {
    if (ownership_ == TAKE_OWNERSHIP) delete functor_;
  }
  bool Evaluate(double const* const* parameters, double* residuals, double** jacobians) const override {
    if (!functor_->Evaluate(parameters, residuals, nullptr)) return false;
    if (!jacobians) return true;
    const int n = num_residuals();
    const int sizes[2] = {N0, N1};
    std::vector<double> plus(static_cast<size_t>(n)), minus(static_cast<size_t>(n));
    for (int b = 0; b < 2; ++b) {
      if (!jacobians[b]) continue;
      std::vector<double> x0(parameters[0], parameters[0] + N0), x1(parameters[1], parameters[1] + N1);
      double* px[2] = {x0.data(), x1.data()};
      for (int a = 0; a < sizes[b]; ++a) {
        const double keep = px[b][a], h = 1e-6 * std::fmax(1.0, std::fabs(keep));
        px[b][a] = keep + h;
        if (!functor_->Evaluate(px, plus.data(), nullptr)) return false;
        px[b][a] = keep - h;
        if (!functor_->Evaluate(px, minus.data(), nullptr)) return false;
        px[b][a] = keep;
        for (int i = 0; i < n; ++i) jacobians[b][sizes[b] * i + a] = (plus[static_cast<size_t>(i)] - minus[static_cast<size_t>(i)]) / (2.0 * h);
      }
    }
    return true;
  }

 private:
  Functor* functor_;
  Ownership ownership_;
};

struct Solver {
  struct Options {
    int max_num_iterations = 50;
    double function_tolerance = 1e-6, gradient_tolerance = 1e-10, parameter_tolerance = 1e-8;
    double initial_trust_region_radius = 1e4;
    double max_solver_time_in_seconds = 1e9;   // (accepted, not enforced)
    int num_threads = 1;
    LinearSolverType linear_solver_type = DENSE_QR;   // (accepted: the stand-in solves dense normal equations)
    EvaluationCallback* evaluation_callback = nullptr;
  };
  struct Summary {
    int num_iterations = 0, num_evaluations = 0;   // evaluations: cost-only + with Jacobians
    int num_cost_only_evaluations = 0, num_jacobian_evaluations = 0;
    double initial_cost = 0, final_cost = 0;
    const char* termination = "";
    bool IsSolutionUsable() const { return termination[0] == 'C' || termination[0] == 'N'; }  // CONVERGENCE / NO_CONVERGENCE
    std::string BriefReport() const {
      return std::string("stand-in Ceres: ") + termination + ", iterations " + std::to_string(num_iterations) +
             ", cost " + std::to_string(initial_cost) + " -> " + std::to_string(final_cost);
    }
    std::string FullReport() const { return BriefReport(); }
  };
};

struct SolverImpl {
  // cost = 0.5 sum r^2; g = J^T r; H = J^T J over the free parameters (dense)
  // new_evaluation_point: false when the parameters are where the previous evaluation left them (Ceres evaluates the
  // Jacobians of a step it has just accepted on a cost-only evaluation that way)
  static bool Evaluate(const Solver::Options& o, Problem* p, const std::vector<int>& offset, int nf, double* cost,
                       std::vector<double>* g, std::vector<double>* H, bool new_evaluation_point = true) {
    if (o.evaluation_callback) o.evaluation_callback->PrepareForEvaluation(true, new_evaluation_point);
    *cost = 0;
    g->assign(static_cast<size_t>(nf), 0.0);
    H->assign(static_cast<size_t>(nf) * nf, 0.0);
    for (Problem::Block& b : p->blocks_) {
      const int nr = b.cost->num_residuals();
      std::vector<double> r(static_cast<size_t>(nr)), J0(static_cast<size_t>(nr) * 4), J1(static_cast<size_t>(nr) * 4);
      double* params[2] = {p->params_[static_cast<size_t>(b.param[0])], p->params_[static_cast<size_t>(b.param[1])]};
      double* jac[2] = {p->constant_[static_cast<size_t>(b.param[0])] ? nullptr : J0.data(),
                        p->constant_[static_cast<size_t>(b.param[1])] ? nullptr : J1.data()};
      if (!b.cost->Evaluate(params, r.data(), jac)) return false;
      for (int i = 0; i < nr; ++i) *cost += 0.5 * r[static_cast<size_t>(i)] * r[static_cast<size_t>(i)];
      for (int s = 0; s < 2; ++s) {
        if (!jac[s]) continue;
        const int os = offset[static_cast<size_t>(b.param[s])];
        for (int i = 0; i < nr; ++i)
          for (int a = 0; a < 4; ++a) {
            const double ja = jac[s][4 * i + a];
            (*g)[static_cast<size_t>(os + a)] += ja * r[static_cast<size_t>(i)];
            for (int t = 0; t < 2; ++t) {
              if (!jac[t]) continue;
              const int ot = offset[static_cast<size_t>(b.param[t])];
              for (int c = 0; c < 4; ++c)
                (*H)[static_cast<size_t>(os + a) * nf + static_cast<size_t>(ot + c)] += ja * jac[t][4 * i + c];
            }
          }
      }
    }
    return true;
  }
  // cost only, as Ceres' trust-region loop evaluates every trial step: `jacobians == nullptr` for every block, the
  // callback told so beforehand (ProgramEvaluator: PrepareForEvaluation(jacobian != nullptr || gradient != nullptr, ...))
  static bool EvaluateCost(const Solver::Options& o, Problem* p, double* cost) {
    if (o.evaluation_callback) o.evaluation_callback->PrepareForEvaluation(false, true);
    *cost = 0;
    for (Problem::Block& b : p->blocks_) {
      const int nr = b.cost->num_residuals();
      std::vector<double> r(static_cast<size_t>(nr));
      double* params[2] = {p->params_[static_cast<size_t>(b.param[0])], p->params_[static_cast<size_t>(b.param[1])]};
      if (!b.cost->Evaluate(params, r.data(), nullptr)) return false;
      for (int i = 0; i < nr; ++i) *cost += 0.5 * r[static_cast<size_t>(i)] * r[static_cast<size_t>(i)];
    }
    return true;
  }
  // solves (H + diag(d)) x = -g by Cholesky; false if not positive definite
  static bool SolveDamped(const std::vector<double>& H, const std::vector<double>& d, const std::vector<double>& g, int n,
                          std::vector<double>* x) {
    std::vector<double> L(H);
    for (int i = 0; i < n; ++i) L[static_cast<size_t>(i) * n + i] += d[static_cast<size_t>(i)];
    for (int j = 0; j < n; ++j) {
      double s = L[static_cast<size_t>(j) * n + j];
      for (int k = 0; k < j; ++k) s -= L[static_cast<size_t>(j) * n + k] * L[static_cast<size_t>(j) * n + k];
      if (!(s > 0)) return false;
      L[static_cast<size_t>(j) * n + j] = std::sqrt(s);
      for (int i = j + 1; i < n; ++i) {
        double t = L[static_cast<size_t>(i) * n + j];
        for (int k = 0; k < j; ++k) t -= L[static_cast<size_t>(i) * n + k] * L[static_cast<size_t>(j) * n + k];
        L[static_cast<size_t>(i) * n + j] = t / L[static_cast<size_t>(j) * n + j];
      }
    }
    x->assign(static_cast<size_t>(n), 0.0);
    for (int i = 0; i < n; ++i) {
      double t = -g[static_cast<size_t>(i)];
      for (int k = 0; k < i; ++k) t -= L[static_cast<size_t>(i) * n + k] * (*x)[static_cast<size_t>(k)];
      (*x)[static_cast<size_t>(i)] = t / L[static_cast<size_t>(i) * n + i];
    }
    for (int i = n - 1; i >= 0; --i) {
      double t = (*x)[static_cast<size_t>(i)];
      for (int k = i + 1; k < n; ++k) t -= L[static_cast<size_t>(k) * n + i] * (*x)[static_cast<size_t>(k)];
      (*x)[static_cast<size_t>(i)] = t / L[static_cast<size_t>(i) * n + i];
    }
    return true;
  }
};

// ceres::Covariance (pose_graph.cpp:117-163): dense stand-in -- (J^T J)^-1 over the free parameter blocks at the current
// parameter values, Jacobians requested from every cost function exactly as an evaluation with Jacobians requests them.
// Ceres (covariance_impl.cc) fails on a rank-deficient Jacobian unless told otherwise, as a failed Cholesky does here;
// a constant block has zero covariance; the local parameterizations voxgraph uses (identity, angle wrap: node.cpp) have
// identity Jacobians, so the tangent-space lift is the identity.  A block that was not requested is not served.
class Covariance {
 public:
  struct Options {};
  explicit Covariance(const Options&) {}
  bool Compute(const std::vector<std::pair<const double*, const double*> >& blocks, Problem* p) {
    problem_ = p;
    requested_ = blocks;
    offset_.assign(p->params_.size(), -1);
    nf_ = 0;
    for (size_t k = 0; k < p->params_.size(); ++k)
      if (!p->constant_[k]) {
        offset_[k] = nf_;
        nf_ += p->sizes_[k];
      }
    Solver::Options o;
    double cost = 0;
    std::vector<double> g, H;
    if (!SolverImpl::Evaluate(o, p, offset_, nf_, &cost, &g, &H)) return false;
    inverse_.assign(static_cast<size_t>(nf_) * nf_, 0.0);
    const std::vector<double> no_damping(static_cast<size_t>(nf_), 0.0);
    std::vector<double> e(static_cast<size_t>(nf_)), x;
    for (int i = 0; i < nf_; ++i) {
      e.assign(static_cast<size_t>(nf_), 0.0);
      e[static_cast<size_t>(i)] = -1.0;   // SolveDamped solves H x = -g
      if (!SolverImpl::SolveDamped(H, no_damping, e, nf_, &x)) return false;
      for (int j = 0; j < nf_; ++j) inverse_[static_cast<size_t>(j) * nf_ + i] = x[static_cast<size_t>(j)];
    }
    return true;
  }
  bool GetCovarianceBlock(const double* a, const double* b, double* out) const {
    if (!problem_) return false;
    bool asked = false;
    for (const auto& r : requested_) asked = asked || (r.first == a && r.second == b) || (r.first == b && r.second == a);
    int ka = -1, kb = -1;
    for (size_t k = 0; k < problem_->params_.size(); ++k) {
      if (problem_->params_[k] == a) ka = static_cast<int>(k);
      if (problem_->params_[k] == b) kb = static_cast<int>(k);
    }
    if (!asked || ka < 0 || kb < 0) return false;
    const int na = problem_->sizes_[static_cast<size_t>(ka)], nb = problem_->sizes_[static_cast<size_t>(kb)];
    for (int i = 0; i < na; ++i)
      for (int j = 0; j < nb; ++j) {
        const int oa = offset_[static_cast<size_t>(ka)], ob = offset_[static_cast<size_t>(kb)];
        out[i * nb + j] = (oa < 0 || ob < 0) ? 0.0 : inverse_[static_cast<size_t>(oa + i) * nf_ + static_cast<size_t>(ob + j)];
      }
    return true;
  }

 private:
  Problem* problem_ = nullptr;
  std::vector<std::pair<const double*, const double*> > requested_;
  std::vector<int> offset_;
  int nf_ = 0;
  std::vector<double> inverse_;
};

inline void Solve(const Solver::Options& o, Problem* p, Solver::Summary* summary) {
  std::vector<int> offset(p->params_.size(), -1);
  int nf = 0;
  for (size_t k = 0; k < p->params_.size(); ++k)
    if (!p->constant_[k]) {
      offset[k] = nf;
      nf += p->sizes_[k];
    }
  auto get = [&](std::vector<double>* x) {
    x->assign(static_cast<size_t>(nf), 0.0);
    for (size_t k = 0; k < p->params_.size(); ++k)
      if (offset[k] >= 0)
        for (int a = 0; a < p->sizes_[k]; ++a) (*x)[static_cast<size_t>(offset[k] + a)] = p->params_[k][a];
  };
  // x <- Plus(x, step) per parameter block (its local parameterization, or plain addition)
  auto plus = [&](const std::vector<double>& x, const std::vector<double>& step, std::vector<double>* out) {
    *out = x;
    for (size_t k = 0; k < p->params_.size(); ++k) {
      if (offset[k] < 0) continue;
      const size_t o = static_cast<size_t>(offset[k]);
      if (p->parameterization_[k]) {
        p->parameterization_[k]->Plus(&x[o], &step[o], &(*out)[o]);
      } else {
        for (int a = 0; a < p->sizes_[k]; ++a) (*out)[o + static_cast<size_t>(a)] = x[o + static_cast<size_t>(a)] + step[o + static_cast<size_t>(a)];
      }
    }
  };
  auto set = [&](const std::vector<double>& x) {
    for (size_t k = 0; k < p->params_.size(); ++k)
      if (offset[k] >= 0)
        for (int a = 0; a < p->sizes_[k]; ++a) p->params_[k][a] = x[static_cast<size_t>(offset[k] + a)];
  };
  Solver::Summary& s = *summary;
  s = Solver::Summary();
  std::vector<double> x, g, H, ng, nH, step, d(static_cast<size_t>(nf));
  get(&x);
  double cost = 0;
  s.termination = "FAILURE";
  if (!SolverImpl::Evaluate(o, p, offset, nf, &cost, &g, &H)) return;
  s.num_evaluations = s.num_jacobian_evaluations = 1;
  s.initial_cost = s.final_cost = cost;
  double radius = o.initial_trust_region_radius, decrease = 2.0;
  s.termination = "NO_CONVERGENCE";
  while (s.num_iterations < o.max_num_iterations) {
    ++s.num_iterations;
    double gmax = 0, xnorm = 0;
    for (int i = 0; i < nf; ++i) {
      gmax = std::fmax(gmax, std::fabs(g[static_cast<size_t>(i)]));
      xnorm += x[static_cast<size_t>(i)] * x[static_cast<size_t>(i)];
      d[static_cast<size_t>(i)] = std::fmin(std::fmax(H[static_cast<size_t>(i) * nf + i], 1e-6), 1e32) / radius;
    }
    if (gmax <= o.gradient_tolerance) {
      s.termination = "CONVERGENCE (gradient)";
      break;
    }
    if (!SolverImpl::SolveDamped(H, d, g, nf, &step)) {
      radius /= decrease;
      decrease *= 2;
      continue;
    }
    double snorm = 0, model = 0;
    for (int i = 0; i < nf; ++i) {
      snorm += step[static_cast<size_t>(i)] * step[static_cast<size_t>(i)];
      double hs = 0;
      for (int j = 0; j < nf; ++j) hs += H[static_cast<size_t>(i) * nf + j] * step[static_cast<size_t>(j)];
      model -= step[static_cast<size_t>(i)] * (g[static_cast<size_t>(i)] + 0.5 * hs);
    }
    if (std::sqrt(snorm) <= o.parameter_tolerance * (std::sqrt(xnorm) + o.parameter_tolerance)) {
      s.termination = "CONVERGENCE (parameter)";
      break;
    }
    std::vector<double> cand;
    plus(x, step, &cand);
    set(cand);
    // the trial step: its cost alone (TrustRegionMinimizer: evaluator->Evaluate(x_candidate, &cost, nullptr, nullptr, nullptr))
    double ncost = 0;
    if (!SolverImpl::EvaluateCost(o, p, &ncost)) {
      set(x);
      return;
    }
    ++s.num_evaluations;
    ++s.num_cost_only_evaluations;
    const double rho = model > 0 ? (cost - ncost) / model : -1.0;
    if (rho > 1e-3) {
      const double rel = std::fabs(cost - ncost) / std::fmax(cost, 1e-300);
      // accepted: gradient and Jacobian at the point just evaluated (EvaluateGradientAndJacobian(new_evaluation_point = false))
      if (!SolverImpl::Evaluate(o, p, offset, nf, &ncost, &ng, &nH, /*new_evaluation_point=*/false)) {
        set(x);
        return;
      }
      ++s.num_evaluations;
      ++s.num_jacobian_evaluations;
      x = cand;
      cost = ncost;
      g = ng;
      H = nH;
      const double t = 2.0 * rho - 1.0;
      radius = std::fmin(radius / std::fmax(1.0 / 3.0, 1.0 - t * t * t), 1e16);
      decrease = 2.0;
      if (rel <= o.function_tolerance) {
        s.termination = "CONVERGENCE (function)";
        break;
      }
    } else {
      set(x);
      radius /= decrease;
      decrease *= 2.0;
    }
  }
  set(x);
  s.final_cost = cost;
}
}  // namespace ceres
#endif

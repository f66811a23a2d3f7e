// Batched Ceres integration: every registration residual block of a pose graph is
// evaluated by ONE GPU launch per solver evaluation instead of one launch + 72 B/residual
// of PCIe traffic per block.
//
//   voxgraph_amd::GpuRegistrationBatch batch(gpu_ctx);
//   // registration_constraint.cpp:33-42, for every registration constraint:
//   problem->AddResidualBlock(batch.AddConstraint(reg_handle, pose_first, pose_second),
//                             nullptr, pose_first, pose_second);
//   batch.Finalize();
//   ceres_options.evaluation_callback = &batch;      // next to pose_graph.cpp:93-97
//   ceres::Solve(ceres_options, problem, &summary);  // pose_graph.cpp:101, unchanged
//
// Each constraint appears to Ceres as a 9-residual block [J_c | r_c] with exactly the
// normal equations of its N-residual original (vgx_reg_compress_normal): legitimate
// because the reference attaches no robust loss to registration constraints
// (registration_constraint.cpp:10, constraint.h:34).
//
// Ceres' `evaluate_jacobians` is honoured (the reference does no Jacobian work when `jacobians == nullptr`,
// registration_cost_function.cpp:179, and Levenberg-Marquardt evaluates every trial step that way): a cost-only
// request runs the cost-only pass (vgx_reg_batch_evaluate_cost: one sum per constraint, no gradient, nothing to
// compress) and every block answers with the residual vector (sqrt(cost), 0, ..., 0) -- all Ceres reads there is its
// squared norm, and that cost is, bit for bit, the one the full evaluation at the same point reports.
//
// Evaluations Ceres does NOT announce (round 6): with the callback in Solver::Options (the Ceres the reference builds
// against) only ceres::Solve calls PrepareForEvaluation -- Problem::Evaluate (PoseGraph::getVisualizationEdges,
// pose_graph.cpp:173-174) and Covariance::Compute (getEdgeCovarianceMap, :140) evaluate the blocks without it, possibly
// right after a solve whose last announced point was a REJECTED step.  A block therefore checks the `parameters` it is
// handed against the poses its cache was computed at (bit compare, 8 doubles) and, on a mismatch or when Jacobians are
// asked of a cost-only cache, has the batch evaluated again at the point the user's parameter blocks hold -- which is
// where Ceres evaluates in both cases.  If the parameters are not what the user's blocks hold either (an evaluation at
// explicitly passed parameter values), the block reports an evaluation failure rather than a value for another point.
#ifndef VOXGRAPH_AMD_CPP_GPU_REGISTRATION_BATCH_H_
#define VOXGRAPH_AMD_CPP_GPU_REGISTRATION_BATCH_H_

#include <ceres/ceres.h>

#include <cmath>
#include <cstring>
#include <map>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

#include "voxgraph_amd.h"

namespace voxgraph_amd {

// What the single- and multi-GPU callbacks share: the constraint list, the pose-block -> node
// numbering, the [n][45] normal blocks of the last evaluation and their 9-residual stand-ins.
// A derived class only says how the normal blocks are produced (EvaluateNormals).
class GpuRegistrationBlocks : public ceres::EvaluationCallback {
 public:
  // ceres::EvaluationCallback: the user's parameter blocks already hold the point to
  // evaluate when this is called.
  //
  // new_evaluation_point == false: what is cached answers if it holds what is asked for -- the Jacobians of a step
  // Ceres has just accepted on its cost are NOT in a cost-only cache and are evaluated then, at the same point.  With
  // SAMPLING constraints that second evaluation draws anew (as every Evaluate of the reference does), and a request
  // the cache can answer does not draw where the reference would: fewer draws, each a legal one (include/voxgraph_amd.h,
  // vgx_reg_batch_evaluate_cost).
  void PrepareForEvaluation(bool evaluate_jacobians, bool new_evaluation_point) override {
    std::lock_guard<std::mutex> lk(mu_);
    Prepare(evaluate_jacobians, new_evaluation_point);
  }

  int num_constraints() const { return static_cast<int>(regs_.size()); }
  // evaluations so far by route: the full fused pass (normal equations, compressed) / the cost-only pass
  long full_evaluations() const { return full_evaluations_; }
  long cost_only_evaluations() const { return cost_only_evaluations_; }
  // evaluations a block asked for itself because its parameters were not the cached point's (see the header comment)
  long unannounced_evaluations() const { return unannounced_evaluations_; }

 protected:
  virtual void EvaluateNormals(const double* poses, int32_t n_nodes, double* normal, int32_t* status) = 0;
  virtual void EvaluateCosts(const double* poses, int32_t n_nodes, double* cost, int32_t* status) = 0;

 private:
  void Prepare(bool evaluate_jacobians, bool new_evaluation_point) {   // (mu_ held)
    if (!finalized_) throw std::logic_error("GpuRegistrationBatch: Finalize() was not called");
    if (!new_evaluation_point && valid_ && (have_jacobians_ || !evaluate_jacobians) && PosesCurrent()) return;
    for (size_t k = 0; k < nodes_.size(); ++k) std::memcpy(&poses_[4 * k], nodes_[k], 4 * sizeof(double));
    if (evaluate_jacobians) {
      EvaluateNormals(poses_.data(), static_cast<int32_t>(nodes_.size()), normal_.data(), status_.data());
      for (size_t c = 0; c < regs_.size(); ++c)
        vgx_reg_compress_normal(&normal_[45 * c], &compressed_r_[9 * c], &compressed_j_[72 * c]);
      ++full_evaluations_;
    } else {
      EvaluateCosts(poses_.data(), static_cast<int32_t>(nodes_.size()), cost_.data(), status_.data());
      for (size_t c = 0; c < regs_.size(); ++c) {
        double* r = &compressed_r_[9 * c];
        r[0] = std::sqrt(cost_[c] > 0.0 ? cost_[c] : 0.0);
        for (int m = 1; m < 9; ++m) r[m] = 0.0;
      }
      ++cost_only_evaluations_;
    }
    have_jacobians_ = evaluate_jacobians;
    valid_ = true;
  }
  // the user's parameter blocks still hold the poses the cache was computed at
  bool PosesCurrent() const {
    for (size_t k = 0; k < nodes_.size(); ++k)
      if (std::memcmp(&poses_[4 * k], nodes_[k], 4 * sizeof(double)) != 0) return false;
    return true;
  }
  // the cache answers block `index` for these parameters
  bool Serves(int index, double const* const* parameters, bool jacobians) const {
    if (!valid_ || (jacobians && !have_jacobians_)) return false;
    for (int side = 0; side < 2; ++side)
      if (std::memcmp(parameters[side], &poses_[4 * static_cast<size_t>(node_pair_[2 * static_cast<size_t>(index) + side])],
                      4 * sizeof(double)) != 0)
        return false;
    return true;
  }

 protected:
  ceres::CostFunction* AddBlock(vgx_reg reg, const double* pose_reference, const double* pose_reading) {
    if (finalized_) throw std::logic_error("GpuRegistrationBatch: AddConstraint after Finalize");
    regs_.push_back(reg);
    node_pair_.push_back(NodeIndex(pose_reference));
    node_pair_.push_back(NodeIndex(pose_reading));
    return new Block(this, static_cast<int>(regs_.size()) - 1);
  }

  void FinalizeBlocks() {
    const size_t n = regs_.size();
    normal_.assign(n * 45, 0.0);
    cost_.assign(n, 0.0);
    status_.assign(n, 0);
    compressed_r_.assign(n * 9, 0.0);
    compressed_j_.assign(n * 72, 0.0);
    poses_.assign(nodes_.size() * 4, 0.0);
    finalized_ = true;
  }

  std::vector<vgx_reg> regs_;
  std::vector<int32_t> node_pair_;

 private:
  // The 9-residual stand-in for one RegistrationCostFunction.
  class Block : public ceres::SizedCostFunction<9, 4, 4> {
   public:
    Block(GpuRegistrationBlocks* owner, int index) : owner_(owner), index_(index) {}
    bool Evaluate(double const* const* parameters, double* residuals,
                  double** jacobians) const override {
      GpuRegistrationBlocks& o = *owner_;
      std::lock_guard<std::mutex> lk(o.mu_);   // (Ceres evaluates blocks from several threads: pose_graph.cpp:96)
      if (!o.Serves(index_, parameters, jacobians != nullptr)) {
        // not announced (Problem::Evaluate, Covariance::Compute; see the header comment): the point in the user's blocks
        o.Prepare(jacobians != nullptr || o.have_jacobians_, /*new_evaluation_point=*/true);
        ++o.unannounced_evaluations_;
        if (!o.Serves(index_, parameters, jacobians != nullptr)) return false;  // parameters from somewhere else: no guess
      }
      if (o.status_[static_cast<size_t>(index_)] == VGX_EVALUATE_FALSE) return false;  // .cpp:273
      std::memcpy(residuals, &o.compressed_r_[9 * static_cast<size_t>(index_)], 9 * sizeof(double));
      if (jacobians) {
        const double* J = &o.compressed_j_[72 * static_cast<size_t>(index_)];
        for (int side = 0; side < 2; ++side)
          if (jacobians[side])
            for (int m = 0; m < 9; ++m)
              std::memcpy(&jacobians[side][4 * m], &J[8 * m + 4 * side], 4 * sizeof(double));
      }
      return true;
    }

   private:
    GpuRegistrationBlocks* owner_;
    int index_;
  };

  int32_t NodeIndex(const double* pose) {
    auto it = node_of_.find(pose);
    if (it != node_of_.end()) return it->second;
    const int32_t k = static_cast<int32_t>(nodes_.size());
    nodes_.push_back(pose);
    node_of_[pose] = k;
    return k;
  }

  std::vector<const double*> nodes_;
  std::map<const double*, int32_t> node_of_;
  std::vector<double> poses_, normal_, cost_, compressed_r_, compressed_j_;
  std::vector<int32_t> status_;
  bool finalized_ = false;
  bool valid_ = false;
  bool have_jacobians_ = false;   // the cache holds the compressed Jacobians (else residual-only blocks of a cost-only pass)
  long full_evaluations_ = 0, cost_only_evaluations_ = 0, unannounced_evaluations_ = 0;
  std::mutex mu_;   // the cache: PrepareForEvaluation / a block's own refresh write it, every block's Evaluate reads it
};

// One GPU.
class GpuRegistrationBatch : public GpuRegistrationBlocks {
 public:
  explicit GpuRegistrationBatch(vgx_ctx ctx) : ctx_(ctx) {}
  ~GpuRegistrationBatch() override {
    if (batch_) vgx_reg_batch_destroy(batch_);
  }
  GpuRegistrationBatch(const GpuRegistrationBatch&) = delete;
  GpuRegistrationBatch& operator=(const GpuRegistrationBatch&) = delete;

  // One registration constraint.  pose_* are the parameter blocks Ceres optimises
  // ({x,y,z,yaw} doubles, Pose4D::optimizationVectorData()); the returned cost function
  // is handed to Problem::AddResidualBlock (which takes ownership by default).
  ceres::CostFunction* AddConstraint(vgx_reg reg, const double* pose_reference, const double* pose_reading) {
    return AddBlock(reg, pose_reference, pose_reading);
  }

  void Finalize() {
    const int n = static_cast<int>(regs_.size());
    if (vgx_reg_batch_create(ctx_, n, regs_.data(), node_pair_.data(), nullptr, n, &batch_) != VGX_OK)
      throw std::runtime_error(std::string("vgx_reg_batch_create: ") + vgx_last_error(ctx_));
    FinalizeBlocks();
  }

 protected:
  void EvaluateNormals(const double* poses, int32_t n_nodes, double* normal, int32_t* status) override {
    if (vgx_reg_batch_evaluate_normal(batch_, poses, n_nodes, nullptr, normal, status) != VGX_OK)
      throw std::runtime_error(std::string("vgx_reg_batch_evaluate_normal: ") + vgx_last_error(ctx_));
  }
  void EvaluateCosts(const double* poses, int32_t n_nodes, double* cost, int32_t* status) override {
    if (vgx_reg_batch_evaluate_cost(batch_, poses, n_nodes, nullptr, cost, status) != VGX_OK)
      throw std::runtime_error(std::string("vgx_reg_batch_evaluate_cost: ") + vgx_last_error(ctx_));
  }

 private:
  vgx_ctx ctx_;
  vgx_reg_batch batch_ = nullptr;
};

}  // namespace voxgraph_amd

#endif  // VOXGRAPH_AMD_CPP_GPU_REGISTRATION_BATCH_H_

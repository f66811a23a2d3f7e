// Batched Ceres integration that keeps the reference's residual blocks as they are: every registration constraint stays an
// N-residual ceres::CostFunction whose Evaluate returns exactly what voxgraph::RegistrationCostFunction::Evaluate returns
// (f64, Ceres layout, value for value) -- but all of them are evaluated by ONE GPU launch per solver evaluation
// (vgx_reg_batch_evaluate_rows_f64) and a block's Evaluate only fetches its slice (vgx_reg_batch_fetch_rows_f64; SURVEY.md
// 8b's "vgx_reg_fetch").
//
//   voxgraph_amd::GpuRegistrationRows rows(gpu_ctx);
//   // registration_constraint.cpp:33-42, for every registration constraint:
//   problem->AddResidualBlock(rows.AddConstraint(reg_handle, pose_first, pose_second),
//                             loss_function, pose_first, pose_second);
//   rows.Finalize();
//   ceres_options.evaluation_callback = &rows;       // next to pose_graph.cpp:93-97
//
// Against the other two routes:
//   * drop-in GpuRegistrationCostFunction: the same values, one launch PER BLOCK and per evaluation;
//   * GpuRegistrationBatch: one launch and 45 numbers per constraint instead of 72 bytes per residual over PCIe, but each
//     block is a 9-residual stand-in with the same normal equations -- exact only while no ceres::LossFunction is attached
//     (the reference attaches none to registration constraints: registration_constraint.cpp:10) and of no use to a caller
//     who wants the residuals themselves.
// This one is for those cases: a robust loss on registration constraints, per-residual inspection, covariance estimation at
// full fidelity -- with one launch per evaluation.  It moves the drop-in's bytes over PCIe.
//
// Evaluations Ceres does not announce (Problem::Evaluate, Covariance::Compute: gpu_registration_batch.h has the story) are
// noticed the same way: a block compares the parameters it is handed with the poses the rows were evaluated at.
#ifndef VOXGRAPH_AMD_CPP_GPU_REGISTRATION_ROWS_H_
#define VOXGRAPH_AMD_CPP_GPU_REGISTRATION_ROWS_H_

#include <ceres/ceres.h>

#include <cstring>
#include <map>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

#include "voxgraph_amd.h"

namespace voxgraph_amd {

class GpuRegistrationRows : public ceres::EvaluationCallback {
 public:
  explicit GpuRegistrationRows(vgx_ctx ctx) : ctx_(ctx) {}
  ~GpuRegistrationRows() override {
    if (batch_) vgx_reg_batch_destroy(batch_);
  }
  GpuRegistrationRows(const GpuRegistrationRows&) = delete;
  GpuRegistrationRows& operator=(const GpuRegistrationRows&) = delete;

  // One registration constraint; the returned cost function (num_residuals = the constraint's, two blocks of 4) goes to
  // Problem::AddResidualBlock, which takes ownership by default.
  ceres::CostFunction* AddConstraint(vgx_reg reg, const double* pose_reference, const double* pose_reading) {
    if (finalized_) throw std::logic_error("GpuRegistrationRows: AddConstraint after Finalize");
    regs_.push_back(reg);
    node_pair_.push_back(NodeIndex(pose_reference));
    node_pair_.push_back(NodeIndex(pose_reading));
    return new Block(this, static_cast<int>(regs_.size()) - 1, static_cast<int>(vgx_reg_num_residuals(reg)));
  }

  void Finalize() {
    const int n = static_cast<int>(regs_.size());
    if (vgx_reg_batch_create(ctx_, n, regs_.data(), node_pair_.data(), nullptr, n, &batch_) != VGX_OK)
      throw std::runtime_error(std::string("vgx_reg_batch_create: ") + vgx_last_error(ctx_));
    status_.assign(static_cast<size_t>(n), 0);
    poses_.assign(nodes_.size() * 4, 0.0);
    finalized_ = true;
  }

  // ceres::EvaluationCallback: the user's parameter blocks hold the point to evaluate.  new_evaluation_point == false and
  // rows that hold what is asked for: nothing to do (with SAMPLING constraints an evaluation that does run draws anew, as
  // every Evaluate of the reference does).
  void PrepareForEvaluation(bool evaluate_jacobians, bool new_evaluation_point) override {
    std::lock_guard<std::mutex> lk(mu_);
    Prepare(evaluate_jacobians, new_evaluation_point);
  }

  int num_constraints() const { return static_cast<int>(regs_.size()); }
  long evaluations() const { return evaluations_; }
  long unannounced_evaluations() const { return unannounced_evaluations_; }

 private:
  void Prepare(bool evaluate_jacobians, bool new_evaluation_point) {   // (mu_ held)
    if (!finalized_) throw std::logic_error("GpuRegistrationRows: Finalize() was not called");
    if (!new_evaluation_point && valid_ && (have_jacobians_ || !evaluate_jacobians) && PosesCurrent()) return;
    for (size_t k = 0; k < nodes_.size(); ++k) std::memcpy(&poses_[4 * k], nodes_[k], 4 * sizeof(double));
    const int want = evaluate_jacobians ? 1 : 0;
    if (vgx_reg_batch_evaluate_rows_f64(batch_, poses_.data(), static_cast<int32_t>(nodes_.size()), want, want,
                                        status_.data()) != VGX_OK)
      throw std::runtime_error(std::string("vgx_reg_batch_evaluate_rows_f64: ") + vgx_last_error(ctx_));
    have_jacobians_ = evaluate_jacobians;
    valid_ = true;
    ++evaluations_;
  }
  bool PosesCurrent() const {
    for (size_t k = 0; k < nodes_.size(); ++k)
      if (std::memcmp(&poses_[4 * k], nodes_[k], 4 * sizeof(double)) != 0) return false;
    return true;
  }
  bool Serves(int index, double const* const* parameters, bool jacobians) const {
    if (!valid_ || (jacobians && !have_jacobians_)) return false;
    for (int side = 0; side < 2; ++side)
      if (std::memcmp(parameters[side], &poses_[4 * static_cast<size_t>(node_pair_[2 * static_cast<size_t>(index) + side])],
                      4 * sizeof(double)) != 0)
        return false;
    return true;
  }

  // One RegistrationCostFunction, as Ceres sees it: the reference's sizes (registration_cost_function.cpp:40-55).
  class Block : public ceres::CostFunction {
   public:
    Block(GpuRegistrationRows* owner, int index, int num_residuals) : owner_(owner), index_(index) {
      mutable_parameter_block_sizes()->push_back(4);
      mutable_parameter_block_sizes()->push_back(4);
      set_num_residuals(num_residuals);
    }
    bool Evaluate(double const* const* parameters, double* residuals, double** jacobians) const override {
      GpuRegistrationRows& o = *owner_;
      std::lock_guard<std::mutex> lk(o.mu_);   // (Ceres evaluates blocks from several threads: pose_graph.cpp:96)
      if (!o.Serves(index_, parameters, jacobians != nullptr)) {
        o.Prepare(jacobians != nullptr || o.have_jacobians_, /*new_evaluation_point=*/true);
        ++o.unannounced_evaluations_;
        if (!o.Serves(index_, parameters, jacobians != nullptr)) return false;   // parameters nobody holds: no guess
      }
      if (o.status_[static_cast<size_t>(index_)] == VGX_EVALUATE_FALSE) return false;   // .cpp:273
      const int rc = vgx_reg_batch_fetch_rows_f64(o.batch_, index_, residuals, jacobians ? jacobians[0] : nullptr,
                                                  jacobians ? jacobians[1] : nullptr);
      return rc == VGX_OK;
    }

   private:
    GpuRegistrationRows* owner_;
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

  vgx_ctx ctx_;
  vgx_reg_batch batch_ = nullptr;
  std::vector<vgx_reg> regs_;
  std::vector<int32_t> node_pair_, status_;
  std::vector<const double*> nodes_;
  std::map<const double*, int32_t> node_of_;
  std::vector<double> poses_;
  bool finalized_ = false, valid_ = false, have_jacobians_ = false;
  long evaluations_ = 0, unannounced_evaluations_ = 0;
  std::mutex mu_;
};

}  // namespace voxgraph_amd

#endif  // VOXGRAPH_AMD_CPP_GPU_REGISTRATION_ROWS_H_

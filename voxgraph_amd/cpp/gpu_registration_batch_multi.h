// Multi-GPU form of GpuRegistrationBatch: the same ceres::EvaluationCallback, the constraint list
// pair-sharded over several GPUs of ONE process (voxgraph is one process,
// voxgraph/src/voxgraph_mapping_node.cpp:6-26).
//
//   std::vector<vgx_ctx> gpus = ...;                 // one vgx_ctx per GPU, every finished submap
//                                                    // uploaded to each (UploadFinishedSubmap)
//   voxgraph_amd::GpuRegistrationBatchMulti batch(gpus);
//   // for every registration constraint c (registration_constraint.cpp:33-42):
//   //   weight[c] = registration points of its first submap;  batch.PlanShards(weight) -> shard[c]
//   //   reg[c]    = vgx_reg_create(gpus[shard[c]], first_on_that_gpu, second_on_that_gpu, cfg)
//   problem->AddResidualBlock(batch.AddConstraint(reg[c], pose_first, pose_second), nullptr, ...);
//   batch.Finalize();
//   ceres_options.evaluation_callback = &batch;      // next to pose_graph.cpp:93-97
//
// Per solver evaluation every GPU runs the fused pass on its share from its own host thread; the
// per-constraint normal blocks come back to the host (no collective is needed for residual
// blocks).  Solvers that consume normal equations directly use EvaluateFused(): the per-constraint
// blocks are gathered on GPU 0 through xGMI peer mappings and assembled once, in list order -- the
// single-GPU buffer bit for bit, whatever the placement (vgx_reg_multi_evaluate_fused).
#ifndef VOXGRAPH_AMD_CPP_GPU_REGISTRATION_BATCH_MULTI_H_
#define VOXGRAPH_AMD_CPP_GPU_REGISTRATION_BATCH_MULTI_H_

#include "gpu_registration_batch.h"

namespace voxgraph_amd {

class GpuRegistrationBatchMulti : public GpuRegistrationBlocks {
 public:
  explicit GpuRegistrationBatchMulti(const std::vector<vgx_ctx>& gpus) : gpus_(gpus) {
    if (gpus_.empty()) throw std::invalid_argument("GpuRegistrationBatchMulti: no contexts");
  }
  ~GpuRegistrationBatchMulti() override {
    if (multi_) vgx_reg_multi_destroy(multi_);
  }
  GpuRegistrationBatchMulti(const GpuRegistrationBatchMulti&) = delete;
  GpuRegistrationBatchMulti& operator=(const GpuRegistrationBatchMulti&) = delete;

  // Placement: shard[c] = index into the context list on which constraint c's cost function must be
  // created.  contiguous = true (default): the list cut into consecutive runs of equal weight
  // (vgx_contiguous_shards) -- voxgraph creates constraints in submap order, so a GPU then needs only
  // the submaps of one stretch of the map (a third of them at 8 GPUs) and balances as well as LPT;
  // false: greedy longest-processing-time (vgx_lpt_shards).  Results do not depend on the choice.
  std::vector<int32_t> PlanShards(const std::vector<int64_t>& weight_per_constraint, bool contiguous = true) const {
    std::vector<int32_t> shard(weight_per_constraint.size());
    const int rc = (contiguous ? vgx_contiguous_shards : vgx_lpt_shards)(
        static_cast<int32_t>(shard.size()), weight_per_constraint.data(), static_cast<int32_t>(gpus_.size()), shard.data());
    if (rc != VGX_OK) throw std::runtime_error("constraint placement failed");
    return shard;
  }

  ceres::CostFunction* AddConstraint(vgx_reg reg, const double* pose_reference, const double* pose_reading) {
    return AddBlock(reg, pose_reference, pose_reading);
  }

  void Finalize() {
    const int n = static_cast<int>(regs_.size());
    if (vgx_reg_multi_create(static_cast<int32_t>(gpus_.size()), gpus_.data(), n, regs_.data(),
                             node_pair_.data(), &multi_) != VGX_OK)
      throw std::runtime_error(std::string("vgx_reg_multi_create: ") + vgx_last_error(gpus_[0]));
    FinalizeBlocks();
  }

  // How EvaluateFused's per-GPU buffers meet: VGX_REDUCE_PEER_SUM (default: GPU 0 sums them in GPU order
  // over xGMI peer mappings, bitwise reproducible) or VGX_REDUCE_RCCL (one ncclAllReduce per evaluation).
  // Call after Finalize().
  void SetReduction(int32_t reduction) {
    if (vgx_reg_multi_set_reduction(multi_, reduction) != VGX_OK)
      throw std::runtime_error(std::string("vgx_reg_multi_set_reduction: ") + vgx_last_error(gpus_[0]));
  }

  // The assembled normal equations of every registration constraint at `poses` ([n_nodes][4], node
  // numbering = order in which pose blocks were first passed to AddConstraint):
  // [cost | J^T r (4 n) | diagonal blocks (16 n) | off-diagonal blocks (16 m)], summed over the GPUs.
  std::vector<double> EvaluateFused(const double* poses, int32_t n_nodes) {
    std::vector<double> fused(static_cast<size_t>(vgx_reg_fused_size(n_nodes, num_constraints())));
    if (vgx_reg_multi_evaluate_fused(multi_, poses, n_nodes, fused.data(), nullptr) != VGX_OK)
      throw std::runtime_error(std::string("vgx_reg_multi_evaluate_fused: ") + vgx_last_error(gpus_[0]));
    return fused;
  }

 protected:
  void EvaluateNormals(const double* poses, int32_t n_nodes, double* normal, int32_t* status) override {
    if (vgx_reg_multi_evaluate_normal(multi_, poses, n_nodes, normal, status) != VGX_OK)
      throw std::runtime_error(std::string("vgx_reg_multi_evaluate_normal: ") + vgx_last_error(gpus_[0]));
  }
  void EvaluateCosts(const double* poses, int32_t n_nodes, double* cost, int32_t* status) override {
    if (vgx_reg_multi_evaluate_cost(multi_, poses, n_nodes, cost, status) != VGX_OK)
      throw std::runtime_error(std::string("vgx_reg_multi_evaluate_cost: ") + vgx_last_error(gpus_[0]));
  }

 private:
  std::vector<vgx_ctx> gpus_;
  vgx_reg_multi multi_ = nullptr;
};

}  // namespace voxgraph_amd

#endif  // VOXGRAPH_AMD_CPP_GPU_REGISTRATION_BATCH_MULTI_H_

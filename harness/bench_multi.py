"""bench.py's `multi_context` block: the in-process multi-GPU component (vgx_reg_multi_*)."""
import os
import sys
import time

import numpy as np

from harness.bench_common import HBM_PEAK_GBS, place  # noqa: F401

class GpuBackendLite:
    """single batch: fused pass + assembly + copy of the fused buffer to the host (what
    vgx_reg_multi_evaluate_fused returns), without the harness around it"""

    def __init__(self, capi, ctx, batch, n_nodes, torch):
        self.ctx, self.batch, self.n_nodes = ctx, batch, n_nodes
        self.buf = torch.zeros(capi.fused_size(n_nodes, batch.n_global), dtype=torch.float64, device="cuda")
        self.host = torch.zeros_like(self.buf, device="cpu").pin_memory()
        torch.cuda.current_stream().synchronize()
        self.torch = torch

    def __call__(self):
        self.batch.evaluate_normal(self._poses, to_host=False)
        self.batch.assemble(self.n_nodes, self.buf.data_ptr(), zero_first=True)
        self.ctx.synchronize()
        self.host.copy_(self.buf, non_blocking=True)
        self.torch.cuda.current_stream().synchronize()
        return self.host.numpy()


def multi_context_bench(capi, ctx0, torch, args, devices, submaps0, true_poses, pairs, weights, poses, cfg, single_batch,
                        n_sub, n_con):
    """vgx_reg_multi_evaluate_fused over len(devices) contexts (context 0 = ctx0, whose submaps are
    resident already; every other context gets the submaps its LPT share of the constraints needs)."""
    n_ctx = len(devices)
    t_setup = time.perf_counter()
    ctxs = [ctx0] + [capi.Context(d) for d in devices[1:]]
    shard_of = place(weights, n_ctx)
    subs = [dict(enumerate(submaps0))] + [dict() for _ in range(n_ctx - 1)]
    for k_ctx in range(1, n_ctx):
        need = sorted({int(s_) for c in range(n_con) if shard_of[c] == k_ctx for s_ in pairs[c]})
        for k in need:
            sm = capi.Submap.synth_city(ctxs[k_ctx], k, args.voxel_size, 16, args.block_min, args.block_dims,
                                        args.truncation, args.esdf_max, 10.0, true_poses[k], args.seed)
            sm.extract_voxel_points(1.0, 0.3, True)
            sm.release_raw_layers()
            subs[k_ctx][k] = sm
    cfs_m = [capi.RegistrationCostFunction(ctxs[shard_of[c]], subs[shard_of[c]][int(a)], subs[shard_of[c]][int(b)], cfg)
             for c, (a, b) in enumerate(pairs)]
    multi = capi.RegistrationMulti(ctxs, cfs_m, pairs)
    setup_s = time.perf_counter() - t_setup
    for _ in range(2):
        fused_m, _ = multi.evaluate_fused(poses)
    m0 = time.perf_counter()
    for _ in range(args.steps):
        fused_m, _ = multi.evaluate_fused(poses)
    m_ms = (time.perf_counter() - m0) / args.steps * 1e3
    out = {"contexts": n_ctx, "devices": len(set(devices)), "device_ids": devices,
           "what": "vgx_reg_multi_evaluate_fused (shards by bytes moved, one host thread per context, the per-constraint "
                   "blocks gathered on context 0 over peer mappings in event order, ONE assembly, result on the host)",
           "ms_per_evaluation": m_ms,
           "Mresiduals_per_s": float(sum(cf.num_residuals() for cf in cfs_m)) / m_ms / 1e3,
           "constraints_per_context": [int((shard_of == k).sum()) for k in range(n_ctx)],
           "cost": float(fused_m[0]), "setup_s": setup_s}
    if len(set(devices)) == n_ctx and n_ctx > 1:
        # SURVEY.md 8(e) "compare": the same evaluation with ONE ncclAllReduce of the fused buffer instead of
        # the fixed-order sum over peer mappings
        try:
            multi.set_reduction(True)
            for _ in range(2):
                fused_r, _ = multi.evaluate_fused(poses)
            r0 = time.perf_counter()
            for _ in range(args.steps):
                fused_r, _ = multi.evaluate_fused(poses)
            out["rccl_allreduce"] = {"ms_per_evaluation": (time.perf_counter() - r0) / args.steps * 1e3,
                                     "max_rel_diff_vs_peer_sum": float(np.abs(fused_r - fused_m).max() / np.abs(fused_m).max()),
                                     "what": "vgx_reg_multi_set_reduction(VGX_REDUCE_RCCL): ncclAllReduce(sum, f64) in place "
                                             "on every context's stream, one communicator per context"}
            multi.set_reduction(False)
        except Exception as e:                                   # never sink the line on the optional variant
            out["rccl_allreduce"] = {"error": repr(e)}
    if single_batch is not None:
        single = GpuBackendLite(capi, ctx0, single_batch, n_sub, torch)
        single._poses = poses
        for _ in range(2):
            ref_buf = single()
        s0_ = time.perf_counter()
        for _ in range(args.steps):
            ref_buf = single()
        out["single_batch_ms_per_evaluation"] = (time.perf_counter() - s0_) / args.steps * 1e3
        out["single_batch_what"] = "the single batch (evaluate + assemble + copy to the host) on context 0 alone"
        out["max_rel_diff_vs_single_batch"] = float(np.abs(fused_m - ref_buf).max() / np.abs(ref_buf).max())
    multi.destroy()
    for o in cfs_m:
        o.destroy()
    for k_ctx in range(1, n_ctx):
        for sm in subs[k_ctx].values():
            sm.destroy()
        ctxs[k_ctx].close()
    return out


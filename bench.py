#!/usr/bin/env python3
"""bench.py -- REG hot path on BASELINE config 3 (200 submaps @ 256^3, pair-sharded).

One "step" = `--inner` (default 10: a converged solve of this graph takes 9) consecutive
passes of the registration hot path over the whole pose graph's registration
constraints: in every pass every constraint's residuals and both Jacobians are
evaluated (materialising f32 form, 88 B/evaluation) by ONE batched
launch per rank.  Inputs (sampling grids, registration points) are resident in
HBM before the timed region; the only per-pass host->device traffic is the 64-B
pose pack per constraint.  `value` counts evaluations, so it does not depend on
`--inner`; the driver's 20 steps then time >= 1 s instead of 0.1 s.

A second REG workload on the same submaps measures the gather-dominated regime the
88 B contract figure describes (`roofline_full_overlap`): every submap registered against
duplicates of itself perturbed by N(0, 0.3 m) / N(0, 0.05 rad) -- the reference's own
test-bench design (registration_test_bench.cpp:178-185) -- so ~100 % of the evaluations
interpolate.  Constraints are ordered so that consecutive ones never share a submap.

N > 1: the constraint list is sharded across ranks (greedy LPT by residual
count), submaps are replicated, no data-path collective is needed
for this materialising pass; the fixed graph makes this STRONG scaling.  The
fused pass + RCCL all-reduce of the normal-equation buffer (what a solver
iteration needs) is timed after the headline region and reported under
"fused" (it does not contribute to `value`).

  python bench.py --gpus 1 --steps 10 --warmup 3
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
      --master-port P bench.py --gpus N --steps K --warmup W
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# the host driver only supports dmabuf IPC: without this RCCL's peer mapping fails with
# "hipIpcGetMemHandle: invalid argument" (already exported on the GPU boxes; kept for safety)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

from harness.bench_common import (HBM_PEAK_GBS, BYTES_PER_EVAL, BYTES_NO_CORR, BYTES_OUT, BYTES_POINT,  # noqa: E402
                                   BYTES_NEIGHBOURS, BYTES_PER_EVAL_FUSED, BYTES_NO_CORR_FUSED, lpt_shards, shards)
from harness import bench_common  # noqa: E402
from harness.bench_cpu import cpu_baseline  # noqa: E402
from harness.bench_tsdf import tsdf_bench, finish_bench  # noqa: E402
from harness.bench_config5 import config5_bench  # noqa: E402
from harness.bench_multi import multi_context_bench  # noqa: E402

PROFILE_TRAFFIC = {}           # profiles/hbm_traffic.json: PMC bytes per launch of the workloads below


def build_graph(args):
    """Config 3: grid of submaps over the analytic city scene, constraints between
    submaps whose cubes overlap."""
    gw, gh = args.grid
    rng = np.random.default_rng(args.seed)
    dims = np.array(args.block_dims, np.int32)
    extent = dims * 16 * args.voxel_size
    sx, sy = extent[0] * 0.5, extent[1] / 3.0       # 50 % overlap in x, 2/3 in y
    true_poses, ids = [], {}
    for j in range(gh):
        for i in range(gw):
            ids[(i, j)] = len(true_poses)
            true_poses.append([i * sx, j * sy, 0.0, rng.uniform(-0.1, 0.1)])
    true_poses = np.array(true_poses)
    pairs = []
    for j in range(gh):
        for i in range(gw):
            for di, dj in ((1, 0), (0, 1), (1, 1), (-1, 1), (0, 2), (1, 2), (-1, 2)):
                k = (i + di, j + dj)
                if k in ids:
                    pairs.append((ids[(i, j)], ids[k]))
    # initial guess = truth + drift: N(0, 0.3 m), N(0, 0.05 rad); submap 0 fixed
    poses = true_poses + np.concatenate(
        [rng.normal(0, args.pose_sigma, (len(true_poses), 3)),
         rng.normal(0, args.yaw_sigma, (len(true_poses), 1))], axis=1)
    poses[0] = true_poses[0]
    return true_poses, poses, np.array(pairs, np.int32)



# stdout belongs to the ONE line the driver parses.  Libraries write there too -- RCCL prints a five-line version banner
# on stdout when its communicator is created (seen under torch.distributed.run on the GPU box), gloo announces every
# rank -- so file descriptor 1 is pointed at stderr for the whole run and the line goes to a saved copy of the real
# stdout at the end.
_REAL_STDOUT = None


def claim_stdout():
    global _REAL_STDOUT
    if _REAL_STDOUT is None:
        sys.stdout.flush()
        _REAL_STDOUT = os.dup(1)
        os.dup2(2, 1)


def multi_context_child(args, n_devices, timeout_s=240):
    """`bench.py --gpus N --inprocess` (ONE process, N contexts, vgx_reg_multi_*) as a child process on the same graph;
    returns its `multi_context` block (or what went wrong).  Called by rank 0 of a torchrun launch while the other
    ranks wait at the barrier."""
    import subprocess
    import tempfile
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK", "ROLE_RANK", "ROLE_WORLD_SIZE",
                        "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID", "TORCHELASTIC_RESTART_COUNT",
                        "TORCHELASTIC_MAX_RESTARTS", "TORCHELASTIC_USE_AGENT_STORE", "TORCH_NCCL_ASYNC_ERROR_HANDLING")}
    with tempfile.TemporaryDirectory() as tmp:
        detail = os.path.join(tmp, "inprocess.json")
        cmd = [sys.executable, os.path.abspath(__file__), "--gpus", str(n_devices), "--inprocess", "--no-cpu-baseline", "--no-tsdf",
               "--no-solve", "--no-config5", "--no-config2", "--no-parity", "--no-full-overlap", "--no-shipped",
               "--placement-candidates", "1",      # (only the child's multi_context block is used)
               "--steps", str(max(min(args.steps, 5), 1)), "--warmup", "1", "--inner", str(args.inner),
               "--grid", *map(str, args.grid), "--block-dims", *map(str, args.block_dims), "--block-min", *map(str, args.block_min),
               "--voxel-size", str(args.voxel_size), "--truncation", str(args.truncation), "--esdf-max", str(args.esdf_max),
               "--pose-sigma", str(args.pose_sigma), "--yaw-sigma", str(args.yaw_sigma), "--seed", str(args.seed),
               "--placement", args.placement, "--detail", detail]
        try:
            r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout_s)
        except subprocess.TimeoutExpired:
            return {"error": f"the child process did not finish within {timeout_s} s", "child": True, "contexts": n_devices}
        if r.returncode != 0 or not os.path.exists(detail):
            return {"error": f"child process exit code {r.returncode}", "stderr_tail": r.stderr[-1500:], "child": True,
                    "contexts": n_devices}
        block = json.load(open(detail)).get("multi_context") or {"error": "the child wrote no multi_context block"}
    block["child"] = True
    return block


def emit(full, detail_path, full_line=False):
    """Everything measured -> `detail_path` (and stderr); ONE compact line -> stdout (harness/bench_line.py:
    contract keys first, numbers only, below bench_line.LINE_LIMIT bytes so that the driver can parse it)."""
    from harness import bench_line
    text = json.dumps(full, indent=1)
    where = None
    try:
        with open(detail_path, "w") as f:
            f.write(text + "\n")
        where = os.path.relpath(detail_path, ROOT) if os.path.abspath(detail_path).startswith(ROOT) else detail_path
    except OSError as e:                          # a read-only checkout must not cost the line
        print(f"bench.py: could not write {detail_path}: {e}", file=sys.stderr)
    print(text, file=sys.stderr)
    sys.stderr.flush()
    _, line = bench_line.compact(full, where)
    out = (json.dumps(full) if full_line else line) + "\n"
    sys.stdout.flush()
    if _REAL_STDOUT is not None:
        os.write(_REAL_STDOUT, out.encode())
    else:
        sys.stdout.write(out)
        sys.stdout.flush()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--inner", type=int, default=25,
                    help="passes over all constraints per step (the driver's 20 steps then time > 2 s)")
    ap.add_argument("--no-parity", action="store_true",
                    help="skip the same-run parity gate (harness/parity_gate.py: a sample of the timed launches' "
                         "constraints against the reference source / the oracle)")
    ap.add_argument("--overlap-copies", type=int, default=6,
                    help="full-overlap workload: perturbed duplicates per submap (config 3 has ~6 "
                         "constraints per reference submap)")
    ap.add_argument("--no-full-overlap", action="store_true")
    ap.add_argument("--placement-candidates", type=int, default=6,
                    help="allocations of each row array the materialising kernel is timed on before the timed region; the "
                         "fastest of each kind is kept (vgx_reg_batch_choose_outputs; 1 = take what the allocator gives)")
    ap.add_argument("--no-fo-plain", action="store_true",
                    help="skip the plain-launch-order measurement of the full-overlap workload (PMC passes: "
                         "one launch order per grid size)")
    ap.add_argument("--no-shipped", action="store_true",
                    help="skip the shipped-yaml evaluation (isosurface points, sampling_ratio 0.05, mirrored)")
    ap.add_argument("--grid", type=int, nargs=2, default=[20, 10], help="submap grid (200 submaps)")
    ap.add_argument("--block-dims", type=int, nargs=3, default=[16, 16, 16], help="256^3 voxels")
    ap.add_argument("--block-min", type=int, nargs=3, default=[-8, -8, -4])
    ap.add_argument("--voxel-size", type=float, default=0.2)
    ap.add_argument("--truncation", type=float, default=0.6)
    ap.add_argument("--esdf-max", type=float, default=2.0)
    ap.add_argument("--pose-sigma", type=float, default=0.3)
    ap.add_argument("--yaw-sigma", type=float, default=0.05)
    ap.add_argument("--seed", type=int, default=2)
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-fused", action="store_true")
    ap.add_argument("--no-solve", action="store_true")
    ap.add_argument("--no-tsdf", action="store_true")
    ap.add_argument("--pipeline", action="store_true",
                    help="run the config-2 stand-in at full length (30 submaps x 100 scans, harness/pipeline.py) "
                         "instead of the bounded one (10 submaps x 30 scans) the default line carries")
    ap.add_argument("--no-config2", action="store_true")
    ap.add_argument("--no-multi-ctx", action="store_true")
    ap.add_argument("--no-quad", action="store_true",
                    help="skip the shipped configuration's second measurement (apron bricks, for comparison)")
    ap.add_argument("--inprocess", action="store_true",
                    help="ONE process drives --gpus N devices through vgx_reg_multi_* (the product's in-process "
                         "multi-GPU component); launch WITHOUT torch.distributed.run.  The headline loop then "
                         "runs on device 0 alone and `multi_context` carries the N-GPU evaluation")
    ap.add_argument("--no-config5", action="store_true")
    ap.add_argument("--config", type=int, default=3, choices=[3, 5],
                    help="5: only BASELINE configs[4] (1000 submaps @ 128^3, loop closures, two-stage solve)")
    ap.add_argument("--config5-grid", type=int, nargs=2, default=[25, 40], help="lanes x submaps per lane")
    ap.add_argument("--placement", choices=["contiguous", "lpt"], default="contiguous",
                    help="how the constraint list is sharded over ranks / contexts: vgx_contiguous_shards (consecutive "
                         "runs of equal weight: better measured balance and a third of the submaps per shard, "
                         "profiles/r04_shard_balance.json) or vgx_lpt_shards.  Results do not depend on it.")
    ap.add_argument("--detail", default=os.path.join(ROOT, "bench_detail.json"),
                    help="where the FULL result object goes (every block with its notes); the stdout line is the "
                         "compact summary of harness/bench_line.py")
    ap.add_argument("--full-line", action="store_true",
                    help="print the FULL object on stdout instead of the compact line (profiles/ab_*.sh pick nested "
                         "keys from it); never what the driver runs")
    ap.add_argument("--calibrate", action="store_true",
                    help="PMC calibration: first launch evaluates poses 10 km apart, so every "
                         "evaluation reads exactly 20 B and writes exactly 36 B (profiles/README.md)")
    args = ap.parse_args()
    claim_stdout()           # (after --help had its chance to print)
    bench_common.PLACEMENT = args.placement

    lib_path = os.path.join(ROOT, "voxgraph_amd", "lib", "libvoxgraph_amd.so")
    if not os.path.exists(lib_path):            # checkout without the (git-ignored) built library
        if int(os.environ.get("LOCAL_RANK", "0")) == 0:
            import __graft_entry__
            __graft_entry__.build()
        else:
            t_wait = time.time()
            while not os.path.exists(lib_path) and time.time() - t_wait < 900:
                time.sleep(1.0)
            time.sleep(2.0)                      # let the linker finish writing
    from voxgraph_amd import capi
    capi.load()
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and not (args.inprocess and world == 1):
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run "
                         "(or pass --inprocess for the one-process multi-GPU component)")
    # VGX_BENCH_DRYRUN=gloo: every rank on cuda:0 with the gloo backend, to walk the N>1
    # code path on a one-GPU box (profiles/README.md); never used for reported numbers
    dryrun = os.environ.get("VGX_BENCH_DRYRUN", "")
    if dryrun:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    # under torch.distributed.run (RANK set) the RCCL group is always created, also for
    # one rank, so the collective code path below is the one that runs at every N
    use_dist = world > 1 or "RANK" in os.environ
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if dryrun:
            dist.init_process_group(dryrun)
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    def barrier():
        if use_dist:
            dist.barrier()

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import pyoracle
        pyoracle.use_native_build()      # the CPU port as BASELINE.md times it: -O3 -march=native, built on this host
    ctx = capi.Context(local_rank)
    # one explicit (non-null) stream shared by the HIP library, torch ops and RCCL,
    # so kernel -> all-reduce ordering is by stream order, not by host syncs
    stream = torch.cuda.Stream()
    torch.cuda.set_stream(stream)
    ctx.set_stream(stream.cuda_stream)

    if args.config == 5:
        c5 = config5_bench(capi, ctx, torch, dist, use_dist, rank, world, args)
        if rank == 0:
            emit({"metric": "full pose-graph solve ms (1000 submaps, two-stage)", "value": c5["solve_ms"],
                  "unit": "ms", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": c5["solve_ms"],
                  "higher_is_better": False, "scaling": "strong",
                  "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                  "config": {"workload": c5["workload"], "submaps": c5["submaps"],
                             "constraints": c5["registration_constraints"], "parallelism": c5["parallelism"]},
                  "config5": c5, "parity": c5.get("parity")}, args.detail, args.full_line)
        if use_dist:
            dist.barrier()
            dist.destroy_process_group()
        return

    true_poses, poses, pairs = build_graph(args)
    n_sub, n_con = len(true_poses), len(pairs)
    global PROFILE_TRAFFIC
    tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    PROFILE_TRAFFIC = json.load(open(tpath)) if os.path.exists(tpath) else {}

    # ---- resident inputs: every rank holds every finished submap -------------
    t_setup = time.perf_counter()
    submaps, n_points, n_iso = [], [], []
    for k in range(n_sub):
        sm = capi.Submap.synth_city(ctx, k, args.voxel_size, 16, args.block_min, args.block_dims,
                                    args.truncation, args.esdf_max, 10.0, true_poses[k], args.seed)
        n_points.append(sm.extract_voxel_points(1.0, 0.3, True))      # voxgraph_submap.h:27-28
        if not args.no_shipped:
            n_iso.append(sm.extract_isosurface_points(1.0))            # finishSubmap(), voxgraph_submap.cpp:97
        sm.release_raw_layers()
        submaps.append(sm)
    ctx.synchronize()
    setup_s = time.perf_counter() - t_setup

    cfg = capi.default_config(registration_point_type=capi.POINTS_VOXELS)
    weights = np.array([n_points[a] for a, _ in pairs], np.int64)
    if world > 1 or not args.no_multi_ctx:
        # what a constraint costs is the bytes it moves at these poses, not its residual count: 36 B per
        # row written + ~45 B per residual in a chunk that can touch the reading submap
        # (vgx_reg_batch_count_live_each; profiles/shard_balance.py: slowest-shard balance at N = 8
        # 0.980 -> 0.991).  Every rank computes the same weights from the same replicated inputs.
        cfs_all = [capi.RegistrationCostFunction(ctx, submaps[a], submaps[b], cfg) for a, b in pairs]
        probe = capi.RegistrationBatch(ctx, cfs_all, pairs)
        weights = 36 * weights + 45 * probe.count_live_each(poses)
        probe.destroy()
        for cf in cfs_all:
            cf.destroy()
    weights_bytes = weights
    mine = shards(weights, world)[rank]
    cfs = [capi.RegistrationCostFunction(ctx, submaps[pairs[c][0]], submaps[pairs[c][1]], cfg)
           for c in mine]
    batch = capi.RegistrationBatch(ctx, cfs, pairs[mine], global_index=mine, n_global=n_con)
    R = batch.num_residuals()

    # ---- second workload: ~100 % correspondence (gather-dominated regime) ---------
    # submap k against K perturbed duplicates of itself; node n_sub * (1 + p) + k carries the
    # p-th perturbed pose of submap k.  Order (p, k): consecutive constraints share nothing.
    fo = None
    if not args.no_full_overlap:
        K = args.overlap_copies
        rng_fo = np.random.default_rng(args.seed + 100)
        poses_fo = np.concatenate([true_poses] + [
            true_poses + np.concatenate([rng_fo.normal(0, args.pose_sigma, (n_sub, 3)),
                                         rng_fo.normal(0, args.yaw_sigma, (n_sub, 1))], axis=1)
            for _ in range(K)])
        pairs_fo = np.array([(k, n_sub * (1 + p) + k) for p in range(K) for k in range(n_sub)], np.int32)
        mine_fo = shards([n_points[a] for a, _ in pairs_fo], world)[rank]
        cfs_fo = [capi.RegistrationCostFunction(ctx, submaps[pairs_fo[c][0]], submaps[pairs_fo[c][0]], cfg)
                  for c in mine_fo]
        batch_fo = capi.RegistrationBatch(ctx, cfs_fo, pairs_fo[mine_fo], global_index=mine_fo,
                                          n_global=len(pairs_fo))
        fo = {"batch": batch_fo, "poses": poses_fo, "R": batch_fo.num_residuals(), "n": len(pairs_fo)}

    R_buf = max(R, fo["R"] if fo else 0)
    # Where the three row arrays lie in physical memory decides which of two speeds the materialising kernel runs at
    # (4.4-4.7 or 5.3-5.7 ms on this workload; half of all sets of three are slow ones: profiles/r05_points_placement.txt, r05_headline_ab.txt
    # addendum 4).  The library's answer is placement by measurement (vgx_reg_batch_choose_outputs): several
    # candidate allocations, the batch's own launch timed on them, the best array of each kind kept, the rest freed --
    # here, before anything is timed, as a caller that owns its row buffers would do once per batch.
    free_b, _total_b = torch.cuda.mem_get_info()
    n_cand = int(max(1, min(args.placement_candidates, (0.55 * free_b) // (36 * R_buf + (64 << 20)))))
    cand = [(torch.empty(R_buf, dtype=torch.float32, device="cuda"), torch.empty((R_buf, 4), dtype=torch.float32, device="cuda"),
             torch.empty((R_buf, 4), dtype=torch.float32, device="cuda")) for _ in range(n_cand)]
    torch.cuda.synchronize()
    placement = {"candidates": n_cand, "chosen": [0, 0, 0], "ms_chosen": None, "ms_trials": None}
    if n_cand > 1 and R > 0:
        try:
            chosen, ms_chosen, ms_trials = batch.choose_outputs(poses, [c_[0].data_ptr() for c_ in cand], [c_[1].data_ptr() for c_ in cand],
                                                                [c_[2].data_ptr() for c_ in cand], launches=3)
            placement.update(chosen=chosen, ms_chosen=ms_chosen, ms_trials=ms_trials, ms_sets=ms_trials[:n_cand],
                             what="vgx_reg_batch_choose_outputs: sets first, then jac_read / jac_ref / residuals array by array "
                                  "(-1: no new trial needed); ms per launch, 3 launches per trial")
        except Exception as e:   # noqa: BLE001  (the selection is an optimisation: without it the first set is used, and the line says so)
            placement.update(candidates=1, error=repr(e)[:300])
            print(f"bench.py: placement selection failed, using the first allocation: {e!r}", file=sys.stderr)
    residuals, jac_ref, jac_read = cand[placement["chosen"][0]][0], cand[placement["chosen"][1]][1], cand[placement["chosen"][2]][2]
    del cand
    torch.cuda.empty_cache()

    def one_pass():
        batch.evaluate_points(poses, residuals.data_ptr(), jac_ref.data_ptr(), jac_read.data_ptr())

    def step():
        for _ in range(args.inner):
            one_pass()

    # ---- which box is this?  clocks / power cap / partition modes, and what its memory system sustains for a float4
    # copy and for a non-temporal fill of the very row buffers the timed kernel writes
    from harness import box_state
    box = box_state.snapshot(local_rank) if rank == 0 else None
    n16 = (R * 16) & ~15

    def ceilings():
        copy_ms = capi.stream_ceiling_ms(ctx, jac_ref.data_ptr(), n16, jac_read.data_ptr(), n16, 3)
        fill_ms = capi.stream_ceiling_ms(ctx, None, 0, jac_read.data_ptr(), n16, 3)
        return {"copy_GBs": 2 * n16 / (copy_ms * 1e-3) / 1e9, "fill_GBs": n16 / (fill_ms * 1e-3) / 1e9,
                "copy_ms": copy_ms, "fill_ms": fill_ms, "bytes_each_way": n16}
    ceil_before = ceilings()

    if args.calibrate:
        far = poses.copy()
        far[:, 0] += 1e4 * np.arange(n_sub)
        # twice: the first dispatch of a process has been seen to under-count FETCH_SIZE
        # (profiles/README.md); summarize.py calibrates on the second
        for _ in range(2):
            batch.evaluate_points(far, residuals.data_ptr(), jac_ref.data_ptr(), jac_read.data_ptr())
            torch.cuda.synchronize()
        assert int((jac_ref[:R].abs().sum(dim=1) > 0).sum().item()) == 0
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    barrier()
    sampler = box_state.Sampler(local_rank if rank == 0 else 0)
    if rank == 0:
        sampler.__enter__()                # a host thread reading sysfs clocks / power every 20 ms; touches no queue
    ctx.timer_start()                      # HIP events on the stream the kernel runs on
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    kernel_ms = ctx.timer_stop() / max(args.steps * args.inner, 1)
    torch.cuda.synchronize()
    barrier()
    dt = time.perf_counter() - t0
    if rank == 0:
        sampler.__exit__()
    tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
    rtot = torch.tensor([float(R)], dtype=torch.float64, device="cuda")
    kmax = torch.tensor([kernel_ms], dtype=torch.float64, device="cuda")
    if use_dist:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(rtot, op=dist.ReduceOp.SUM)
        dist.all_reduce(kmax, op=dist.ReduceOp.MAX)
    dt, total_evals, kernel_ms_max = float(tmax.item()), float(rtot.item()), float(kmax.item())

    # correspondences on this rank (for the conservative byte count)
    with_corr = int((jac_ref[:R].abs().sum(dim=1) > 0).sum().item())
    checksum = float(residuals[:R].double().pow(2).sum().item())
    wc_tot = torch.tensor([float(with_corr)], dtype=torch.float64, device="cuda")
    if use_dist:
        dist.all_reduce(wc_tot, op=dist.ReduceOp.SUM)
    with_corr_total = float(wc_tot.item())

    # ---- parity gate, config 3: rows of the LAST TIMED PASS (still in the output buffers) ----
    parity_entries = []

    def layers_of_config3(k):
        sm = capi.Submap.synth_city(ctx, k, args.voxel_size, 16, args.block_min, args.block_dims,
                                    args.truncation, args.esdf_max, 10.0, true_poses[k], args.seed)
        td, tw, ed, eo = sm.download_layers(16)
        bi = sm.block_index()
        sm.destroy()
        return bi, td, tw, ed, eo
    if rank == 0 and not args.no_parity:
        from harness import parity_gate
        t_par = time.perf_counter()
        live_each = batch.count_live_each(poses)
        chosen = parity_gate.choose(np.diff(batch.row_offsets()), live_each, n_total=8, n_partial=3, n_dead=1)
        e = parity_gate.check(capi, ctx, torch, "config 3 (rows of the last timed pass)", layers_of_config3, submaps, batch,
                              pairs[mine], poses, chosen, (residuals, jac_ref, jac_read), args.voxel_size)
        e["partly_culled_constraints"] = int(sum(1 for c in chosen if 0 < live_each[c] < 0.9 * np.diff(batch.row_offsets())[c]))
        e["seconds"] = time.perf_counter() - t_par
        parity_entries.append(e)

    # points the materialising kernel reads at these poses: tiles whose chunks all miss the reading
    # submap's block box write zeros without reading (vgx_reg.hip reg_eval_points_body); where the
    # launch order groups the constraints of a reference submap on one XCD they read its points once
    def points_read(bt, ps):
        lv, uq = bt.count_live(ps, unique=True)
        grouped = bt.launch_order(points_pass=True)
        return {"live": int(lv), "distinct": int(uq), "grouped": int(grouped),
                "read": int(uq if grouped == 1 else lv)}
    pr = points_read(batch, poses)
    # the same two ceilings again after the timed region (a box that heats up or is throttled shows here), and a
    # launch of the headline kernel's own shape: its algorithmic read : write proportion, non-temporal stores
    # (the output buffers hold nothing any more that a later section reads: the parity gate above is done)
    ceil_after = ceilings()
    rd = ((pr["read"] * BYTES_POINT + with_corr * BYTES_NEIGHBOURS) * n16 // max(R * BYTES_OUT, 1)) & ~15
    rd = min(rd, n16)
    mix_ms = capi.stream_ceiling_ms(ctx, jac_ref.data_ptr(), rd, jac_read.data_ptr(), n16, 3)
    ceil_after["kernel_shaped_GBs"] = (rd + n16) / (mix_ms * 1e-3) / 1e9
    ceil_after["kernel_shaped_read_fraction"] = rd / max(rd + n16, 1)
    # the placement-proof entry point on the same workload, same run: ONE array of tile blocks (one write front,
    # vgx_reg_batch_evaluate_points_blocked) -- what a device-resident consumer gets without choosing anything
    blocked_ms = None
    if R > 0:
        try:
            nbytes_blk, _rows, _first = batch.blocked_layout()
            blocks = torch.empty(nbytes_blk // 4, dtype=torch.float32, device="cuda")
            for _ in range(2):
                batch.evaluate_points_blocked(poses, blocks.data_ptr())
            torch.cuda.synchronize()
            ctx.timer_start()
            for _ in range(10):
                batch.evaluate_points_blocked(poses, blocks.data_ptr())
            blocked_ms = ctx.timer_stop() / 10
            del blocks
            torch.cuda.empty_cache()
        except Exception as e:   # noqa: BLE001
            print(f"bench.py: blocked-layout timing skipped: {e!r}", file=sys.stderr)
    # the same pass in Ceres' own types (vgx_reg_batch_evaluate_points_f64: f64 rows, 72 B written per row -- SURVEY.md
    # 8d's "124 B" variant, reported alongside, never instead); its f32 rounding must be the timed pass's rows
    f64_rows_ms, f64_rows_match = None, None
    if R > 0 and world == 1:
        try:
            r64 = torch.empty(R, dtype=torch.float64, device="cuda")
            jo64 = torch.empty((R, 4), dtype=torch.float64, device="cuda")
            je64 = torch.empty((R, 4), dtype=torch.float64, device="cuda")
            for _ in range(2):
                batch.evaluate_points_f64(poses, r64.data_ptr(), jo64.data_ptr(), je64.data_ptr())
            torch.cuda.synchronize()
            ctx.timer_start()
            for _ in range(10):
                batch.evaluate_points_f64(poses, r64.data_ptr(), jo64.data_ptr(), je64.data_ptr())
            f64_rows_ms = ctx.timer_stop() / 10
            # (the ceiling launches above wrote over the f32 arrays: the f32 pass once more, then the comparison)
            batch.evaluate_points(poses, residuals.data_ptr(), jac_ref.data_ptr(), jac_read.data_ptr())
            torch.cuda.synchronize()
            f64_rows_match = bool(torch.equal(r64.float().view(torch.int32), residuals[:R].view(torch.int32))
                                  and torch.equal(jo64.float().view(torch.int32), jac_ref[:R].view(torch.int32))
                                  and torch.equal(je64.float().view(torch.int32), jac_read[:R].view(torch.int32)))
            del r64, jo64, je64
            torch.cuda.empty_cache()
        except Exception as e:   # noqa: BLE001
            print(f"bench.py: f64-rows timing skipped: {e!r}", file=sys.stderr)

    # ---- full-overlap workload, timed the same way (HIP events on the kernel's stream) ----
    fo_out = None
    if fo:
        def time_fo(bt):
            def fo_pass():
                bt.evaluate_points(fo["poses"], residuals.data_ptr(), jac_ref.data_ptr(), jac_read.data_ptr())
            n_fo = max(args.steps, 1)
            for _ in range(2):
                fo_pass()
            torch.cuda.synchronize()
            barrier()
            ctx.timer_start()
            f0 = time.perf_counter()
            for _ in range(n_fo):
                fo_pass()
            k_ms = ctx.timer_stop() / n_fo
            torch.cuda.synchronize()
            barrier()
            return n_fo, time.perf_counter() - f0, k_ms
        n_fo, fo_dt, fo_kernel_ms = time_fo(fo["batch"])
        fo_corr = int((jac_ref[:fo["R"]].abs().sum(dim=1) > 0).sum().item())
        fo_pr = points_read(fo["batch"], fo["poses"])
        # the same constraints in plain constraint-major launch order (nothing shared through the L2:
        # the regime the 88 B-per-evaluation contract figure describes); the order is chosen at a
        # batch's first evaluation, VGX_POINTS_TILE_ORDER is read then
        fo_plain_ms = None
        if not args.no_fo_plain:
            prev = os.environ.get("VGX_POINTS_TILE_ORDER")
            os.environ["VGX_POINTS_TILE_ORDER"] = "0"
            plain = capi.RegistrationBatch(ctx, cfs_fo, pairs_fo[mine_fo], global_index=mine_fo, n_global=len(pairs_fo))
            _, _, fo_plain_ms = time_fo(plain)
            assert plain.launch_order(points_pass=True) == 0
            plain.destroy()
            if prev is None:
                del os.environ["VGX_POINTS_TILE_ORDER"]
            else:
                os.environ["VGX_POINTS_TILE_ORDER"] = prev
        red = torch.tensor([fo_dt, fo_kernel_ms, fo_plain_ms or 0.0], dtype=torch.float64, device="cuda")
        tot = torch.tensor([float(fo["R"]), float(fo_corr)], dtype=torch.float64, device="cuda")
        if use_dist:
            dist.all_reduce(red, op=dist.ReduceOp.MAX)
            dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        fo_out = {"dt": float(red[0].item()), "kernel_ms_max": float(red[1].item()), "kernel_ms": fo_kernel_ms,
                  "plain_kernel_ms": fo_plain_ms, "plain_kernel_ms_max": float(red[2].item()) if fo_plain_ms else None,
                  "passes": n_fo, "R": fo["R"], "R_total": float(tot[0].item()), "with_corr": fo_corr,
                  "with_corr_total": float(tot[1].item()), "points": fo_pr}

    if fo and rank == 0 and not args.no_parity:
        t_par = time.perf_counter()
        fo["batch"].evaluate_points(fo["poses"], residuals.data_ptr(), jac_ref.data_ptr(), jac_read.data_ptr())
        torch.cuda.synchronize()
        n_fo_local = len(mine_fo)
        chosen = [int(c) for c in np.unique(np.linspace(0, n_fo_local - 1, 4).round().astype(int))]
        e = parity_gate.check(capi, ctx, torch, "full overlap (one more launch of the timed batch)", layers_of_config3,
                              submaps, fo["batch"], pairs_fo[mine_fo], fo["poses"], chosen,
                              (residuals, jac_ref, jac_read), args.voxel_size, submap_of_node=lambda nd: nd % n_sub)
        e["seconds"] = time.perf_counter() - t_par
        parity_entries.append(e)

    # ---- fused pass + all-reduce (solver-iteration form), reported separately --
    def fused_bench(bt, ps, n_nodes, n_global, corr_total, evals_total, ref_cost, traffic_key, pairs_global):
        size = capi.fused_size(n_nodes, n_global)
        tr = (PROFILE_TRAFFIC.get("fused") or {}).get(traffic_key) or {}
        tr_bytes = tr.get("hbm_bytes_per_launch") if (tr.get("evaluations") == evals_total and world == 1) else None
        buf = torch.zeros(size, dtype=torch.float64, device="cuda")
        # N ranks: ONE all-reduce of the [n_global][45] array of per-constraint blocks (a row is written by exactly
        # one rank: the sum is exact in any order), then every rank assembles the fused buffer in list order --
        # the single-GPU buffer bit for bit at every N (include/voxgraph_amd.h "Sharding-independent assembly")
        blocks = torch.zeros((n_global, capi.NORMAL_SIZE), dtype=torch.float64, device="cuda") if use_dist else None
        assembler = capi.RegistrationAssembler(ctx, pairs_global) if use_dist else None
        torch.cuda.synchronize()

        def fused_step():
            bt.evaluate_normal(ps, to_host=False)
            if use_dist:
                bt.scatter_normal(blocks.data_ptr(), zero_first=True)
                dist.all_reduce(blocks.view(torch.int64))      # integer sum of disjoint rows: the bit patterns, in any order
                assembler.assemble(blocks.data_ptr(), n_nodes, buf.data_ptr())
            else:
                bt.assemble(n_nodes, buf.data_ptr(), zero_first=True)

        for _ in range(2):
            fused_step()
        torch.cuda.synchronize()
        barrier()
        n_f = max(args.steps, 1)
        ctx.timer_start()
        f0 = time.perf_counter()
        for _ in range(n_f):
            fused_step()
        f_kernel_ms = ctx.timer_stop() / n_f           # evaluate + finalize + assemble (+ all-reduce)
        torch.cuda.synchronize()
        barrier()
        fdt = torch.tensor([time.perf_counter() - f0], dtype=torch.float64, device="cuda")
        # points the kernel actually loads: chunks whose bounding sphere cannot touch the reading
        # grid are skipped without reading them (vgx_reg_batch_count_live); constraints that share a
        # reference submap load the SAME points, which the launch order lets them share in one L2
        lv, uq = bt.count_live(ps, unique=True)
        live = torch.tensor([float(lv), float(uq)], dtype=torch.float64, device="cuda")
        if use_dist:
            dist.all_reduce(fdt, op=dist.ReduceOp.MAX)
            dist.all_reduce(live, op=dist.ReduceOp.SUM)
        fdt, live, unique_pts = float(fdt.item()), float(live[0].item()), float(live[1].item())
        # algorithmic bytes: every DISTINCT loaded point once (20 B) + the 8 neighbours of every
        # evaluation that interpolates (32 B); the per-evaluation pricing of SURVEY.md 8d (52 B / 20 B /
        # 0 B), which re-counts a point for every constraint that reads it, is reported beside it
        alg_bytes = unique_pts * BYTES_NO_CORR_FUSED + corr_total * (BYTES_PER_EVAL_FUSED - BYTES_NO_CORR_FUSED)
        per_eval_bytes = corr_total * BYTES_PER_EVAL_FUSED + max(live - corr_total, 0.0) * BYTES_NO_CORR_FUSED
        out = {"value": evals_total * n_f / fdt / 1e6, "unit": "Mresiduals+Jacobians/s",
               "ms_per_step": fdt / n_f * 1e3, "stream_ms_per_step": f_kernel_ms,
               "kernels": "reg_eval_reduce_lean_kernel + reg_finalize_kernel + reg_assemble_kernel"
                          + (" + RCCL all-reduce" if use_dist else ""),
               "evaluations": evals_total, "with_correspondence": corr_total, "loaded_after_culling": live,
               "distinct_points_loaded": unique_pts,
               "algorithmic_bytes_per_step": alg_bytes,
               "per_evaluation_pricing_bytes_per_step": per_eval_bytes,
               "per_evaluation_pricing_GBs": per_eval_bytes * n_f / fdt / 1e9,
               "pricing": "algorithmic = 20 B x distinct loaded points + 32 B x interpolating evaluations; "
                          "per_evaluation_pricing re-counts a point for every constraint that reads it (52 / 20 / 0 B) "
                          "and can exceed the HBM peak where constraints share points through the L2",
               "algorithmic_GBs": alg_bytes * n_f / fdt / 1e9,
               # NOT an HBM fraction: algorithmic bytes include neighbour bytes the L2 / MALL serve
               "algorithmic_over_hbm_peak": alg_bytes * n_f / fdt / 1e9 / HBM_PEAK_GBS,
               # counter bytes (profiles/hbm_traffic.json, a separate rocprofv3 --pmc run of this workload)
               # / this run's time / 8 TB/s: the HBM fraction proper
               "traffic_from_profiles": tr_bytes,
               "hbm_frac": (tr_bytes / (f_kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if tr_bytes else None,
               "value_with_correspondence": corr_total * n_f / fdt / 1e6,
               "allreduce_bytes": int(n_global * capi.NORMAL_SIZE * 8) if use_dist else 0,
               "cost": float(buf[0].item()), "cost_vs_materialised": None,
               "fused_sha256": __import__("hashlib").sha256(buf.cpu().numpy().tobytes()).hexdigest()}
        if assembler:
            assembler.destroy()
        if world == 1 and ref_cost is not None:
            out["cost_vs_materialised"] = abs(out["cost"] - ref_cost) / max(ref_cost, 1e-30)
        # the cost-only pass (vgx_reg_batch_evaluate_cost): what Ceres asks for at every trial step (`jacobians == nullptr`,
        # registration_cost_function.cpp:179) -- the same tiles, one running sum per lane; its costs must be the full
        # pass's costs bit for bit
        d_cost = torch.zeros(max(bt.n, 1), dtype=torch.float64, device="cuda")
        for _ in range(2):
            bt.evaluate_cost(ps, d_cost=d_cost.data_ptr(), to_host=False)
        torch.cuda.synchronize()
        ctx.timer_start()
        for _ in range(n_f):
            bt.evaluate_cost(ps, d_cost=d_cost.data_ptr(), to_host=False)
        out["cost_only_ms"] = ctx.timer_stop() / n_f
        _, normal = bt.evaluate_normal(ps)
        out["cost_only_equals_full_pass_cost"] = bool(np.array_equal(d_cost.cpu().numpy()[:bt.n].view(np.uint64),
                                                                     normal[:, 0].copy().view(np.uint64)))
        out["cost_only_kernels"] = "reg_eval_reduce_lean_kernel<cost only> + reg_finalize_cost_kernel"
        return out

    fused = None
    fused_fo = None
    if not args.no_fused:
        fused = fused_bench(batch, poses, n_sub, n_con, with_corr_total, total_evals, checksum, "config3", pairs)
        if fo:
            fo_cost = float(residuals[:fo["R"]].double().pow(2).sum().item()) if world == 1 else None
            fused_fo = fused_bench(fo["batch"], fo["poses"], len(fo["poses"]), fo["n"],
                                   fo_out["with_corr_total"], fo_out["R_total"], fo_cost, "full_overlap", pairs_fo)

    # ---- the reference's SHIPPED configuration (voxgraph_mapper.yaml:34-35): explicit_to_implicit =
    # isosurface points, sampling_ratio 0.05, both directions of every pair (pose_graph.cpp:62-71),
    # ESDF distance (registration_cost_function.h:35) -- one solver evaluation = one fused batched pass;
    # the per-point-set std::mt19937 streams are generated on the device (mt_generate_kernel)
    shipped = None
    if not args.no_shipped and not args.no_fused:
        cfg_s = capi.default_config(registration_point_type=capi.POINTS_ISOSURFACE, sampling_ratio=0.05)
        shard = shards([n_iso[a] + n_iso[b] for a, b in pairs], world)[rank]
        pairs_s = [(int(a), int(b)) for c in shard for a, b in (pairs[c], pairs[c][::-1])]
        gidx_s = [2 * c + k for c in shard for k in (0, 1)]
        buf_s = torch.zeros(capi.fused_size(n_sub, 2 * n_con), dtype=torch.float64, device="cuda")
        pairs_s_global = [(int(a), int(b)) for c in range(n_con) for a, b in (pairs[c], pairs[c][::-1])]
        blocks_s = torch.zeros((2 * n_con, capi.NORMAL_SIZE), dtype=torch.float64, device="cuda") if use_dist else None

        def shipped_eval(ctx_s, submaps_s):
            cfs_s = [capi.RegistrationCostFunction(ctx_s, submaps_s[a], submaps_s[b], cfg_s) for a, b in pairs_s]
            batch_s = capi.RegistrationBatch(ctx_s, cfs_s, pairs_s, global_index=gidx_s, n_global=2 * n_con)

            asm_s = capi.RegistrationAssembler(ctx_s, pairs_s_global) if use_dist else None

            def shipped_step():
                batch_s.evaluate_normal(poses, to_host=False)
                if use_dist:
                    batch_s.scatter_normal(blocks_s.data_ptr(), zero_first=True)
                    dist.all_reduce(blocks_s.view(torch.int64))
                    asm_s.assemble(blocks_s.data_ptr(), n_sub, buf_s.data_ptr())
                else:
                    batch_s.assemble(n_sub, buf_s.data_ptr(), zero_first=True)

            for _ in range(2):
                shipped_step()
            torch.cuda.synchronize()
            barrier()
            n_s = max(args.steps, 1)
            ctx_s.timer_start()
            s0 = time.perf_counter()
            for _ in range(n_s):
                shipped_step()
            stream_ms = ctx_s.timer_stop() / n_s
            torch.cuda.synchronize()
            barrier()
            sdt = torch.tensor([time.perf_counter() - s0], dtype=torch.float64, device="cuda")
            rs = torch.tensor([float(batch_s.num_residuals())], dtype=torch.float64, device="cuda")
            if use_dist:
                dist.all_reduce(sdt, op=dist.ReduceOp.MAX)
                dist.all_reduce(rs, op=dist.ReduceOp.SUM)
            e = {"residuals_per_evaluation": float(rs.item()),
                 "ms_per_evaluation": float(sdt.item()) / n_s * 1e3,
                 "stream_ms_per_evaluation": stream_ms,
                 "Mresiduals_per_s": float(rs.item()) * n_s / float(sdt.item()) / 1e6,
                 "cost": float(buf_s[0].item())}
            for o in [batch_s] + cfs_s + ([asm_s] if asm_s else []):
                o.destroy()
            return e
        shipped = {"config": "registration_method explicit_to_implicit (isosurface points), sampling_ratio 0.05, "
                             "mirrored constraints, ESDF distance (voxgraph_mapper.yaml:34-35, pose_graph.cpp:62-71)",
                   "constraints": 2 * n_con, "isosurface_points_per_submap": float(np.mean(n_iso)),
                   "brick_layout_chosen": "quad, made on demand from the headline's apron submaps "
                                          "(VGX_SAMPLING_BRICKS_QUAD, the default: every constraint of the batch samples)",
                   "what": "one solver evaluation: device mt19937 streams + fused normal equations of every "
                           "constraint + assembly" + (" + RCCL all-reduce" if use_dist else "")}
        shipped.update(shipped_eval(ctx, submaps))
        tr = (PROFILE_TRAFFIC.get("fused") or {}).get("shipped") or {}
        tr_bytes = tr.get("hbm_bytes_per_launch") if (tr.get("evaluations") == shipped["residuals_per_evaluation"]
                                                     and world == 1) else None
        shipped["traffic_from_profiles"] = tr_bytes
        shipped["fused_kernel_ms_from_profiles"] = tr.get("avg_ms_rocprof") if tr_bytes else None
        shipped["hbm_frac"] = (tr_bytes / (tr["avg_ms_rocprof"] * 1e-3) / 1e9 / HBM_PEAK_GBS) \
            if (tr_bytes and tr.get("avg_ms_rocprof")) else None
        shipped["hbm_frac_note"] = "fused kernel alone: counter bytes / its rocprofv3 average duration / 8 TB/s (profiles/)"
        if tr_bytes:
            shipped["traffic_over_algorithmic"] = tr_bytes / (52.0 * shipped["residuals_per_evaluation"])
        if not args.no_quad:
            # for comparison: the same evaluation on the bricks everything else reads (a context set to
            # VGX_SAMPLING_BRICKS_SAME: apron bricks); fresh default-seeded engines, so the same draws
            ctx_a = capi.Context(local_rank)
            ctx_a.set_stream(stream.cuda_stream)
            ctx_a.set_sampling_bricks(capi.SAMPLING_BRICKS_SAME)
            subs_a = []
            for k in range(n_sub):
                sm = capi.Submap.synth_city(ctx_a, k, args.voxel_size, 16, args.block_min, args.block_dims,
                                            args.truncation, args.esdf_max, 10.0, true_poses[k], args.seed)
                sm.extract_isosurface_points(1.0)
                sm.release_raw_layers()
                subs_a.append(sm)
            q = shipped_eval(ctx_a, subs_a)
            q["brick_layout"] = "apron (vgx_ctx_set_sampling_bricks(VGX_SAMPLING_BRICKS_SAME))"
            trq = (PROFILE_TRAFFIC.get("fused") or {}).get("shipped_apron") or {}
            okq = trq.get("evaluations") == q["residuals_per_evaluation"] and world == 1 and trq.get("avg_ms_rocprof")
            q["traffic_from_profiles"] = trq.get("hbm_bytes_per_launch") if okq else None
            q["hbm_frac"] = (trq["hbm_bytes_per_launch"] / (trq["avg_ms_rocprof"] * 1e-3) / 1e9 / HBM_PEAK_GBS) if okq else None
            q["cost_equals_default"] = bool(q["cost"] == shipped["cost"])
            shipped["apron_bricks"] = q
            for sm in subs_a:
                sm.destroy()
            ctx_a.close()

    # ---- the in-process multi-GPU component (vgx_reg_multi_*: one process, one vgx_ctx + host thread per
    # GPU, fixed-order sum over xGMI peer mappings) -- the PRODUCT's multi-GPU path (voxgraph is one process).
    #   N = 1 (default)          : two contexts on this one GPU: what the threads, events and the sum cost
    #   --inprocess --gpus N     : one process, contexts on devices 0..N-1
    #   N > 1 under torchrun     : after the per-rank measurements every rank waits at a barrier while
    #                              rank 0 drives all N GPUs through the component, so the driver's
    #                              scaling runs time it beside the one-process-per-GPU RCCL route
    multi_ctx = None
    if not args.no_fused and not args.no_multi_ctx:
        barrier()
        if rank == 0:
            if world > 1:
                devices = [0] * world if dryrun else list(range(world))
            elif args.inprocess:
                devices = [0] * args.gpus if dryrun else list(range(args.gpus))
            else:
                devices = [local_rank, local_rank]
            try:
                if world > 1:
                    # Under torchrun the component runs in a CHILD process (this same script, --inprocess): peer access
                    # and cross-device events over N real devices have never run on hardware, and a fault there must
                    # cost this optional block, not the line the other ranks are waiting to contribute to.
                    multi_ctx = multi_context_child(args, world)
                else:
                    multi_ctx = multi_context_bench(capi, ctx, torch, args, devices, submaps, true_poses, pairs, weights_bytes,
                                                    poses, cfg, batch if world == 1 else None, n_sub, n_con)
            except Exception as e:       # an optional section must never cost the line (peer access, memory, ...)
                multi_ctx = {"error": repr(e), "device_ids": devices}
            finally:
                # the library selects each context's device for its calls; PyTorch must find the rank's own
                # device current again (its allocator and the RCCL communicator live there)
                torch.cuda.set_device(local_rank)
        barrier()

    # ---- metric 2: full pose-graph solve (harness LM, stand-in for ceres::Solve) ---
    solve = None
    if not args.no_solve:
        from harness import lm
        from harness.backends import GpuBackend
        info = [1.0, 1.0, 2500.0, 2500.0]                       # voxgraph_mapper.yaml:41-47
        edges = [lm.RelativePoseEdge.from_poses(k, k + 1, poses[k], poses[k + 1], info)
                 for k in range(n_sub - 1)]
        backend = GpuBackend(capi, ctx, batch, n_sub, dist if use_dist else None, node_pair_global=pairs)
        backend(poses)                                           # warm

        def rmse(p):
            # gauge: submap 0 is fixed at its true pose
            return float(np.sqrt(((p[:, :3] - true_poses[:, :3]) ** 2).sum(1).mean()))

        def timed_solve(**kw):
            # one untimed solve first (imports, index tables, allocator warm-up), like
            # the warm-up steps of the headline loop
            lm.solve(lm.Problem(backend, n_sub, pairs, edges), poses, max_seconds=1e9, **kw)
            torch.cuda.synchronize()
            barrier()
            s0 = time.perf_counter()
            prob = lm.Problem(backend, n_sub, pairs, edges)
            # no wall-clock stop rule here: every rank must take the same number of
            # evaluations (each one is a collective)
            x, summ = lm.solve(prob, poses, max_seconds=1e9, **kw)
            torch.cuda.synchronize()
            barrier()
            sdt = torch.tensor([time.perf_counter() - s0], dtype=torch.float64, device="cuda")
            if use_dist:
                dist.all_reduce(sdt, op=dist.ReduceOp.MAX)
            return {"ms": float(sdt.item()) * 1e3, "iterations": summ["iterations"],
                    "gpu_evaluation_ms": summ["backend_seconds"] * 1e3,
                    "host_linear_algebra_ms": summ["host_linear_algebra_seconds"] * 1e3,
                    "evaluations": summ["evaluations"], "termination": summ["termination"],
                    "initial_cost": summ["initial_cost"], "final_cost": summ["final_cost"],
                    "position_rmse_m_before": rmse(poses), "position_rmse_m_after": rmse(x)}

        # (a) to convergence: Ceres' default function_tolerance 1e-6 decides
        solve = timed_solve(parameter_tolerance=1e-10)
        solve["stop_rule"] = "function_tolerance 1e-6 (Ceres default), parameter_tolerance off"
        # (b) the reference's exact rule: Ceres stops when |step| <= 3e-3 (|x| + 3e-3)
        #     (pose_graph.cpp:93); |x| is ~4 km here, so it fires on the first step
        solve["reference_stop_rule"] = timed_solve(parameter_tolerance=3e-3)
        solve["reference_stop_rule"]["stop_rule"] = "parameter_tolerance 3e-3 relative to |x| (pose_graph.cpp:93)"
        solve["solver"] = "harness/lm.py (LM, banded Cholesky on the host; Ceres absent)"
        solve["split"] = ("gpu_evaluation_ms = time inside the registration backend (fused pass, assembly, copy of "
                          "the fused buffer, all-reduce): the product; host_linear_algebra_ms = the harness' own "
                          "assembly + banded Cholesky, which Ceres does in the real system")

    out = None
    if rank == 0:
        passes = args.steps * args.inner
        value = total_evals * passes / dt / 1e6
        # roofline of the dominant kernel (reg_eval_points_kernel<16,float,4>) on rank 0
        bytes_contract = R * BYTES_PER_EVAL
        # `achieved` prices what this launch has to move (DESIGN.md "Roofline accounting"): 36 B written
        # per evaluation, 20 B per registration point the kernel reads (none for the tiles whose chunks
        # all miss the reading submap's block box; once per group of constraints where the launch order
        # lets them share a reference submap's points in one L2, else once per constraint), and 32 B of
        # neighbours per evaluation that interpolates.  Chunk-granular count: conservative, the kernel
        # culls whole 1024-point tiles.
        bytes_conservative = R * BYTES_OUT + pr["read"] * BYTES_POINT + with_corr * BYTES_NEIGHBOURS
        achieved = bytes_conservative / (kernel_ms * 1e-3) / 1e9
        traffic = traffic_fo = traffic_fo_plain = None
        if PROFILE_TRAFFIC:
            t = PROFILE_TRAFFIC
            # same workload (per-launch residual count) as the PMC passes were taken on
            if t.get("residuals_per_launch") == R and t.get("n_gpus") == world:
                traffic = t.get("hbm_bytes_per_launch")
            tf = t.get("full_overlap") or {}
            if fo and tf.get("residuals_per_launch") == fo["R"] and t.get("n_gpus") == world:
                traffic_fo = tf.get("hbm_bytes_per_launch")
            tp = t.get("full_overlap_plain_order") or {}
            if fo and tp.get("residuals_per_launch") == fo["R"] and t.get("n_gpus") == world:
                traffic_fo_plain = tp.get("hbm_bytes_per_launch")
        out = {
            "metric": "Mresiduals+Jacobians/s per GPU; full pose-graph solve ms (200 submaps)",
            "value": value, "unit": "Mresiduals+Jacobians/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "inprocess_gpus": args.gpus if args.inprocess else None,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic" if not dryrun else f"synthetic (DRY RUN: all ranks on one GPU, {dryrun})",
            "config": {"workload": f"configs[2]: synthetic {n_sub} submaps @ "
                                   f"{args.block_dims[0] * 16}x{args.block_dims[1] * 16}x{args.block_dims[2] * 16} voxels "
                                   f"({args.voxel_size} m), {n_con} overlap constraints, kVoxels points, "
                                   "all points (sampling_ratio -1), residual + both Jacobians",
                       "submaps": n_sub, "constraints": n_con,
                       "residuals_per_pass": int(total_evals),
                       "passes_per_step": args.inner,
                       "step": f"{args.inner} consecutive passes over all {n_con} constraints "
                               "(one batched launch per pass per rank)",
                       "parallelism": f"pair-sharded x{world} ({args.placement}" + (", weights = bytes moved at the initial poses" if world > 1 else "")
                                      + "), submaps replicated",
                       "point_order": "extraction (block, then voxel linear index)",
                       "output_placement": (f"row arrays: the fastest of {placement['candidates']} allocations of each kind, chosen by "
                                            "vgx_reg_batch_choose_outputs before the timed region (DESIGN.md 3)")
                       if placement["candidates"] > 1 else "row arrays as the allocator gave them",
                       "solve_stop_rule": "solve.ms: Ceres-default function_tolerance 1e-6 (NOT the reference's "
                                          "rule); solve.reference_stop_rule: parameter_tolerance 3e-3 "
                                          "(pose_graph.cpp:93), which fires on the first step of this graph"},
            "value_per_gpu": value / world,
            "value_with_correspondence": with_corr_total * passes / dt / 1e6,
            "ms_per_pass": dt / passes * 1e3,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         # `traffic` is REPLAYED from profiles/hbm_traffic.json (rocprofv3 --pmc passes of this
                         # same workload, collected by profiles/collect.sh), not measured in this run
                         "traffic_from_profiles": traffic,
                         "traffic_source": PROFILE_TRAFFIC.get("source") if traffic else None,
                         "hbm_frac": (traffic / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic else None,
                         "hbm_frac_note": "counter bytes / this run's kernel time / 8 TB/s",
                         # SURVEY.md 8(d)'s literal pricing, 88 B x every evaluation: NOT an HBM fraction here
                         # (exceeds 1): most evaluations of this workload find no reading block (20 B + 36 B) and
                         # culled tiles are written without being read (36 B) -- `frac` prices those as such;
                         # roofline_full_overlap.plain_order is the workload the 88 B figure describes
                         "contract_88B_frac": bytes_contract / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "contract_88B_frac_note": "not an HBM fraction: prices bytes this launch does not move",
                         # same-run ceilings of THIS box (vgx_bench_stream_ceiling on the row buffers, before and after
                         # the timed region; the lower of the two float4-copy rates is the one used)
                         "copy_ceiling_GBs": min(ceil_before["copy_GBs"], ceil_after["copy_GBs"]),
                         "fill_ceiling_GBs": min(ceil_before["fill_GBs"], ceil_after["fill_GBs"]),
                         "kernel_shaped_ceiling_GBs": ceil_after["kernel_shaped_GBs"],
                         "frac_of_copy_ceiling": achieved / min(ceil_before["copy_GBs"], ceil_after["copy_GBs"]),
                         "traffic_frac_of_copy_ceiling": (traffic / (kernel_ms * 1e-3) / 1e9
                                                          / min(ceil_before["copy_GBs"], ceil_after["copy_GBs"]))
                         if traffic else None,
                         "frac_of_kernel_shaped_ceiling": achieved / ceil_after["kernel_shaped_GBs"],
                         "ceilings": {"before": ceil_before, "after": ceil_after,
                                      "what": "float4 copy jac_ref -> jac_read and non-temporal fill of jac_read (16 B x "
                                              "residuals each way), 3 launches each; kernel_shaped = reads : writes in the "
                                              "headline launch's algorithmic proportion"},
                         "kernel": "reg_eval_points_kernel<16,float,4>",
                         "output_placement": placement,
                         "placement_ms_sets": placement.get("ms_sets"),
                         # What a caller gets WITHOUT choosing: `frac` above is the kernel on the arrays
                         # vgx_reg_batch_choose_outputs picked among the candidates; the same bytes over the FIRST
                         # candidate set's time (what the allocator hands out), over the MEDIAN set's, and over the
                         # placement-proof blocked entry point's (one write front, nothing to choose), all same run
                         "frac_first_allocation": (bytes_conservative / (placement["ms_sets"][0] * 1e-3) / 1e9 / HBM_PEAK_GBS)
                         if placement.get("ms_sets") else achieved / HBM_PEAK_GBS,
                         "frac_median_allocation": (bytes_conservative / (float(np.median(placement["ms_sets"])) * 1e-3) / 1e9
                                                    / HBM_PEAK_GBS) if placement.get("ms_sets") else achieved / HBM_PEAK_GBS,
                         "frac_blocked_layout": (bytes_conservative / (blocked_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if blocked_ms else None,
                         "blocked_layout_ms": blocked_ms,
                         # f64 rows (Ceres' types; 36 more bytes written per row): ms, fraction of peak by its own bytes,
                         # and whether every value's f32 rounding is the timed f32 pass's value
                         "f64_rows_ms": f64_rows_ms,
                         "f64_rows_frac": ((bytes_conservative + 36.0 * R) / (f64_rows_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if f64_rows_ms else None,
                         "f64_rows_round_to_the_f32_rows": f64_rows_match,
                         "frac_note": "frac: arrays selected among the candidates; frac_first / frac_median_allocation: the "
                                      "first / median candidate set (3-launch trials of the selection); frac_blocked_layout: "
                                      "vgx_reg_batch_evaluate_points_blocked, placement-proof",
                         "kernel_ms": kernel_ms, "kernel_ms_max_over_ranks": kernel_ms_max,
                         "bytes_per_unit": BYTES_PER_EVAL, "bytes_per_unit_without_correspondence": BYTES_NO_CORR,
                         "bytes_per_unit_culled": BYTES_OUT,
                         "units_per_launch": int(R),
                         "bytes_per_launch": int(bytes_conservative),
                         "points_read": pr,
                         "pricing": "36 B out per evaluation + 20 B per registration point read (tiles whose "
                                    "512-point chunks all miss the reading submap's block box are written as "
                                    "zeros without reading their points) + 32 B of neighbours per evaluation "
                                    "that interpolates; see roofline_full_overlap for the workload the 88 B "
                                    "contract figure describes",
                         "all_88B_would_read_GBs": bytes_contract / (kernel_ms * 1e-3) / 1e9,
                         "traffic_GBs": (traffic / (kernel_ms * 1e-3) / 1e9) if traffic else None,
                         "with_correspondence_frac": with_corr / max(R, 1)},
            "fused": fused,
            "shipped_config": shipped,
            "multi_context": multi_ctx,
            "solve": solve,
            "setup_s": setup_s,
            "residual_checksum": checksum,
            "box": {"before": box, "during_timed_region": sampler.summary()},
        }
        if fo_out:
            fk = fo_out["kernel_ms"] * 1e-3
            fp = fo_out["points"]
            fo_bytes = fo_out["R"] * BYTES_OUT + fp["read"] * BYTES_POINT + fo_out["with_corr"] * BYTES_NEIGHBOURS
            plain = None
            if fo_out.get("plain_kernel_ms"):
                pk = fo_out["plain_kernel_ms"] * 1e-3
                plain = {"what": "the same constraints launched in plain constraint-major order "
                                 "(VGX_POINTS_TILE_ORDER=0): no two concurrent constraints share a submap, every "
                                 "evaluation moves its own 88 B -- the regime SURVEY.md 8d's contract figure describes",
                         "kernel_ms": fo_out["plain_kernel_ms"], "kernel_ms_max_over_ranks": fo_out["plain_kernel_ms_max"],
                         "bytes_per_unit": BYTES_PER_EVAL,
                         "achieved": fo_out["R"] * BYTES_PER_EVAL / pk / 1e9,
                         "frac": fo_out["R"] * BYTES_PER_EVAL / pk / 1e9 / HBM_PEAK_GBS,
                         "traffic": traffic_fo_plain, "traffic_from_profiles": traffic_fo_plain,
                         "hbm_frac": (traffic_fo_plain / pk / 1e9 / HBM_PEAK_GBS) if traffic_fo_plain else None,
                         "traffic_GBs": (traffic_fo_plain / pk / 1e9) if traffic_fo_plain else None}
            out["roofline_full_overlap"] = {
                "workload": f"the same {n_sub} submaps, each registered against {args.overlap_copies} duplicates of "
                            f"itself perturbed by N(0, {args.pose_sigma} m) / N(0, {args.yaw_sigma} rad) "
                            f"({fo['n']} constraints)",
                "bound": "hbm", "kernel": "reg_eval_points_kernel<16,float,4>",
                "kernel_ms": fo_out["kernel_ms"], "kernel_ms_max_over_ranks": fo_out["kernel_ms_max"],
                "units_per_launch": int(fo_out["R"]), "bytes_per_unit": BYTES_PER_EVAL,
                "points_read": fp, "bytes_per_launch": int(fo_bytes),
                "pricing": "as roofline: 36 B out per evaluation + 20 B per registration point READ (the launch "
                           "order runs the constraints of a reference submap side by side on one XCD, its points "
                           "are fetched once per group) + 32 B per interpolating evaluation",
                "achieved": fo_bytes / fk / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": fo_bytes / fk / 1e9 / HBM_PEAK_GBS,
                "frac_note": "algorithmic bytes / time / 8 TB/s; part of the neighbour bytes is served by the L2: "
                             "hbm_frac is the HBM fraction proper",
                "traffic_from_profiles": traffic_fo,
                "hbm_frac": (traffic_fo / fk / 1e9 / HBM_PEAK_GBS) if traffic_fo else None,
                # the contract's per-evaluation pricing re-counts a point for every constraint that reads it
                # and can exceed the HBM peak where the L2 serves the repeats; plain_order is where it applies
                "per_evaluation_pricing_GBs": fo_out["R"] * BYTES_PER_EVAL / fk / 1e9,
                "plain_order": plain,
                "with_correspondence_frac": fo_out["with_corr"] / max(fo_out["R"], 1),
                "traffic": traffic_fo,
                "traffic_GBs": (traffic_fo / fk / 1e9) if traffic_fo else None,
                "value": fo_out["R_total"] * fo_out["passes"] / fo_out["dt"] / 1e6,
                "value_with_correspondence": fo_out["with_corr_total"] * fo_out["passes"] / fo_out["dt"] / 1e6,
                "unit_value": "Mresiduals+Jacobians/s",
                "fused": fused_fo}
    # CPU baseline: rank 0, N = 1 only (bounded sample)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(capi, ctx, args, true_poses, poses, pairs,
                                           args.cpu_seconds)
        # metric 2 on the CPU, DERIVED (not run: it would take minutes): the converged solve's
        # evaluation count x this graph's residuals / the measured CPU evaluation rates
        if solve:
            cb = out["cpu_baseline"]
            work = solve["evaluations"] * total_evals / 1e6          # M evaluations in the solve
            est = {"evaluations": solve["evaluations"],
                   "port_all_cores_s": work / cb["value"], "port_4_threads_s": work / cb["value_4_threads"],
                   "note": "derived from the measured rates above; excludes the host linear algebra"}
            rs = cb.get("reference_source") or {}
            if rs.get("value_4_threads"):
                est["reference_source_4_threads_s"] = work / rs["value_4_threads"]
            cb["solve_estimate"] = est
    elif rank == 0:
        out["cpu_baseline"] = None
    # second hot path (does not shard: replicas only) -- rank 0, N = 1
    if rank == 0 and world == 1 and not args.no_tsdf:
        # (thread B of `tsdf.*.latency_under_solve_us`: one fused solver evaluation of the config-3 graph, looped)
        def solver_step():
            batch.evaluate_normal(poses, to_host=True)     # (waits for the registration stream alone, as a solver does)
        out["tsdf"] = tsdf_bench(capi, ctx, torch, solver_step=None if args.no_fused else solver_step,
                                 solver_ms_alone=(fused or {}).get("stream_ms_per_step"))
        out["finish_submap"] = finish_bench(capi, ctx, args, true_poses)
    # config 5 (every rank takes part: its constraints are sharded like config 3's)
    if not args.no_config5:
        for o in [batch] + cfs + ([fo["batch"]] + cfs_fo if fo else []) + submaps:
            o.destroy()                                            # make room: config 5 brings its own 1000 submaps
        del residuals, jac_ref, jac_read
        torch.cuda.empty_cache()
        c5 = config5_bench(capi, ctx, torch, dist, use_dist, rank, world, args)
        if rank == 0:
            out["config5"] = c5
            if c5.get("parity"):
                parity_entries.append(c5["parity"])
    if rank == 0 and world == 1 and not args.no_config2:
        from harness import pipeline
        # 30 submaps, 10 Hz, 10 s per submap = 100 scans per submap; the
        # default line carries a bounded cut of the same session (10 submaps x 30 scans)
        full = args.pipeline
        out["pipeline_config2"] = pipeline.run(capi, ctx, torch, n_submaps=30 if full else 10,
                                               scans_per_submap=100 if full else 30)
        out["pipeline_config2"]["cut"] = "full: 30 submaps x 100 scans" if full else \
            "10 x 30 miniature of the 30 x 100 session (--pipeline runs all of it)"
        out["pipeline_config2"]["tsdf_mode"] = "racing (the default: a different legal interleaving every run)"
        # The racing TSDF mode moves the end state from run to run (xy RMSE 0.059 / 0.066 / 0.114 m for the
        # same inputs).  The same cut with the scans integrated in the REPRODUCIBLE mode gives the same maps and the
        # same number on every run and every box -- reported beside the racing run, which is the one that is timed.
        rep = pipeline.run(capi, ctx, torch, n_submaps=30 if full else 10, scans_per_submap=100 if full else 30,
                           deterministic_tsdf=True)
        out["pipeline_config2"]["reproducible_tsdf_mode"] = {
            k: rep[k] for k in ("tsdf_integrate_ms_per_scan", "xy_rmse_m_odometry_only", "xy_rmse_m_optimised",
                                "solve_ms_total", "solve_evaluations_total", "dropped_updates")}
    if rank == 0:
        if not args.no_parity:
            from harness import parity_gate
            out["parity"] = parity_gate.merge(parity_entries)
        out["rccl_ranks"] = world if (use_dist and not dryrun) else 0
        emit(out, args.detail, args.full_line)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

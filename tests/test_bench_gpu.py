"""bench.py end to end on a miniature graph: the N = 1 line carries every object the contract asks
for, and the N = 2 code path (sharding, all-reduce, collective-per-evaluation solve) is walked on
ONE GPU with the gloo backend (VGX_BENCH_DRYRUN, see profiles/README.md) -- the real RCCL runs are
the driver's."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["--grid", "4", "3", "--block-dims", "4", "4", "4", "--block-min", "-2", "-2", "-1",
         "--steps", "2", "--warmup", "1", "--no-tsdf", "--config5-grid", "3", "5"]

pytestmark = pytest.mark.gpu


LINE_LIMIT = 10_000      # the driver failed to parse a 22.4 KB line; r02's 14.7 KB parsed


def _run(cmd, tmp, env=None, timeout=900):
    """runs bench.py; returns (full result object from --detail, the stdout line parsed, the raw line).
    stdout must be EXACTLY one line, the LAST: the compact JSON summary (everything else goes to stderr / the file)."""
    detail = os.path.join(tmp, "detail.json")
    r = subprocess.run(cmd + ["--detail", detail], capture_output=True, text=True, timeout=timeout, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    # (libraries print on stdout too -- RCCL's version banner, gloo's rank announcements: bench.py points fd 1 at stderr
    # for the run and writes its line to the real stdout, so NOTHING else may arrive here, under any launcher)
    lines = r.stdout.splitlines()
    assert len(lines) == 1 and lines[0].startswith("{"), [l[:200] for l in lines[:-1]] + [r.stdout[-300:]]
    assert len(lines[0]) < LINE_LIMIT, len(lines[0])
    line = json.loads(lines[0])
    full = json.load(open(detail))
    # the line is a cut of the full object: same headline, same contract keys first
    assert list(line)[:12] == ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                               "scaling", "vs_baseline", "dtype", "data"]
    assert abs(line["value"] - full["value"]) <= 1e-5 * full["value"]
    return full, line, lines[0]


@pytest.fixture(scope="module")
def single_run(tmp_path_factory):
    return _run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--cpu-seconds", "1"] + SMALL,
                str(tmp_path_factory.mktemp("bench1")), timeout=600)


@pytest.fixture(scope="module")
def single(single_run):
    return single_run[0]


def test_stdout_line_is_small_and_carries_the_contract(single_run):
    """what the driver parses: ONE line, < 10 KB, headline keys first, roofline / cpu_baseline / parity inside"""
    full, line, raw = single_run
    assert len(raw) < LINE_LIMIT
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "parity"):
        assert k in line, k
    assert line["n_gpus"] == 1 and line["steps"] == 2 and line["warmup"] == 1 and line["dtype"] == "f32"
    assert "workload" in line["config"] and "model" not in line["config"]
    rf = line["roofline"]
    assert set(("bound", "achieved", "peak", "unit", "frac", "traffic", "hbm_frac")) <= set(rf)
    assert rf["bound"] == "hbm" and 0 < rf["frac"] <= 1.0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-4
    cb = line["cpu_baseline"]
    assert set(("value", "unit", "cores", "kind", "sample")) <= set(cb) and cb["value"] > 0 and cb["kind"] == "port"
    assert line["parity"]["exact"] is True and line["parity"]["checked"] > 0
    # numbers only below the contract keys: no prose blocks (notes live in the detail file)
    def strings(o, path=""):
        if isinstance(o, dict):
            for k, v in o.items():
                yield from strings(v, path + "/" + k)
        elif isinstance(o, str):
            yield path, o
    long_strings = [(p_, v) for p_, v in strings(line) if len(v) > 180]
    assert not long_strings, long_strings
    assert line["detail"] and "tsdf" not in line            # SMALL runs --no-tsdf


def test_single_gpu_line_has_the_contract_fields(single):
    d = single
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
              "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["dtype"] == "f32" and d["vs_baseline"] is None
    assert set(("bound", "achieved", "peak", "unit", "frac", "traffic")) <= set(d["roofline"])
    assert set(("value", "unit", "cores", "kind", "sample")) <= set(d["cpu_baseline"])
    assert d["value"] > 0 and d["solve"]["position_rmse_m_after"] < d["solve"]["position_rmse_m_before"]
    assert d["fused"]["cost_vs_materialised"] < 1e-6
    # every reported bandwidth is at or below the HBM peak (accounting)
    fo = d["roofline_full_overlap"]
    assert fo["with_correspondence_frac"] > 0.9 and 0 < fo["frac"] <= 1.0
    assert fo["fused"]["cost_vs_materialised"] < 1e-6
    # materialising pass: points read <= points live <= evaluations; the plain-order measurement of the
    # full-overlap workload carries the 88 B-per-evaluation figure
    for r, n in ((d["roofline"], d["roofline"]["units_per_launch"]), (fo, fo["units_per_launch"])):
        pr = r["points_read"]
        assert 0 < pr["read"] <= pr["live"] <= n and pr["distinct"] <= pr["live"] and pr["grouped"] in (0, 1)
        assert 0 < r["frac"] <= 1.0 and r["bytes_per_launch"] <= 88 * n
    assert fo["plain_order"]["kernel_ms"] > 0 and 0 < fo["plain_order"]["frac"] <= 1.0
    for f in (d["fused"], fo["fused"]):
        assert 0 < f["algorithmic_GBs"] <= d["roofline"]["peak"]
        assert f["with_correspondence"] <= f["loaded_after_culling"] <= f["evaluations"]
        assert 0 < f["distinct_points_loaded"] <= f["loaded_after_culling"]
    assert d["config"]["passes_per_step"] == 25 and d["value_with_correspondence"] <= d["value"]
    # same-run parity gate: a sample of the timed launches' constraints against the reference source
    # (exact) and the fused blocks against the oracle (1e-6)
    par = d["parity"]
    assert par["checked"] > 0 and par["exact"] and par["max_rel"] == 0.0 and par["fused_blocks_within_1e-6"], par
    assert c5_headline(d["config5"])
    assert len(par["per_workload"]) == 3
    # every "*frac*" is an HBM fraction <= 1 or says what else it is
    assert d["roofline"]["contract_88B_frac"] > 0 and "not an HBM fraction" in d["roofline"]["contract_88B_frac_note"]
    assert "algorithmic_over_hbm_peak" in d["fused"] and "frac_of_hbm_peak" not in d["fused"]
    assert d["solve"]["gpu_evaluation_ms"] > 0 and d["solve"]["host_linear_algebra_ms"] > 0
    assert d["solve"]["gpu_evaluation_ms"] + d["solve"]["host_linear_algebra_ms"] <= 1.05 * d["solve"]["ms"]
    # the shipped yaml's configuration (sampled, mirrored isosurface constraints) in one batched pass
    assert d["shipped_config"]["constraints"] == 2 * d["config"]["constraints"]
    assert d["shipped_config"]["ms_per_evaluation"] > 0 and d["shipped_config"]["cost"] > 0
    # ... by default on quad bricks made on demand; on the apron bricks for comparison: same draws, same sums
    assert "quad" in d["shipped_config"]["brick_layout_chosen"]
    assert d["shipped_config"]["apron_bricks"]["cost_equals_default"] and d["shipped_config"]["apron_bricks"]["ms_per_evaluation"] > 0
    # the in-process multi-GPU component, two contexts on this GPU: same buffer as the single batch
    assert d["multi_context"]["max_rel_diff_vs_single_batch"] == 0.0 and d["multi_context"]["ms_per_evaluation"] > 0
    # configs[4] in miniature: loop closures + two-stage optimisation improve on the odometry
    c5 = d["config5"]
    assert c5["submaps"] == 15 and c5["loop_closures"] == 20 and c5["solve_ms"] > 0
    assert c5["position_rmse_m_aligned_after"] < c5["position_rmse_m_aligned_odometry"]
    assert c5["stage1_without_registration"]["evaluations"] >= 1
    # configs[1] stand-in, bounded cut
    c2 = d["pipeline_config2"]
    assert c2["submaps"] == 10 and c2["dropped_updates"] == 0 and c2["solves"] == 9
    # ... and once more with the scans integrated in the reproducible mode: the same number on every run
    assert 0 < c2["reproducible_tsdf_mode"]["xy_rmse_m_optimised"] < c2["xy_rmse_m_odometry_only"]


def c5_headline(c5):
    """config 5's headline is the time to within 1e-3 of the final cost, never above solve_ms"""
    return 0 < c5["solve_ms_to_within_1e-3_of_final_cost"] <= c5["solve_ms"] * 1.001


def test_two_rank_path_dry_run_on_one_gpu(single, tmp_path):
    env = dict(os.environ, VGX_BENCH_DRYRUN="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", "29533", os.path.join(ROOT, "bench.py"), "--gpus", "2"] + SMALL
    d, line, _ = _run(cmd, str(tmp_path), env=env)
    assert line["n_gpus"] == 2 and line["cpu_baseline"] is None and line["rccl_ranks"] == 0     # dry run: gloo
    assert d["n_gpus"] == 2 and "DRY RUN" in d["data"] and d["cpu_baseline"] is None
    assert d["fused"]["allreduce_bytes"] == 45 * 8 * d["config"]["constraints"]       # the per-constraint blocks
    # Round 4: ranks exchange per-constraint blocks (one all-reduce, every row written by exactly one rank) and
    # assemble in list order, so the sharded evaluation IS the single-rank one, bit for bit -- the fused buffers
    # have the same SHA-256 and the solve takes the same path to the same bits
    assert d["config"]["residuals_per_pass"] == single["config"]["residuals_per_pass"]
    assert d["fused"]["fused_sha256"] == single["fused"]["fused_sha256"]
    assert d["roofline_full_overlap"]["fused"]["fused_sha256"] == single["roofline_full_overlap"]["fused"]["fused_sha256"]
    for k in ("iterations", "evaluations", "termination", "final_cost", "position_rmse_m_after"):
        assert d["solve"][k] == single["solve"][k], k
    assert d["fused"]["cost"] == single["fused"]["cost"]
    # the driver's N > 1 runs also drive the in-process multi-GPU component (rank 0, all N devices; here both
    # contexts on the one GPU): same fused buffer as the all-reduced one
    mc = d["multi_context"]
    assert mc.get("child") is True, mc         # (under torchrun the component runs in a child process: a fault there -- N real
    assert "error" not in mc, mc               #  devices have never run it -- must not cost the ranks' line)
    assert mc["contexts"] == 2 and mc["ms_per_evaluation"] > 0
    assert mc["cost"] == d["fused"]["cost"]
    # config 5 sharded over two ranks is the single-rank solve, bit for bit
    for k in ("stage1_without_registration", "stage2_all_constraints"):
        assert d["config5"][k] == single["config5"][k], k
    assert d["config5"]["position_rmse_m_aligned_after"] == single["config5"]["position_rmse_m_aligned_after"]


def test_one_rank_under_torchrun_goes_through_rccl(single, tmp_path):
    """The driver's launch form for N > 1 -- `python -m torch.distributed.run ... bench.py --gpus N` -- with N = 1: the
    RCCL group is created (backend "nccl"), the barriers and the int64 all-reduce of the per-constraint blocks run
    through it (a communicator of one rank: what a one-GPU box can show), and the result is the plain run's bit for
    bit."""
    env = {k: v for k, v in os.environ.items() if k != "VGX_BENCH_DRYRUN"}
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
           "--master-addr", "127.0.0.1", "--master-port", "29541", os.path.join(ROOT, "bench.py"), "--gpus", "1",
           "--cpu-seconds", "1"] + SMALL
    d, line, _ = _run(cmd, str(tmp_path), env=env)
    assert line["n_gpus"] == 1 and line["rccl_ranks"] == 1
    assert "DRY RUN" not in d["data"]
    assert d["fused"]["fused_sha256"] == single["fused"]["fused_sha256"]
    assert d["fused"]["cost"] == single["fused"]["cost"]
    assert d["fused"]["allreduce_bytes"] == 45 * 8 * d["config"]["constraints"]
    for k in ("iterations", "evaluations", "termination", "final_cost", "position_rmse_m_after"):
        assert d["solve"][k] == single["solve"][k], k


def test_inprocess_flag_dry_run_on_one_gpu(single, tmp_path):
    """`bench.py --gpus 2 --inprocess`: ONE process, two contexts (here both on the one GPU), the headline loop
    on context 0 and vgx_reg_multi_evaluate_fused over both in `multi_context` -- same line, same keys"""
    env = dict(os.environ, VGX_BENCH_DRYRUN="gloo")
    d, line, _ = _run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--inprocess", "--no-cpu-baseline",
                       "--no-config5", "--no-config2", "--no-solve", "--no-shipped"] + SMALL, str(tmp_path), env=env, timeout=600)
    assert d["n_gpus"] == 1 and d["inprocess_gpus"] == 2
    mc = d["multi_context"]
    assert mc["contexts"] == 2 and mc["device_ids"] == [0, 0] and sum(mc["constraints_per_context"]) == d["config"]["constraints"]
    assert mc["max_rel_diff_vs_single_batch"] == 0.0
    assert mc["cost"] == single["fused"]["cost"]


def test_tsdf_block_has_a_latency_roofline_and_a_sane_all_cores_row():
    """(a) the all-cores CPU row is total points / wall clock behind a barrier and can never
    exceed cores x the one-core rate; (b) the racing kernel's roofline is the latency model (longest chain of
    dependent exchanges x measured round trip), with the HBM figure beside it"""
    import torch
    from harness.bench_tsdf import tsdf_bench
    from voxgraph_amd import capi
    ctx = capi.Context(0)
    try:
        out = tsdf_bench(capi, ctx, torch, scans=5, cpu_scans=2)
    finally:
        ctx.close()
    assert set(out) == {"rgbd_640x480_0.05m", "lidar_64x1024_0.20m_voxgraph_yaml"}
    for name, t in out.items():
        rf = t["roofline"]
        assert rf["bound"] in ("latency", "atomic-throughput") and rf["unit"] == "ms" and rf["longest_walk_steps"] >= 1, (name, rf)
        assert 20.0 < rf["roundtrip_ns_unloaded"] < 20000.0 and rf["atomic_peak_Gops"] > 0.1, rf
        assert rf["peak"] == max(rf["latency_chain_ms"], rf["atomic_throughput_ms"])
        assert abs(rf["frac"] - rf["peak"] / rf["achieved"]) < 1e-9 and 0 < rf["frac"] <= 1.0, rf   # a LOWER bound on time
        assert 0 < rf["hbm_frac"] < 1.0
        ac = t["cpu_baseline"]["all_cores"]
        assert ac["at_most_cores_x_one_core"], ac
        assert ac["scans_per_replica"] >= 50 and ac["wall_s"] >= ac["slowest_replica_s"] > 0
        assert t["reproducible_mode"]["parity_vs_oracle"]["bit_identical"]

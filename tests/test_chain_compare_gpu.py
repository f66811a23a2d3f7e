"""Data or implementation?  The first six submaps of the config-2 stand-in session go through the
GPU chain and through an oracle-only chain (harness/chain_compare.py) in the reference's default
mode -- mirrored isosurface constraints against the reading submap's ESDF
(registration_cost_function.h:35, pose_graph.cpp:62-71) -- sharing only the scans and the solver.

Measured (profiles/r02_chain_compare_{6,30}submaps.json): the two chains end within a few cm of each
other and drift from the ground truth ALIKE (30-submap lap, from truth: GPU 0.11 m / oracle 0.13 m;
from drift 0.29 / 0.28 m), so the residual registration error of that synthetic session is a
property of the data, not of the kernels.  The one deterministic producer, the ESDF, agrees with
voxblox's queue (restated) to 1e-7 at the 99th percentile and 1-2 mm at worst (the queue's own
min_diff_m = 1 mm slack), never above it."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def result():
    import torch
    from voxgraph_amd import capi
    from harness import chain_compare
    ctx = capi.Context(0)
    out = chain_compare.run(capi, ctx, torch, n_submaps=6, scans_per_submap=30, use_esdf_distance=True,
                            isosurface_points=True)
    ctx.close()
    return out


def test_esdf_fields_agree_on_the_oracles_tsdf(result):
    for m, e in enumerate(result["esdf_gpu_vs_oracle_same_tsdf"]):
        assert e["observed_equal"], m
        assert e["n_observed"] > 100000
        # exact fixed point vs a queue that ignores improvements below 1 mm: equal to f32 rounding
        # almost everywhere, within ~2 x min_diff_m in the few places the queue stopped early,
        # and never farther from the surface than the queue's answer
        assert e["p99"] <= 1e-3 and e["mean"] <= 1e-4 and e["max"] <= 2.5e-3, (m, e)
        assert e["gpu_never_above_oracle"], m


def test_both_chains_end_in_the_same_place(result):
    summary = {k: v for k, v in result.items() if k.startswith("from_") or k.startswith("xy_")}
    print(summary)
    assert result["constraints"] >= 10
    for start in ("from_truth", "from_drift"):
        g, o = result[f"{start}_gpu"]["xy_rmse_m"], result[f"{start}_oracle"]["xy_rmse_m"]
        # The street of this cut constrains the along-street direction weakly, and the GPU's TSDF
        # differs from the oracle's by a legal reordering of racing updates (p99 of |d distance|
        # 7-8 cm) and from run to run: the end states land a few cm apart (measured 0.2-4 cm; on
        # the 30-submap lap the two RMSEs agree to 3-18 %, profiles/r02_chain_compare_30submaps.json)
        assert abs(g - o) <= 0.10, (start, g, o, summary)    # racing mode: r2 driver box saw 0.057
        d = result[f"{start}_end_pose_difference"]
        assert d["xy_max_m"] < 0.20 and d["yaw_max_rad"] < 0.01, (start, d, summary)
    # both improve on the odometry from the drifted start, and neither wanders off from the truth
    assert result["from_drift_gpu"]["xy_rmse_m"] < 0.6 * result["xy_rmse_m_odometry_only"], summary
    assert result["from_truth_gpu"]["xy_rmse_m"] < 0.16 and result["from_truth_oracle"]["xy_rmse_m"] < 0.12, summary


def test_reference_source_agrees_with_the_oracle_chain(result):
    chk = result["reference_source_cost_at_oracle_end_state"]
    if "reference" not in chk:
        pytest.skip("oracle/_ref not present: " + str(chk))
    assert chk["equal"], chk


def test_tsdf_producers_allocate_the_same_blocks(result):
    for t in result["tsdf_gpu_vs_oracle"]:
        assert t["blocks_gpu"] == t["blocks_oracle"] == t["blocks_common"]
        assert t["p50"] <= 1e-6 and t["p99"] < 0.15      # race order only: the bulk is identical

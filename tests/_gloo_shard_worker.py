"""world_size-2 gloo worker for tests/test_harness_cpu.py (CPU only)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import lpt_shards  # noqa: E402
from harness.backends import assemble_fused  # noqa: E402
from oracle import pyoracle as orc  # noqa: E402
from oracle import synth  # noqa: E402
from tests import helpers as H  # noqa: E402


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    sdf = synth.sphere_ground_sdf((1.6, 1.6, 1.2), 1.0, 0.35)
    true = [(0, 0, 0, 0), (0.8, 0.1, 0.0, 0.1), (0.1, 0.9, 0.05, -0.15), (0.9, 0.8, 0.0, 0.2)]
    layers, pts = [], []
    for p in true:
        sm = synth.make_submap(sdf, 0.1, 16, (0, 0, 0), (2, 2, 2), 0.3, p, 1.0, drop_empty_blocks=True)
        layers.append(H.oracle_layer(sm))
        pts.append(H.oracle_points(sm))
    pairs = [(0, 1), (1, 0), (0, 2), (1, 3), (2, 3), (0, 3)]
    poses = np.array(true, np.float64) + np.random.default_rng(3).normal(0, 0.02, (4, 4))

    def normal(c):
        a, b = pairs[c]
        ok, cost, jtr, jtj = orc.reg_evaluate_normal(layers[b], *pts[a], poses[a], poses[b])
        return np.concatenate([[cost], jtr, jtj])

    shards = lpt_shards([len(pts[a][2]) for a, _ in pairs], world)
    assert sorted(sum(shards, [])) == list(range(len(pairs)))
    mine = shards[rank]
    full = assemble_fused([normal(c) for c in range(len(pairs))], pairs, 4)
    # round 1-3 scheme: every rank assembles its shard, the fused buffers are summed -- equal to 1e-12 only
    # (the association of a node's sum depends on the sharding)
    buf = assemble_fused([normal(c) for c in mine], [pairs[c] for c in mine], 4,
                         n_global=len(pairs), global_index=mine)
    t = torch.from_numpy(buf)
    dist.all_reduce(t)
    np.testing.assert_allclose(t.numpy(), full, rtol=1e-12, atol=1e-12)
    # round 4 scheme (include/voxgraph_amd.h "Sharding-independent assembly", harness/backends.py GpuBackend):
    # all-reduce the per-constraint BLOCKS -- every row written by exactly one rank, zero elsewhere, so the
    # sum is exact in any order -- then assemble in list order on every rank: the unsharded buffer BIT FOR BIT
    blocks = np.zeros((len(pairs), 45))
    for c in mine:
        blocks[c] = normal(c)
    blocks[mine[0], 7] = -0.0                       # a negative zero must come through as one
    tb = torch.from_numpy(blocks)
    dist.all_reduce(tb.view(torch.int64))           # integer sum of disjoint rows: the bit patterns themselves
    want = np.stack([normal(c) for c in range(len(pairs))])
    want[shards[0][0], 7] = -0.0
    want[shards[1][0], 7] = -0.0
    assert np.array_equal(tb.numpy().view(np.uint64), want.view(np.uint64))
    assert np.array_equal(assemble_fused(list(tb.numpy()), pairs, 4), assemble_fused(list(want), pairs, 4))
    dist.barrier()
    if rank == 0:
        print("SHARD_ALLREDUCE_OK")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()

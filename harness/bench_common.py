"""Constants and helpers shared by bench.py and its parts (harness/bench_*.py)."""
import numpy as np

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
BYTES_PER_EVAL = 88            # SURVEY.md 8d contract figure (materialising, f32 outputs)
BYTES_NO_CORR = 56             # an evaluation that finds no reading block: 20 B in, 36 B out
BYTES_OUT, BYTES_POINT, BYTES_NEIGHBOURS = 36, 20, 32   # the three parts of the 88 B
BYTES_PER_EVAL_FUSED = 52      # fused form: 20 B point + 32 B neighbours, nothing written per point
BYTES_NO_CORR_FUSED = 20       # a point the fused pass loads but that finds no reading block


def lpt_shards(weights, n):
    """Greedy longest-processing-time partition of constraints onto n ranks: the library's own
    placement (vgx_lpt_shards), the one the in-process multi-GPU component uses."""
    from voxgraph_amd import capi
    shard_of = capi.lpt_shards(weights, n)
    return [[int(c) for c in np.nonzero(shard_of == r)[0]] for r in range(n)]


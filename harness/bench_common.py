"""Constants and helpers shared by bench.py and its parts (harness/bench_*.py)."""
import numpy as np

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
BYTES_PER_EVAL = 88            # SURVEY.md 8d contract figure (materialising, f32 outputs)
BYTES_NO_CORR = 56             # an evaluation that finds no reading block: 20 B in, 36 B out
BYTES_OUT, BYTES_POINT, BYTES_NEIGHBOURS = 36, 20, 32   # the three parts of the 88 B
BYTES_PER_EVAL_FUSED = 52      # fused form: 20 B point + 32 B neighbours, nothing written per point
BYTES_NO_CORR_FUSED = 20       # a point the fused pass loads but that finds no reading block


PLACEMENT = "contiguous"      # bench.py --placement; set once in main()


def lpt_shards(weights, n):
    """Greedy longest-processing-time partition of constraints onto n ranks: the library's own
    placement (vgx_lpt_shards)."""
    from voxgraph_amd import capi
    shard_of = capi.lpt_shards(weights, n)
    return [[int(c) for c in np.nonzero(shard_of == r)[0]] for r in range(n)]


def place(weights, n, placement=None):
    """shard index per constraint by the bench's placement rule: vgx_contiguous_shards (default: consecutive runs
    of equal weight -- a shard touches one stretch of the map; profiles/r04_shard_balance.json: better balance of
    the measured shard times AND a third of the submaps per shard) or vgx_lpt_shards.  Results do not depend on it."""
    from voxgraph_amd import capi
    rule = placement or PLACEMENT
    return (capi.contiguous_shards if rule == "contiguous" else capi.lpt_shards)(np.asarray(weights, np.int64), n)


def shards(weights, n, placement=None):
    """the same as lists of constraint indices per rank"""
    shard_of = place(weights, n, placement)
    return [[int(c) for c in np.nonzero(shard_of == r)[0]] for r in range(n)]


def effective_cores():
    """How many host threads can really run at once here: the smallest of os.cpu_count(), the affinity mask and
    the cgroup CPU quota (a GPU box's container may show 256 CPUs and be allowed a dozen)."""
    import math
    import os
    n = os.cpu_count() or 1
    info = {"cpu_count": n}
    try:
        info["affinity"] = len(os.sched_getaffinity(0))
        n = min(n, info["affinity"])
    except (AttributeError, OSError):
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    info["cgroup_quota_cores"] = float(txt[0]) / float(txt[1])
            else:
                q = float(txt[0])
                if q > 0:
                    info["cgroup_quota_cores"] = q / float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            break
        except (OSError, ValueError, IndexError):
            continue
    if "cgroup_quota_cores" in info:
        n = max(1, min(n, int(math.ceil(info["cgroup_quota_cores"]))))
    info["effective"] = n
    return n, info

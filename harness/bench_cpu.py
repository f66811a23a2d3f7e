"""bench.py's `cpu_baseline` leg for the REG path: the CPU port (oracle/reg_oracle.c) and, where it
travelled, the reference's own source (oracle/_ref) timed on this host's cores.  Checker side: the only
bench part that imports oracle/."""
import os
import sys
import time

import numpy as np

from harness.bench_common import effective_cores

def cpu_baseline(capi, ctx, args, true_poses, poses, pairs, seconds):
    """The CPU oracle ("port") timed on this host on ONE constraint of the same
    workload, replicated over all host cores (one constraint per task, the
    reference's parallelism axis, pose_graph.cpp:96)."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import pyoracle as orc
    a, b = int(pairs[0][0]), int(pairs[0][1])
    subs = {}
    for k in (a, b):
        sm = capi.Submap.synth_city(ctx, k, args.voxel_size, 16, args.block_min, args.block_dims,
                                    args.truncation, args.esdf_max, 10.0, true_poses[k], args.seed)
        td, tw, ed, eo = sm.download_layers(16)
        subs[k] = (sm.block_index(), td, tw, ed, eo)
        sm.destroy()
    bi, td, tw, ed, eo = subs[a]
    xyz, dist, w = orc.find_relevant_voxels(args.voxel_size, 16, bi, td, tw, ed)
    bi, td, tw, ed, eo = subs[b]
    layer = orc.Layer(args.voxel_size, 16, bi, ed, eo)
    cores, cores_info = effective_cores()      # threads that can really run at once (affinity, cgroup quota)
    n = len(w)

    def task(_):
        ok, r, jo, je = orc.reg_evaluate(layer, xyz, dist, w, poses[a], poses[b])
        return n

    task(0)                                   # page everything in
    done, t0 = 0, time.perf_counter()
    with ThreadPoolExecutor(cores) as ex:
        while time.perf_counter() - t0 < seconds:
            done += sum(ex.map(task, range(cores)))
    dt = time.perf_counter() - t0
    # the reference's own setting: Ceres num_threads = 4 (pose_graph.cpp:96)
    done4, t4 = 0, time.perf_counter()
    with ThreadPoolExecutor(4) as ex:
        while time.perf_counter() - t4 < max(2.0, seconds / 4):
            done4 += sum(ex.map(task, range(4)))
    dt4 = time.perf_counter() - t4
    out = {"value": done / dt / 1e6, "unit": "Mresiduals+Jacobians/s", "cores": cores,
           "value_4_threads": done4 / dt4 / 1e6, "host_cpus": cores_info,
           "kind": "port",
           "sample": f"constraint 0 of the same graph ({n} residuals, one 256^3 pair) evaluated "
                     f"{done // n} times, one evaluation per task on {cores} threads, {dt:.1f} s; "
                     "oracle/reg_oracle.c (" + orc.build_flags() + ")"}
    # The reference's OWN RegistrationCostFunction::Evaluate (oracle/_ref: its source compiled
    # against stand-in headers, hashed-block voxblox layer included), when the prebuilt library
    # travelled here: same constraint, same poses, its results checked against the port's.
    try:
        from oracle import ref_reg
        if ref_reg.available():
            subs_ref = {}
            for k in (a, b):
                bi, td, tw, ed, eo = subs[k]
                subs_ref[k] = ref_reg.Submap(k, true_poses[k], args.voxel_size, 16, bi, td, tw, ed, eo)
            # the reference walks its hash map in its own block order: same point SET, so compare sorted
            cf0 = ref_reg.RegistrationCostFunction(subs_ref[a], subs_ref[b])
            ok_r, r_ref, _, _ = cf0.Evaluate(poses[a], poses[b])
            ok_p, r_port, _, _ = orc.reg_evaluate(layer, xyz, dist, w, poses[a], poses[b])
            same = bool(ok_r and ok_p and np.array_equal(np.sort(r_ref), np.sort(r_port)))
            n_thr = min(cores, 64)
            cfs = [ref_reg.RegistrationCostFunction(subs_ref[a], subs_ref[b]) for _ in range(n_thr)]

            def ref_task(i):
                cfs[i].Evaluate(poses[a], poses[b])
                return cfs[i].num_residuals()

            def timed(threads, budget):
                cnt, t = 0, time.perf_counter()
                with ThreadPoolExecutor(threads) as ex:
                    while time.perf_counter() - t < budget:
                        cnt += sum(ex.map(ref_task, range(threads)))
                return cnt / (time.perf_counter() - t) / 1e6
            out["reference_source"] = {
                "kind": "reference", "unit": "Mresiduals+Jacobians/s",
                "value": timed(n_thr, max(2.0, seconds / 3)), "cores": n_thr,
                "value_4_threads": timed(4, max(2.0, seconds / 6)),
                "residuals_equal_to_port": same,
                "sample": "the same constraint through /root/reference's registration_cost_function.cpp, "
                          "compiled (g++ -O2) against oracle/ref_shims (hashed 16^3 blocks of 12/20-byte "
                          "voxels behind shared_ptr, minimal Eigen); one cost function per thread"}
    except Exception as e:                                    # the checker must never sink the bench
        out["reference_source"] = {"error": repr(e)}
    return out


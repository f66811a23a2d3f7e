#!/usr/bin/env python3
"""Lines the counter rows of profiles/placement_pmc.sh's passes up with the labels profiles/probes/points_placement_pmc.py
wrote: per pass and counter, the value per launch on the FAST and on the SLOW set of that process and their ratio; for the
raw per-channel passes (json output keeps the 16 x 8 TCC instances) the spread over channels.
usage: placement_pmc_summary.py gpurun_out/placement_pmc"""
import collections
import csv
import glob
import json
import os
import sys

out_dir = sys.argv[1]


def labelled_rows(name):
    lab = json.load(open(os.path.join(out_dir, f"{name}_labels.json")))
    files = sorted(glob.glob(os.path.join(out_dir, name, "**", "*counter_collection.csv"), recursive=True))
    per = collections.OrderedDict()                       # dispatch id -> {counter: value}
    times = {}
    for f in files:
        for x in csv.DictReader(open(f)):
            if "reg_eval_points_kernel" not in x["Kernel_Name"]:
                continue
            d = int(x["Dispatch_Id"])
            per.setdefault(d, collections.defaultdict(float))[x["Counter_Name"]] += float(x["Counter_Value"])
            if x.get("Start_Timestamp") and x.get("End_Timestamp"):
                times[d] = (int(x["End_Timestamp"]) - int(x["Start_Timestamp"])) * 1e-6
    ids = sorted(per)
    n_lab = len(lab["labelled"])
    tail = ids[-(n_lab + 10):-10] if len(ids) >= n_lab + 10 else []     # 2 x (1 + 4) timed launches follow the labelled ones
    return lab, [(l, per[d], times.get(d)) for l, d in zip(lab["labelled"], tail)], len(ids)


def main():
    names = sorted({os.path.basename(p)[:-len("_labels.json")] for p in glob.glob(os.path.join(out_dir, "*_labels.json"))})
    for name in names:
        lab = json.load(open(os.path.join(out_dir, f"{name}_labels.json")))
        print(f"== pass {name}: sets {['%.3f' % x for x in lab['ms_round0']]} / {['%.3f' % x for x in lab['ms_round1']]} ms; "
              f"fast set {lab['fast_set']} {lab['fast_ms']:.3f} ms, slow set {lab['slow_set']} {lab['slow_ms']:.3f} ms "
              f"(after the labelled launches: {lab['ms_after']['fast']:.3f} / {lab['ms_after']['slow']:.3f})")
        if name == "unprofiled":
            continue
        try:
            lab, rows, n = labelled_rows(name)
        except Exception as e:   # noqa: BLE001
            print("   no counter rows:", repr(e))
            continue
        if not rows:
            print(f"   {n} dispatches of the kernel in the trace, fewer than the labelled tail needs")
            continue
        counters = sorted({c for _, r, _ in rows for c in r})
        for c in counters:
            v = {k: [r[c] for l, r, _ in rows if l == k] for k in ("fast", "slow")}
            mf, msl = sum(v["fast"]) / len(v["fast"]), sum(v["slow"]) / len(v["slow"])
            print(f"   {c:44s} fast {mf:16.1f}  slow {msl:16.1f}  slow/fast {msl / mf if mf else float('nan'):7.3f}   "
                  f"(fast launches {[int(x) for x in v['fast']]}, slow {[int(x) for x in v['slow']]})")
        t = {k: [x for l, _, x in rows if l == k and x] for k in ("fast", "slow")}
        if t["fast"] and t["slow"]:
            print(f"   kernel ms under the profiler: fast {sum(t['fast']) / len(t['fast']):.3f}  slow {sum(t['slow']) / len(t['slow']):.3f}")
        # per-instance values, where the json output has them
        for jf in sorted(glob.glob(os.path.join(out_dir, name, "**", "*results.json"), recursive=True)):
            try:
                per_channel(jf, lab)
            except Exception as e:   # noqa: BLE001
                print("   per-channel parse of", os.path.basename(jf), "failed:", repr(e)[:200])


def per_channel(jf, lab):
    j = json.load(open(jf))
    sdk = j["rocprofiler-sdk-tool"][0]
    names = {}
    for c in sdk.get("counters", []):
        names[c["id"]["handle"] if isinstance(c.get("id"), dict) else c.get("id")] = c.get("name")
    recs = sdk["callback_records"]["counter_collection"] if "callback_records" in sdk and "counter_collection" in sdk["callback_records"] \
        else sdk.get("buffer_records", {}).get("counter_collection", [])
    kern = {k["kernel_id"]: k.get("formatted_kernel_name", k.get("kernel_name", "")) for k in sdk.get("kernel_symbols", [])}
    disp = []
    for r in recs:
        info = r.get("dispatch_data", {}).get("dispatch_info", {})
        if "reg_eval_points_kernel" not in kern.get(info.get("kernel_id"), ""):
            continue
        vals = collections.defaultdict(list)
        for rec in r.get("records", []):
            cid = rec.get("counter_id", {}).get("handle") if isinstance(rec.get("counter_id"), dict) else rec.get("counter_id")
            vals[names.get(cid, str(cid))].append(float(rec.get("value", 0.0)))
        disp.append((info.get("dispatch_id"), vals))
    disp.sort(key=lambda x: x[0])
    n_lab = len(lab["labelled"])
    tail = disp[-(n_lab + 10):-10]
    for cname in sorted({c for _, v in tail for c in v}):
        for which in ("fast", "slow"):
            rows = [v[cname] for (l, (_, v)) in zip(lab["labelled"], tail) if l == which and cname in v]
            if not rows:
                continue
            n_inst = len(rows[0])
            mean = [sum(r[i] for r in rows) / len(rows) for i in range(n_inst)]
            tot = sum(mean)
            if n_inst > 1 and tot > 0:
                srt = sorted(mean)
                print(f"   {cname:30s} {which}: {n_inst} instances, total {tot:16.0f}, min {srt[0]:12.0f} median {srt[n_inst // 2]:12.0f} "
                      f"max {srt[-1]:12.0f}  max/mean {srt[-1] / (tot / n_inst):6.3f}")


main()

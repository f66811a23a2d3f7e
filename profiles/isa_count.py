#!/usr/bin/env python3
"""Static instruction mix of one kernel from hipcc's -save-temps assembly (no GPU needed).

    hipcc --offload-arch=gfx950 ... -save-temps=obj -c vgx_reg.hip -o /tmp/vgx_reg.o
    python profiles/isa_count.py /tmp/vgx_reg-hip-amdgcn-amd-amdhsa-gfx950.s reg_eval_reduce_kernelILi16

Prints, for the first kernel whose mangled name contains the pattern: VGPR/SGPR counts and the
instruction count per class for the whole body and for its hottest loop (the longest
back-branch region), which is what the per-point VALU figures in profiles/README.md come from.
"""
import collections
import re
import sys


def classify(op):
    if op.startswith("v_pk_"):
        return "valu_pk"
    if op.startswith("v_") and ("_f64" in op or "f64_" in op):
        return "valu_f64"
    if op.startswith("v_cvt"):
        return "valu_cvt"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("s_waitcnt"):
        return "waitcnt"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("global_load") or op.startswith("flat_load") or op.startswith("buffer_load"):
        return "vmem_load"
    if op.startswith("global_store") or op.startswith("flat_store") or op.startswith("buffer_store"):
        return "vmem_store"
    if op.startswith("global_atomic") or op.startswith("flat_atomic"):
        return "vmem_atomic"
    if op.startswith("ds_"):
        return "lds"
    return "other"


def main():
    path, pat = sys.argv[1], sys.argv[2]
    lines = open(path).read().split("\n")
    start = None
    for i, l in enumerate(lines):
        if re.match(r"^_Z\S*:", l) and pat in l:
            start = i
            name = l.rstrip(":")
            break
    if start is None:
        raise SystemExit("kernel not found")
    body = []
    for l in lines[start + 1:]:
        if l.startswith("\t.section") or l.startswith(".Lfunc_end"):
            break
        body.append(l)
    labels, insts = {}, []
    for l in body:
        m = re.match(r"^(\.LBB\S+):", l)
        if m:
            labels[m.group(1)] = len(insts)
            continue
        s = l.strip()
        if not s or s.startswith(";") or s.startswith("."):
            continue
        insts.append(s.split()[0] + " " + " ".join(s.split()[1:]))
    total = collections.Counter(classify(i.split()[0]) for i in insts)
    # hottest loop = longest region closed by a backward branch
    best = (0, 0)
    for k, ins in enumerate(insts):
        op = ins.split()[0]
        if op.startswith("s_cbranch") or op == "s_branch":
            tgt = ins.split()[1].rstrip(",")
            if tgt in labels and labels[tgt] <= k and k - labels[tgt] > best[1] - best[0]:
                best = (labels[tgt], k + 1)
    loop = collections.Counter(classify(i.split()[0]) for i in insts[best[0]:best[1]])
    meta = {}
    txt = "\n".join(lines)
    m = re.search(r"\.name:\s+" + re.escape(name) + r"\n(.*?)\.wavefront_size", txt, re.S)
    if m:
        for key in ("sgpr_count", "vgpr_count", "vgpr_spill_count", "group_segment_fixed_size"):
            mm = re.search(key + r":\s+(\d+)", m.group(1))
            if mm:
                meta[key] = int(mm.group(1))
    m2 = re.search(r"\.name:\s+" + re.escape(name) + r"\n", txt)
    print(name)
    print(" meta:", meta)
    print(" whole body (%d instructions):" % len(insts), dict(total))
    print(" longest loop (%d instructions):" % (best[1] - best[0]), dict(loop))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""one line per measurement out of scan_latency_probe.py's json lines (stdin or a file)"""
import json
import re
import sys

for line in (open(sys.argv[1]) if len(sys.argv) > 1 else sys.stdin):
    line = line.strip()
    if line.startswith("==="):
        print(line)
        continue
    m = re.match(r"(\w+) under (.*?) (\{.*\})$", line)
    if m:
        d = json.loads(m.group(3))
        print("%-5s %-48s alone p50 %6.0f p99 %6.0f | under p50 %6.0f p99 %6.0f max %6.0f us | solver %.3f ms/eval" % (
            m.group(1), m.group(2), d["alone"]["p50"], d["alone"]["p99"], d["p50"], d["p99"], d["max"],
            d["solver_ms_per_evaluation_meanwhile"]))

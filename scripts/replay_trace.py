#!/usr/bin/env python3
"""Post-process a rocprofv3 --kernel-trace CSV of a script that replays ONE hipGraph many times: the last replay's kernels in
order with duration and the idle gap before each (usage: replay_trace.py kernel_trace.csv last_kernel_substring)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if sys.argv[2] in r["Kernel_Name"]]
a, b = idx[-2] + 1, idx[-1] + 1
prev = int(rows[a - 1]["End_Timestamp"])
tk = tg = 0.0
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%7.2f %7.2f  %s" % ((e - s) / 1e3, (s - prev) / 1e3, r["Kernel_Name"].replace("(anonymous namespace)::", "")[:100]))
    tk += (e - s) / 1e3; tg += (s - prev) / 1e3
    prev = max(prev, e)
print("kernels %.1f us, gaps %.1f us, %d kernels" % (tk, tg, b - a))

#!/usr/bin/env python3
"""Per-kernel launch durations of a rocprofv3 kernel trace (min / median / p90 / max) and, for one window, the launches in start order:
a launch whose median is many times its minimum waited for room beside another kernel (the trace's `start` is when the dispatch
began, not when its first workgroup ran).    python tools/trace_durations.py <kernel_trace.csv> [first hinted launch of the window] [rows]"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
d = collections.defaultdict(list)
for r in rows:
    d[r["Kernel_Name"].split("(")[0].replace("gtx::", "")].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for n, v in d.items():
    if ("gtx" in n or "rocclr" in n) and len(v) > 5:
        v2 = sorted(v)
        print("%-40s n=%4d min %7.1f median %7.1f p90 %7.1f max %7.1f us" % (n[-40:], len(v), v2[0], v2[len(v2) // 2], v2[int(len(v2) * 0.9)], v2[-1]))
if len(sys.argv) > 2:
    hs = [r for r in rows if "gtx_align_hinted" in r["Kernel_Name"]]
    t0 = int(hs[int(sys.argv[2])]["Start_Timestamp"])
    for r in [r for r in rows if int(r["Start_Timestamp"]) >= t0][:int(sys.argv[3]) if len(sys.argv) > 3 else 60]:
        print("%-40s q%-3s %9.1f %9.1f %7.1f  grid %s wg %s" % (r["Kernel_Name"].split("(")[0].replace("gtx::", "")[-40:], r["Queue_Id"], (int(r["Start_Timestamp"]) - t0) / 1e3,
                                                              (int(r["End_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r["Grid_Size_X"], r["Workgroup_Size_X"]))

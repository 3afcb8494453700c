#!/usr/bin/env python3
"""Kernel-by-kernel list of the first part of the last bench step of a rocprofv3 kernel trace (start, gap, duration).
usage: trace_list.py <dir with *_kernel_trace.csv> [max time us]"""
import csv, glob, re, sys
d = sys.argv[1]
tmax = float(sys.argv[2]) if len(sys.argv) > 2 else 3000.0
path = glob.glob(d + "/**/*_kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n); n = re.sub(r"^void ", "", n)
    if "rocprim" in n:
        k = re.search(r"(radix_sort\w*|reduce_by_key\w*|scan\w*|lookback\w*|trivial_runs\w*|select\w*|partition\w*|histogram\w*|transform\w*)", n)
        return "rocprim::" + (k.group(1) if k else "kernel")
    m = re.match(r"([\w:]+)", n); return m.group(1) if m else n[:40]
starts = [i for i, r in enumerate(rows) if "k_emit_rows" in r["Kernel_Name"]]
step_starts = [starts[0]]
for a, b in zip(starts, starts[1:]):
    if int(rows[b]["Start_Timestamp"]) - int(rows[a]["End_Timestamp"]) > 3e6: step_starts.append(b)
sel = rows[step_starts[-1]:]
t0 = int(sel[0]["Start_Timestamp"]); prev = t0
for r in sel:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if (s - t0) / 1e3 > tmax: break
    print("%9.1f  gap %6.1f  dur %7.1f  grid %9s  %s" % ((s - t0) / 1e3, (s - prev) / 1e3, (e - s) / 1e3, r.get("Grid_Size_X", r.get("Grid_Size", "?")), short(r["Kernel_Name"])))
    prev = max(prev, e)

#!/usr/bin/env python3
"""Timeline of the last bench step of a rocprofv3 kernel trace: phases (assembly / cascade / fine solve), idle gaps.
usage: trace_gaps.py <dir with *_kernel_trace.csv> [gap threshold us]"""
import csv, glob, re, sys
d = sys.argv[1]
thr = float(sys.argv[2]) if len(sys.argv) > 2 else 15.0
path = glob.glob(d + "/**/*_kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n); n = re.sub(r"^void ", "", n)
    if "rocprim" in n:
        k = re.search(r"(radix_sort\w*|reduce_by_key\w*|scan\w*|lookback\w*|trivial_runs\w*|select\w*|partition\w*)", n)
        return "rocprim::" + (k.group(1) if k else "kernel")
    m = re.match(r"([\w:]+)", n); return m.group(1) if m else n[:40]
# the last step starts at the last k_emit_rows burst: find last index of k_emit_rows preceded by a long gap
starts = [i for i, r in enumerate(rows) if "k_emit_rows" in r["Kernel_Name"]]
# group emit_rows launches into steps (levels emit close together)
step_starts = [starts[0]]
for a, b in zip(starts, starts[1:]):
    if int(rows[b]["Start_Timestamp"]) - int(rows[a]["End_Timestamp"]) > 3e6: step_starts.append(b)
i0 = step_starts[-1]
sel = rows[i0:]
t0 = int(sel[0]["Start_Timestamp"]); prev = t0; busy = 0
print("last step: %d kernels" % len(sel))
for r in sel:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if (s - prev) / 1e3 > thr:
        print("  %9.1f us: idle %.1f us before %s" % ((s - t0) / 1e3, (s - prev) / 1e3, short(r["Kernel_Name"])))
    busy += e - s; prev = max(prev, e)
print("span %.2f ms, kernels busy %.2f ms" % ((prev - t0) / 1e6, busy / 1e6))

#!/usr/bin/env python3
"""Per-stream (queue) timeline of the ASSEMBLY phase of the last step of a rocprofv3 kernel trace: for every queue the first
and last kernel, busy time, launches; then the kernels of each queue in order.
usage: trace_streams.py <dir with *_kernel_trace.csv> [t_max us]"""
import csv, glob, re, sys
d = sys.argv[1]
tmax = float(sys.argv[2]) if len(sys.argv) > 2 else 4000.0
path = glob.glob(d + "/**/*_kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
print("columns:", list(rows[0].keys()))
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
t0 = int(sel[0]["Start_Timestamp"])
qkey = "Stream_Id" if "Stream_Id" in sel[0] else "Queue_Id"
byq = {}
for r in sel:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if (s - t0) / 1e3 > tmax: break
    byq.setdefault(r[qkey], []).append(((s - t0) / 1e3, (e - s) / 1e3, short(r["Kernel_Name"]), r.get("Grid_Size_X", r.get("Grid_Size", "?"))))
for q, ks in byq.items():
    print("   queues of this stream:", sorted({r["Queue_Id"] for r in sel if r[qkey] == q}))
    print("\n== %s %s: %d kernels, first %.1f us, last ends %.1f us, busy %.1f us" % (qkey, q, len(ks), ks[0][0], ks[-1][0] + ks[-1][1], sum(k[1] for k in ks)))
    prev = ks[0][0]
    for (s, dur, name, grid) in ks:
        print("  %9.1f gap %7.1f dur %7.1f grid %9s %s" % (s, s - prev, dur, grid, name))
        prev = s + dur

#!/usr/bin/env python3
"""The head of the last bench step of a rocprofv3 kernel trace with the hardware queue and the HIP stream of every launch:
which chains of the assembly share a queue (and so run one after the other).
usage: trace_queues.py <dir with *_kernel_trace.csv> [max time us]"""
import csv, glob, re, sys
d = sys.argv[1]
tmax = float(sys.argv[2]) if len(sys.argv) > 2 else 1700.0
path = glob.glob(d + "/**/*_kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n); n = re.sub(r"^void ", "", n)
    if "rocprim" in n:
        k = re.search(r"(radix_sort\w*|reduce_by_key\w*|scan\w*|lookback\w*|trivial_runs\w*|select\w*|partition\w*|histogram\w*|transform\w*)", n)
        return "rocprim::" + (k.group(1) if k else "kernel")
    m = re.match(r"([\w:]+(<\w+)?)", n); return m.group(1) if m else n[:40]
starts = [i for i, r in enumerate(rows) if "k_emit_rows" in r["Kernel_Name"]]
step_starts = [starts[0]]
for a, b in zip(starts, starts[1:]):
    if int(rows[b]["Start_Timestamp"]) - int(rows[a]["End_Timestamp"]) > 3e6: step_starts.append(b)
sel = rows[step_starts[-1]:]
t0 = int(sel[0]["Start_Timestamp"])
queues = {}
print("%9s %9s  %5s %6s %6s  %9s  kernel" % ("start", "end", "queue", "stream", "thread", "grid"))
for r in sel:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if (s - t0) / 1e3 > tmax: break
    q = r.get("Queue_Id", "?"); st = r.get("Stream_Id", "?")
    queues.setdefault(q, set()).add(st)
    print("%9.1f %9.1f  %5s %6s %6s  %9s  %s" % ((s - t0) / 1e3, (e - t0) / 1e3, q, st, r.get("Thread_Id", "?")[-4:], r.get("Grid_Size_X", "?"), short(r["Kernel_Name"])))
print()
for q, st in sorted(queues.items()):
    print("queue %s carries streams %s" % (q, sorted(st)))
# the whole run: which streams every hardware queue carried, with launch counts
whole = {}
for r in rows:
    whole.setdefault((r.get("Queue_Id", "?"), r.get("Stream_Id", "?")), [0, r["Kernel_Name"]])[0] += 1
print()
for (q, st), (n, first) in sorted(whole.items()):
    print("whole run: queue %s stream %s: %6d launches, first %s" % (q, st, n, short(first)))

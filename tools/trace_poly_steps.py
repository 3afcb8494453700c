#!/usr/bin/env python3
"""The launches of one outer iteration of the polynomial PCG, by position, from a rocprofv3 kernel trace: between a
k_pcg_resid and the next k_pcg_xp come the Chebyshev steps 1..d-1 (k_apply_march3d with the epilogue), in front of the
k_pcg_resid the full apply.  Levels are told apart by the duration of k_pcg_resid (the finest level's is the long one).
usage: trace_poly_steps.py <dir with *_kernel_trace.csv>   -> markdown table on stdout"""
import csv, glob, sys
d = sys.argv[1]
path = glob.glob(d + "/**/*_kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
def dur(r): return (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
res = [dur(r) for r in rows if "k_pcg_resid" in r["Kernel_Name"]]
if not res:
    print("no k_pcg_resid launches in the trace"); sys.exit(0)
cut = 0.5 * max(res)           # finest level: the long launches
acc = {}
def add(key, v): acc.setdefault(key, []).append(v)
i = 0
while i < len(rows):
    r = rows[i]
    if "k_pcg_resid" in r["Kernel_Name"] and dur(r) > cut:
        # the full apply right in front of it
        j = i - 1
        while j >= 0 and "k_apply_march3d" not in rows[j]["Kernel_Name"] and i - j < 4: j -= 1
        if j >= 0 and "k_apply_march3d" in rows[j]["Kernel_Name"]: add("full apply (in front of k_pcg_resid)", dur(rows[j]))
        add("k_pcg_resid", dur(r))
        k, step = i + 1, 0
        while k < len(rows) and "k_pcg_xp" not in rows[k]["Kernel_Name"] and k - i < 12:
            if "k_apply_march3d" in rows[k]["Kernel_Name"]:
                step += 1
                add("Chebyshev step %d" % step, dur(rows[k]))
            k += 1
        if k < len(rows) and "k_pcg_xp" in rows[k]["Kernel_Name"]: add("k_pcg_xp", dur(rows[k]))
        i = k
    i += 1
print("| launch of an outer iteration (finest level) | launches | avg us | min | max |")
print("|---|---|---|---|---|")
for key in ["full apply (in front of k_pcg_resid)", "k_pcg_resid"] + sorted(k for k in acc if k.startswith("Chebyshev")) + ["k_pcg_xp"]:
    v = acc.get(key, [])
    if v: print("| %s | %d | %.1f | %.1f | %.1f |" % (key, len(v), sum(v) / len(v), min(v), max(v)))

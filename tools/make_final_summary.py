#!/usr/bin/env python3
"""Writes profiles/r1_final_summary.md from the artefacts of one gpurun call (see profiles/README.md):
gpurun_out/prof_stats (rocprofv3 --kernel-trace --stats of bench.py), profiles/r1_bench_n1.json, profiles/r1_traffic_*.json."""
import csv
import glob
import json
import os
import subprocess
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.chdir(ROOT)
stats = subprocess.run([sys.executable, "tools/prof_summary.py", "profiles/r1_final_kernel_stats.csv", "x"],
                       capture_output=True, text=True).stdout.split("\n", 4)[4]
rows = list(csv.DictReader(open(glob.glob("gpurun_out/prof_stats/*/*_kernel_trace.csv")[0])))
acc = defaultdict(list)
for r in rows:
    for k in ("k_apply_march3d", "k_cg_resid_f", "k_cg_xp_f"):
        if k in r["Kernel_Name"]:
            acc[(k, int(r["Grid_Size_X"]))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
lev = ""
fine = {}
for (k, g), v in sorted(acc.items(), key=lambda kv: (kv[0][0], -kv[0][1])):
    w = [x for x in v if x > 0.5 * max(v)]
    if len(w) < 5:
        continue
    fine.setdefault(k, (sum(w) / len(w), min(w), max(w)))
    lev += "| `%s` | %d | %d | %d | %.1f | %.1f | %.1f |\n" % (k, g, len(v), len(w), sum(w) / len(w), min(w), max(w))
b = json.load(open("profiles/r1_bench_n1.json"))
ta, tx, tr = (json.load(open("profiles/r1_traffic_%s.json" % n)) for n in ("apply_c4", "cg_xp", "cg_resid"))


def sp(x):
    return format(int(x), ",").replace(",", " ")


def row(name, t, exp):
    return "| %s | %s KB | %.1f MB | %s KB = %.1f MB | **%.1f MB** | %s |\n" % (
        name, sp(t["fetch_size_kb_raw"]), t["read_bytes_corrected"] / 1e6, sp(t["write_size_kb"]), t["write_bytes"] / 1e6,
        t["traffic_bytes"] / 1e6, exp)


out = """# Round 1 final: rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 5 --warmup 2 --cpu-side 0

Config 4 (256^3 fp32, 1 M value constraints), 2 coarser levels (bench default); launches of all three levels are pooled per kernel
name: the finest-level launches are the long ones (k_apply_march3d %.0f-%.0f us, k_cg_xp_f %.0f us, k_cg_resid_f %.0f us);
the coarsest level (64^3) holds all 10^6 points in 2.6e5 cells, so its apply is all cell work (the `..., true>` = PACK instantiation).
Bench line of the same build without the profiler: profiles/r1_bench_n1.json (%.2e lattice points/s, %.1f ms per step,
`roofline.launch_ms` %.1f us from HIP events inside the timed region).

| kernel | calls | avg us | min us | max us | %% |
|---|---|---|---|---|---|
%s
## The same trace split by launch size (the kernel trace of the same run; grid = threads)

rocprofv3's statistics pool the levels of the cascade under one kernel name.  Per level (launches that exited
at once on the stop flag -- below half the longest -- left out of the averages):

| kernel | grid | launches | working | avg us | min | max |
|---|---|---|---|---|---|---|
%s
The finest level is the largest grid of each kernel: its `k_apply_march3d` average is what `bench.py` reports as
`roofline.launch_ms` (HIP events on the solver stream, every 4th apply of the timed region; the two clocks differ by a few us).

## HBM traffic (rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in two separate passes of `bench.py --steps 1 --warmup 0`)

Finest-level launches only (`tools/pmc_traffic.py` drops the launches that exited on the stop flag and the
coarse levels: values below half the maximum).  FETCH_SIZE is doubled for 16-B/lane streams as
MI355X_MICROARCH.md prescribes for gfx950; the correction is validated in the same run by the two vector
kernels whose traffic is known exactly.

| kernel | FETCH_SIZE raw | reads (x2) | WRITE_SIZE | traffic | expected |
|---|---|---|---|---|---|
""" % (fine["k_apply_march3d"][1], fine["k_apply_march3d"][2], fine["k_cg_xp_f"][0], fine["k_cg_resid_f"][0], b["value"],
       b["ms_per_step"], b["roofline"]["launch_ms"] * 1e3, stats, lev)
out += row("`k_apply_march3d` (config 4, 970 420 cells)", ta,
           "172.4 MB algorithmic (%.2fx: 4 overlap planes per 22-plane chunk, halo ring misses, border cells listed twice, "
           "the second records of two-row cells)" % (ta["traffic_bytes"] / 172397152.0))
out += row("`k_cg_xp_f` (reads x, p, r, Dinv; writes x, p)", tx, "6 x 67.1 = 402.7 MB")
out += row("`k_cg_resid_f` (reads r, q, Dinv; writes r)", tr, "4 x 67.1 = 268.4 MB")
out += """
`profiles/r1_traffic_apply_c4.json`, `r1_traffic_cg_xp.json`, `r1_traffic_cg_resid.json` hold the raw numbers;
`bench.py` reports the first as `roofline.traffic`.
"""
open("profiles/r1_final_summary.md", "w").write(out)
print(out)

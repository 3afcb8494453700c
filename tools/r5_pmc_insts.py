#!/usr/bin/env python3
"""Summary of tools/r5_pmc_insts.sh: per kernel (finest-level launches only: the largest grid of the name) the mean of every counter."""
import csv, glob, sys, collections, re
d = sys.argv[1]
want = {"cheb": "k_apply_march3d<float, false, true, false, 32, false, true, false>",
        "cheb_first": "k_apply_march3d<float, false, true, false, 32, false, true, true>",
        "apply64": "k_apply_march3d<double, false, true, true, 32, false, false, false>",
        "fused32": "k_apply_march3d<float, false, true, true, 32, false, true, false>",
        "step_mixed": "k_mg_step_mixed"}
res = collections.defaultdict(dict)
for path in sorted(glob.glob(d + "/p*/**/*counter_collection.csv", recursive=True)):
    rows = list(csv.DictReader(open(path)))
    for key, name in want.items():
        sel = [r for r in rows if name in r["Kernel_Name"]]
        if not sel: continue
        gmax = max(int(r["Grid_Size"]) for r in sel)
        sel = [r for r in sel if int(r["Grid_Size"]) == gmax]
        by = collections.defaultdict(list)
        for r in sel: by[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for c, v in by.items(): res[key][c] = (sum(v) / len(v), len(v), gmax)
for key in want:
    print("==", key, want[key])
    for c, (m, n, g) in sorted(res[key].items()):
        print("   %-24s %16.0f   (%d launches, grid %d)" % (c, m, n, g))

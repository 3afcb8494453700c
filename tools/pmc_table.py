#!/usr/bin/env python3
"""Scratch: mean counter value per kernel (substring match) from rocprofv3 --pmc csv files.
usage: pmc_table.py <needle> <csv> [<csv> ...]"""
import csv
import sys
from collections import defaultdict
needle = sys.argv[1]
acc = defaultdict(list)
for path in sys.argv[2:]:
    for r in csv.DictReader(open(path)):
        if needle in r["Kernel_Name"]:
            acc[(r["Kernel_Name"][:60], r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(acc.items()):
    print("%-62s %-24s n=%3d mean=%.4g" % (k, c, len(v), sum(v) / len(v)))

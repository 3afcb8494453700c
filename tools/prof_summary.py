#!/usr/bin/env python3
"""Condenses a rocprofv3 `*_kernel_stats.csv` into a short markdown table (for profiles/)."""
import csv
import re
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([\w:]+(?:<[^()]*?>)?)\(", name)
    if "rocprim" in name:
        k = re.search(r"(radix_sort\w*|reduce_by_key\w*|scan\w*|lookback\w*|trivial_runs\w*)", name)
        return "rocprim::" + (k.group(1) if k else "kernel")
    return m.group(1) if m else name[:70]


def main(path, title):
    rows = list(csv.DictReader(open(path)))
    print("# %s\n" % title)
    print("| kernel | calls | avg us | min us | max us | % |")
    print("|---|---|---|---|---|---|")
    for r in rows[:28]:
        print("| `%s` | %s | %.1f | %.1f | %.1f | %.2f |" % (short(r["Name"]), r["Calls"], float(r["AverageNs"]) / 1e3,
                                                     float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3,
                                                     float(r["Percentage"])))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else sys.argv[1])

#!/usr/bin/env python3
"""Per-kernel, per-launch-size table of a rocprofv3 kernel trace (the levels of a cascade share kernel names; the
grid size tells them apart).  usage: trace_by_grid.py <dir with *_kernel_trace.csv> [min share of the longest launch]"""
import csv
import glob
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    if "rocprim" in name:
        k = re.search(r"(radix_sort\w*|reduce_by_key\w*|scan\w*|lookback\w*|trivial_runs\w*|select\w*|partition\w*)", name)
        return "rocprim::" + (k.group(1) if k else "kernel")
    m = re.match(r"([\w:]+(?:<[^()]*?>)?)\(", name)
    return m.group(1) if m else name[:60]


def main(d, share=0.5):
    path = glob.glob(d + "/**/*_kernel_trace.csv", recursive=True)[0]
    rows = list(csv.DictReader(open(path)))
    acc = defaultdict(list)
    t0 = min(int(r["Start_Timestamp"]) for r in rows)
    t1 = max(int(r["End_Timestamp"]) for r in rows)
    for r in rows:
        acc[(short(r["Kernel_Name"]), int(r["Grid_Size_X"]))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    tot = sum(sum(v) for v in acc.values())
    print("kernel time %.2f ms in a trace of %.2f ms" % (tot / 1e3, (t1 - t0) / 1e6))
    print("| kernel | grid | launches | working | avg us | min | max | total ms | % |")
    print("|---|---|---|---|---|---|---|---|---|")
    for (k, g), v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
        if sum(v) < 0.003 * tot:
            continue
        w = [x for x in v if x > share * max(v)]
        print("| `%s` | %d | %d | %d | %.1f | %.1f | %.1f | %.2f | %.1f |" % (k, g, len(v), len(w), sum(w) / len(w), min(w), max(w),
                                                                          sum(v) / 1e3, 100 * sum(v) / tot))


if __name__ == "__main__":
    main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 0.5)

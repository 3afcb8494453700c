#!/usr/bin/env python3
"""HBM traffic launch by launch from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; KB) of the same deterministic command:
the launches of the kernels whose name contains one of the needles, with a given grid size, in dispatch order, and the mean
of every group of launches that move the same bytes (+-3 %).
usage: pmc_by_launch.py <fetch.csv> <write.csv> <grid> <needle> [<needle> ...]
reads = 2 x FETCH_SIZE (MI355X_MICROARCH.md: gfx950 reports half the bytes of 16 B / lane streams), writes = WRITE_SIZE."""
import csv
import sys


def rows(path, counter, grid, needles):
    out = []
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter or int(r["Grid_Size"]) != grid:
            continue
        for k, n in enumerate(needles):
            if n in r["Kernel_Name"]:
                out.append((int(r["Dispatch_Id"]), k, float(r["Counter_Value"])))
                break
    out.sort()
    return out


def main():
    fetch, write, grid = sys.argv[1], sys.argv[2], int(sys.argv[3])
    needles = sys.argv[4:]
    f = rows(fetch, "FETCH_SIZE", grid, needles)
    w = rows(write, "WRITE_SIZE", grid, needles)
    assert len(f) == len(w) and all(a[1] == b[1] for a, b in zip(f, w)), "the two passes launched different kernels"
    print("# launch needle read_MB write_MB total_MB")
    groups = []
    for i, (a, b) in enumerate(zip(f, w)):
        rd, wr = 2.0 * a[2] * 1024 / 1e6, b[2] * 1024 / 1e6
        print("%4d %d %8.1f %8.1f %8.1f" % (i, a[1], rd, wr, rd + wr))
        for g in groups:
            if g[0] == a[1] and abs(g[1] / g[3] - rd) <= 0.03 * rd + 0.5 and abs(g[2] / g[3] - wr) <= 0.03 * wr + 0.5:
                g[1] += rd; g[2] += wr; g[3] += 1
                break
        else:
            groups.append([a[1], rd, wr, 1])
    print("# groups of launches that move the same bytes: needle launches read_MB write_MB total_MB")
    for g in sorted(groups, key=lambda g: (g[0], g[1] + g[2])):
        print("# %d %4d %8.1f %8.1f %8.1f   (%s)" % (g[0], g[3], g[1] / g[3], g[2] / g[3], (g[1] + g[2]) / g[3], needles[g[0]][-40:]))


if __name__ == "__main__":
    main()

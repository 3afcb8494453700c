#!/usr/bin/env python3
"""Per-launch HBM traffic of one kernel from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; KB units).

usage: pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <kernel substring> <out.json> [share]
(launches whose counter is below `share` (default 0.5) of the largest are dropped: early exits on the stop flag, coarser
levels, and -- with 0.9 -- the 4-pass first step of a Chebyshev polynomial beside its 5-pass steps)
gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE reports exactly half the bytes of a wide (16 B/lane)
coalesced stream; WRITE_SIZE is exact.  Both the raw and the doubled read figure are recorded; `traffic`
uses the doubled reads because the kernel's plane loads are 16 B/lane streams."""
import csv
import json
import sys


SHARE = 0.5


def avg(path, counter, needle):
    needles = needle.split("|")           # several kernel names: the mean over the launches of all of them
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(path))
            if any(n in r["Kernel_Name"] for n in needles) and r["Counter_Name"] == counter]
    big = [v for v in vals if v > SHARE * max(vals)]        # drop the launches that exited on the done flag
    return sum(big) / len(big), len(big)


def main():
    global SHARE
    fetch_csv, write_csv, needle, out = sys.argv[1:5]
    if len(sys.argv) > 5:
        SHARE = float(sys.argv[5])
    f, nf = avg(fetch_csv, "FETCH_SIZE", needle)
    w, nw = avg(write_csv, "WRITE_SIZE", needle)
    res = {"kernel": needle, "launches": [nf, nw], "fetch_size_kb_raw": f, "write_size_kb": w,
           "read_bytes_corrected": 2.0 * f * 1024.0, "write_bytes": w * 1024.0,
           "traffic_bytes": 2.0 * f * 1024.0 + w * 1024.0}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()

#!/bin/bash
# calibrates FETCH_SIZE / WRITE_SIZE per access width (tools/micro/fetch_probe.hip)
cd "$GRAFT_REPO_ROOT"; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5_fetch_probe; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 tools/micro/fetch_probe.hip -o $O/fetch_probe || exit 1
cd /tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/f -- $O/fetch_probe > $O/f.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/w -- $O/fetch_probe > $O/w.log 2>&1 || exit 1
cd $R
python3 - <<PY
import csv, glob
for tag, counter in (("f", "FETCH_SIZE"), ("w", "WRITE_SIZE")):
    path = glob.glob("$O/%s/**/*counter_collection.csv" % tag, recursive=True)[0]
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter and ("k_read" in r["Kernel_Name"] or "k_write" in r["Kernel_Name"]):
            kb = float(r["Counter_Value"])
            print("%-11s %-40s %12.0f KB = %.3f of the 262144 KB stream" % (counter, r["Kernel_Name"][:40], kb, kb / 262144.0))
PY
rm -rf $O/f $O/w $O/fetch_probe

#!/bin/bash
# profiles/r5_traffic_cheb_by_launch_c4.txt: the finest level's smoother launches of ONE config-4 step, launch by launch
cd "$GRAFT_REPO_ROOT"; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5_bl; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
ARGS="--steps 1 --warmup 0 --datasets 1 --cpu-side 0 --no-accuracy --no-cold"
cd /tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/f -- python3 $R/bench.py $ARGS > $O/f.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/w -- python3 $R/bench.py $ARGS > $O/w.log 2>&1 || exit 1
cd $R
F=$(ls $O/f/*/*counter_collection.csv | head -1); W=$(ls $O/w/*/*counter_collection.csv | head -1)
python3 tools/pmc_by_launch.py $F $W 262144 "k_apply_march3d<float, false, true, false, 32, false, true, true>" "k_apply_march3d<float, false, true, false, 32, false, true, false>" > $O/by_launch.txt
python3 tools/pmc_by_launch.py $F $W 196608 "k_apply_march3d<double, false, true, true, 32, false, false, false>" > $O/by_launch_apply.txt
rm -rf $O/f $O/w
grep "^#" $O/by_launch.txt; grep "^#" $O/by_launch_apply.txt

#!/bin/bash
# kernel-by-kernel timeline of one step of the accurate leg (fp64 CG + fp32 V-cycle, config 4): tools/r3_trace_accurate.sh
cd "$GRAFT_REPO_ROOT"; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/acc_trace; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
cd /tmp
NOREF=1 MODES="mgmix3:f64:3:1:1:0:0:1e-7" rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 $R/tools/exp.py > $O/trace.log 2>&1
cd $R
python3 tools/trace_list.py $O/trace 20000 > $O/list.txt
python3 tools/trace_gaps.py $O/trace 15 > $O/gaps.txt
rm -rf $O/trace
echo "launches longer than 120 us:"; awk '$5 > 120' $O/list.txt | head -40; tail -3 $O/gaps.txt

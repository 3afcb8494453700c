#!/bin/bash
# Round-6 evidence for profiles/: bench line, rocprofv3 kernel statistics and per-grid tables, PMC traffic (separate passes).
# usage (on the GPU box): bash tools/r6_evidence.sh <part>     part: c4 | c4_512 | c4_fast | c5 | c3 | c2
cd "$GRAFT_REPO_ROOT"
R=$GRAFT_REPO_ROOT
part=$1
O=$R/gpurun_out/r6_ev_$part
rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
PMC=1
case $part in
  c4)      ARGS="--cpu-side 0" ;;
  c4_fast) ARGS="--cpu-side 0 --fast" ;;
  c4_512)  ARGS="--cpu-side 0 --side 512 --points 8000000"; PMC=0 ;;
  c5)      ARGS="--cpu-side 0 --config 5 --no-accuracy"; PMC=0 ;;
  c3)      ARGS="--cpu-side 0 --config 3 --no-accuracy"; PMC=0 ;;
  c2)      ARGS="--cpu-side 0 --config 2 --no-accuracy"; PMC=0 ;;
esac
python bench.py --steps 3 --warmup 1 $ARGS --no-roofline-512 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py --steps 3 --warmup 1 $ARGS --no-accuracy --no-cold --no-host-io --no-roofline-512 > $O/trace.log 2>&1; echo "trace rc=$?"
python3 $R/tools/trace_by_grid.py $O/trace > $O/by_grid.md
python3 $R/tools/trace_list.py $O/trace 60000 > $O/step_timeline.txt
python3 $R/tools/trace_gaps.py $O/trace 30 > $O/step_gaps.txt
cp $(ls $O/trace/*/*kernel_stats.csv | head -1) $O/kernel_stats.csv
python3 $R/tools/prof_summary.py $O/kernel_stats.csv "r6 $part: rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 3 --warmup 1 $ARGS --no-accuracy --no-cold --no-host-io --no-roofline-512" > $O/kernel_stats.md
if [ $PMC = 1 ]; then
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py --steps 1 --warmup 0 $ARGS --no-accuracy --no-cold --no-host-io --no-roofline-512 > $O/pmc_fetch.log 2>&1; echo "fetch rc=$?"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 $R/bench.py --steps 1 --warmup 0 $ARGS --no-accuracy --no-cold --no-host-io --no-roofline-512 > $O/pmc_write.log 2>&1; echo "write rc=$?"
cd $R
F=$(ls $O/pmc_fetch/*/*counter_collection.csv | head -1); W=$(ls $O/pmc_write/*/*counter_collection.csv | head -1)
if [ $part = c4 ]; then
  python3 tools/pmc_traffic.py $F $W "k_apply_march3d<double, false, true, true, 32, false, false, false>" $O/traffic_apply.json
else
  python3 tools/pmc_traffic.py $F $W "k_apply_march3d<float, false, true, true, 32, false, false, false>" $O/traffic_apply.json
fi
# all Chebyshev steps of the finest level: the first (operand formed on load: the last template flag) and the others
python3 tools/pmc_traffic.py $F $W "k_apply_march3d<float, false, true, false, 32, false, true, false>|k_apply_march3d<float, false, true, false, 32, false, true, true>" $O/traffic_cheb.json 0.3
python3 tools/pmc_traffic.py $F $W "k_mg_step_mixed" $O/traffic_mg_step_mixed.json || true
fi
cd $R
rm -rf $O/pmc_fetch $O/pmc_write $O/trace
python3 tools/bench_brief.py $O/bench.json; head -24 $O/kernel_stats.md | cut -c1-200

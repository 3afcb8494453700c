#!/bin/bash
# Round-3 evidence for profiles/: bench line, rocprofv3 kernel statistics and trace splits, PMC traffic (separate passes).
# usage (on the GPU box): bash tools/r3_evidence.sh <part>     part: c4 | c4_512 | c5 | c3 | c2 | c4acc
cd "$GRAFT_REPO_ROOT"
R=$GRAFT_REPO_ROOT
part=$1
O=$R/gpurun_out/r3_ev_$part
mkdir -p $O
export TMPDIR=/tmp
PMC=1
case $part in
  c4)     ARGS="--cpu-side 0" ;;
  c4_512) ARGS="--cpu-side 0 --side 512 --points 8000000 --no-accuracy" ;;
  c5)     ARGS="--cpu-side 0 --config 5 --no-accuracy" ;;
  c3)     ARGS="--cpu-side 0 --config 3 --no-accuracy"; PMC=0 ;;
  c2)     ARGS="--cpu-side 0 --config 2 --no-accuracy"; PMC=0 ;;
esac
python bench.py --steps 3 --warmup 1 $ARGS > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py --steps 3 --warmup 1 $ARGS --no-accuracy --no-cold > $O/trace.log 2>&1; echo "trace rc=$?"
python3 $R/tools/trace_by_grid.py $O/trace > $O/by_grid.md
python3 $R/tools/trace_poly_steps.py $O/trace > $O/poly_steps.md
cp $(ls $O/trace/*/*kernel_stats.csv | head -1) $O/kernel_stats.csv
python3 $R/tools/prof_summary.py $O/kernel_stats.csv "r3 $part: rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 3 --warmup 1 $ARGS --no-accuracy --no-cold" > $O/kernel_stats.md
if [ $PMC = 1 ]; then
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py --steps 1 --warmup 0 $ARGS --no-accuracy --no-cold > $O/pmc_fetch.log 2>&1; echo "fetch rc=$?"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 $R/bench.py --steps 1 --warmup 0 $ARGS --no-accuracy --no-cold > $O/pmc_write.log 2>&1; echo "write rc=$?"
cd $R
F=$(ls $O/pmc_fetch/*/*counter_collection.csv | head -1); W=$(ls $O/pmc_write/*/*counter_collection.csv | head -1)
if [ $part = c5 ]; then
  python3 tools/pmc_traffic.py $F $W "k_apply_march3d<double, false, true, true" $O/traffic_apply.json
  python3 tools/pmc_traffic.py $F $W "k_apply_march3d<float, false, true, false, 32, false, true" $O/traffic_cheb.json 0.3
else
  python3 tools/pmc_traffic.py $F $W "k_apply_march3d<float, false, true, true, 32, false, false, false>" $O/traffic_apply.json
  # all Chebyshev steps of the finest level: the first (operand formed on load: the last template flag) and the others
  python3 tools/pmc_traffic.py $F $W "k_apply_march3d<float, false, true, false, 32, false, true, false>|k_apply_march3d<float, false, true, false, 32, false, true, true>" $O/traffic_cheb.json 0.3
  python3 tools/pmc_traffic.py $F $W "k_pcg_xp" $O/traffic_pcg_xp.json
  python3 tools/pmc_traffic.py $F $W "k_pcg_resid" $O/traffic_pcg_resid.json
fi
fi
cd $R
# keep what is judged small: drop the raw per-dispatch tables after the summaries exist
rm -rf $O/pmc_fetch $O/pmc_write $O/trace
python3 tools/bench_brief.py $O/bench.json; cat $O/poly_steps.md; head -16 $O/kernel_stats.md

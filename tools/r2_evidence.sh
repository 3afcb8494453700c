#!/bin/bash
# Round-2 evidence for profiles/: kernel traces and PMC traffic of the bench step (config 4 at 256^3), of its 512^3
# scale-up, and of config 5.  usage (on the GPU box): bash tools/r2_evidence.sh <part>   part: c4 | c4_512 | c5
cd "$GRAFT_REPO_ROOT"
R=$GRAFT_REPO_ROOT
part=$1
O=$R/gpurun_out/r2_ev_$part
mkdir -p $O
export TMPDIR=/tmp
case $part in
  c4)     ARGS="--cpu-side 0" ;;
  c4_512) ARGS="--cpu-side 0 --side 512 --points 8000000 --no-accuracy" ;;
  c5)     ARGS="--cpu-side 0 --config 5 --no-accuracy" ;;
esac
python bench.py --steps 3 --warmup 1 $ARGS > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py --steps 3 --warmup 1 $ARGS --no-accuracy > $O/trace.log 2>&1; echo "trace rc=$?"
python3 $R/tools/trace_by_grid.py $O/trace > $O/by_grid.md
python3 $R/tools/trace_poly_steps.py $O/trace > $O/poly_steps.md
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py --steps 1 --warmup 0 $ARGS --no-accuracy > $O/pmc_fetch.log 2>&1; echo "fetch rc=$?"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 $R/bench.py --steps 1 --warmup 0 $ARGS --no-accuracy > $O/pmc_write.log 2>&1; echo "write rc=$?"
cd $R
F=$(ls $O/pmc_fetch/*/*counter_collection.csv | head -1); W=$(ls $O/pmc_write/*/*counter_collection.csv | head -1)
if [ $part = c5 ]; then
  python3 tools/pmc_traffic.py $F $W "k_apply_march3d<double, false, true, true" $O/traffic_apply.json
  python3 tools/pmc_traffic.py $F $W "k_apply_march3d<float, false, true, true" $O/traffic_apply_f32.json
else
  python3 tools/pmc_traffic.py $F $W "k_apply_march3d<float, false, true, true, 32, false, false, false>" $O/traffic_apply.json
  python3 tools/pmc_traffic.py $F $W "k_apply_march3d<float, false, true, false, 32, false, true, false>" $O/traffic_cheb.json 0.9
  python3 tools/pmc_traffic.py $F $W "k_pcg_xp" $O/traffic_pcg_xp.json
  python3 tools/pmc_traffic.py $F $W "k_pcg_resid" $O/traffic_pcg_resid.json
fi
# keep what is judged small: drop the raw per-dispatch counter tables after the summaries exist
cp $(ls $O/trace/*/*kernel_stats.csv | head -1) $O/kernel_stats.csv
rm -rf $O/pmc_fetch $O/pmc_write $O/trace
cat $O/bench.json | cut -c1-600; cat $O/by_grid.md | head -14

#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=$GRAFT_REPO_ROOT/gpurun_out/r2_nosamp; mkdir -p $O
for env in "FI_DUMMY=1" "FI_NO_SAMPLES=1"; do
env $env FI_HIP_LIB=$GRAFT_REPO_ROOT/exp_libs/libfi_tb.so timeout -k 10 300 python bench.py --steps 5 --warmup 2 --cpu-side 0 --no-accuracy > $O/bench.json 2> $O/bench.err; python -c "
import json; d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); c=d['config']; print('[$env]', round(d['ms_per_step'],3), '%.4g' % d['value'], c['iterations'], c['coarse_iterations'], round(c['assemble_ms'],2), round(c['solve_ms'],2))"
done
cd /tmp; export TMPDIR=/tmp
export FI_NO_SAMPLES=1 FI_HIP_LIB=$GRAFT_REPO_ROOT/exp_libs/libfi_tb.so
rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --cpu-side 0 --no-accuracy > $O/trace.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/trace_list.py $O/trace 20000 > $O/list.txt
rm -rf $O/trace

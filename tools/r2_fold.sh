#!/bin/bash
# A/B: partial sums by a one-block kernel (shipped) vs folded into the consumers (timing build, FI_POLY_FOLDED).
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2_fold; mkdir -p $O
run() { python -c "
import json; d=json.loads(open('$O/b.json').read().strip().splitlines()[-1]); c=d['config']; print('$1', round(d['ms_per_step'],3), '%.4g' % d['value'], c['iterations'], c['coarse_iterations'], round(c['assemble_ms'],2), round(c['solve_ms'],2))"; }
timeout -k 10 200 python bench.py --steps 5 --warmup 2 --cpu-side 0 --no-accuracy > $O/b.json 2> $O/b.err && run shipped
FI_HIP_LIB=$GRAFT_REPO_ROOT/exp_libs/libfi_tb.so FI_POLY_FOLDED=1 timeout -k 10 200 python bench.py --steps 5 --warmup 2 --cpu-side 0 --no-accuracy > $O/b.json 2> $O/b.err && run folded
FI_HIP_LIB=$GRAFT_REPO_ROOT/exp_libs/libfi_tb.so timeout -k 10 200 python bench.py --steps 5 --warmup 2 --cpu-side 0 --no-accuracy > $O/b.json 2> $O/b.err && run tb-unfolded

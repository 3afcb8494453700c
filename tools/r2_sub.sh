#!/bin/bash
# Experiment: the cascade's coarse level built from a subsample of the points (timing build).
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2_sub; mkdir -p $O
for s in 1 2 4 8; do
  FI_HIP_LIB=$GRAFT_REPO_ROOT/exp_libs/libfi_tb.so FI_COARSE_SUBSAMPLE=$s timeout -k 10 200 python bench.py --steps 5 --warmup 2 --cpu-side 0 --no-accuracy > $O/b$s.json 2> $O/b$s.err && python -c "
import json; d=json.loads(open('$O/b$s.json').read().strip().splitlines()[-1]); c=d['config']; print('sub $s', round(d['ms_per_step'],2), c['iterations'], c['coarse_iterations'], round(c['assemble_ms'],2), round(c['solve_ms'],2))"
done

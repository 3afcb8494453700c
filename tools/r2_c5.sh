#!/bin/bash
# V-cycle with the fused smoother: tests, then config 5 (and the unfused form beside it).
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2_c5; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_multilevel.py tests/test_gpu_slabs.py -x -q -m gpu > $O/tests.log 2>&1; tail -3 $O/tests.log
timeout -k 10 300 python bench.py --config 5 --steps 2 --warmup 1 --cpu-side 0 --no-accuracy > $O/bench_c5.json 2> $O/bench_c5.err && python -c "
import json; d=json.loads(open('$O/bench_c5.json').read().strip().splitlines()[-1]); print('fused  ', d['ms_per_step'], d['config']['iterations'], d['config']['solve_ms'])"
FI_NO_FUSED_SMOOTHER=1 timeout -k 10 300 python bench.py --config 5 --steps 2 --warmup 1 --cpu-side 0 --no-accuracy > $O/bench_c5u.json 2> $O/bench_c5u.err && python -c "
import json; d=json.loads(open('$O/bench_c5u.json').read().strip().splitlines()[-1]); print('unfused', d['ms_per_step'], d['config']['iterations'], d['config']['solve_ms'])"

#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2_two; mkdir -p $O
export FI_BENCH_ONE_DEVICE=1
for env in "FI_DUMMY=1" "FI_NO_OVERLAP=1" "FI_DUMMY=2"; do
env $env timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 tests/two_rank_worker.py > $O/out.txt 2> $O/err.txt
echo "== $env rc=$?"; grep "^RESULTS" $O/out.txt | python -c "
import sys, json
for l in sys.stdin:
    for r in json.loads(l[len('RESULTS '):]):
        print(r['case'], r['iterations'], r.get('iterations_one'), r['coarse_iterations'], r.get('coarse_iterations_one'), r['checksum'])"
done

#!/bin/bash
# round 2, experiment 1: where does V-cycle PCG spend its time on config 4 (vs the cascade + Jacobi-PCG bench default)
set -e
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r2_exp1
mkdir -p $O
python bench.py --steps 5 --warmup 2 --cpu-side 0 > $O/base.json 2> $O/base.err
for deg in 1 2 3; do
  for lev in 2 3 4; do
    FI_MG_DEGREE=$deg python bench.py --steps 3 --warmup 1 --cpu-side 0 --multigrid --levels $lev > $O/mg_d${deg}_l${lev}.json 2>> $O/mg.err
  done
done
cd /tmp
FI_MG_DEGREE=2 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_mg -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --cpu-side 0 --multigrid --levels 3 > $GRAFT_REPO_ROOT/$O/prof_mg.log 2>&1
echo done

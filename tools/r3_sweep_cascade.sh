#!/bin/bash
# sweep of the coarse-to-fine start's effort (experiment knobs FI_CASCADE_JAC / FI_CASCADE_MG): configs 3, 2, 5 and the accurate leg
cd "$GRAFT_REPO_ROOT"
for jm in "48 40" "48 8" "48 4" "0 8" "0 4" "0 2" "16 4" "16 2"; do
  set -- $jm
  export FI_CASCADE_JAC=$1 FI_CASCADE_MG=$2
  for c in 3 2 5; do
    python bench.py --config $c --steps 2 --warmup 1 --cpu-side 0 --no-accuracy --no-cold > gpurun_out/sw.json 2>/dev/null
    echo "jac $1 mg $2: $(python tools/bench_brief.py gpurun_out/sw.json | cut -d: -f2- | cut -c1-110)"
  done
  NOREF=1 MODES="mgmix3:f64:3:1:1:0:0:1e-7" python tools/exp.py 2>&1 | grep mgmix | sed "s/^/jac $1 mg $2 /" | cut -c1-170
done

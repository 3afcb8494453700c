#!/bin/bash
# GPU_MAX_HW_QUEUES: do the assembly's chains share hardware queues?  (profiles/r5_ablation.md section 14)
for q in 4 8 16; do
  for rep in 1 2; do
    echo "== GPU_MAX_HW_QUEUES=$q"
    GPU_MAX_HW_QUEUES=$q python bench.py --steps 30 --warmup 5 --cpu-side 0 --no-accuracy 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], {k: d[k] for k in d if 'phase' in k or 'cold' in k or 'assembl' in k})" || exit 1
  done
done

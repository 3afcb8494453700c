#!/bin/bash
# A/B: fp64 fused marching variants register-allocated for 2 (shipped: no spills) or 3 workgroups per CU (4 VGPRs spilled)
cd "$GRAFT_REPO_ROOT"
for v in shipped f64three; do
  if [ $v = shipped ]; then unset FI_HIP_LIB; else export FI_HIP_LIB=$GRAFT_REPO_ROOT/exp_libs/libfi_$v.so; fi
  for rep in 1 2; do
    NOREF=1 MODES="acc:f64:3:1:1:0:0:1e-7" python tools/exp.py 2>&1 | grep "^acc" | sed "s/^/$v /" | cut -c1-160
  done
  python bench.py --config 5 --steps 2 --warmup 1 --cpu-side 0 --no-accuracy --no-cold > gpurun_out/ab.json 2>/dev/null; echo "$v $(python tools/bench_brief.py gpurun_out/ab.json | cut -c1-200)"
done

#!/bin/bash
# chunk length of every marching launch (FI_ZC; the polynomial's steps pick 16 planes at 256^3 by themselves)
for zc in 0 16 22 26 32 43 64; do
  if [ $zc = 0 ]; then unset FI_ZC; else export FI_ZC=$zc; fi
  python bench.py --steps 20 --warmup 5 --cpu-side 0 --no-accuracy --no-cold 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('zc', '$zc', 'ms', round(d['ms_per_step'],3), 'cheb us', round(r.get('kernel_us', r.get('us',0)),2), 'frac', round(r['frac'],3), 'apply', round(d['roofline_apply']['frac'],3))" || exit 1
done

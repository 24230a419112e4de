#!/bin/bash
run() { python bench.py --cpu-samples 0 $NB 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$TAG', '$NB', round(d['ms_per_step'], 4), (d.get('single_stream') or {}).get('ms_per_step'))
" || exit 1; }
for r in 1 2; do
 for NB in "" "--no-breakdown"; do
  unset VV_NO_WHOLE; unset VV_CTW_PS
  TAG=whole_ps1 run
  export VV_CTW_PS=2; TAG=whole_ps2 run
  unset VV_CTW_PS; export VV_NO_WHOLE=1; TAG=halo run
 done
done

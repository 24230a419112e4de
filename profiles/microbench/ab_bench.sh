#!/bin/bash
# A/B of one env switch on the same box: profiles/microbench/ab_bench.sh VAR  -> runs bench.py with VAR unset, VAR=1, unset, VAR=1
v=$1; shift
for r in 1 2; do
  for m in off on; do
    if [ $m = on ]; then export $v=1; else unset $v; fi
    python bench.py --cpu-samples 0 "$@" 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$v', '$m', round(d['ms_per_step'], 4), d.get('single_stream', {}).get('ms_per_step'), d.get('layer_ms'))
" || exit 1
  done
done

for rep in 1 2; do
for ps in 1 2 4; do
  VV_CTW_PS=$ps python bench.py --steps 400 --warmup 50 --no-breakdown --cpu-samples 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ps', $ps, 'streams 3', round(d['ms_per_step'],4), round(d['value']))"
done
done
for st in 2 3; do for ps in 1 2; do VV_CTW_PS=$ps python bench.py --streams $st --steps 400 --warmup 50 --no-breakdown --cpu-samples 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ps', $ps, 'streams', $st, round(d['ms_per_step'],4), round(d['value']))"; done; done

#!/usr/bin/env python3
"""Turn rocprofv3 CSV output into the per-layer tables committed next to this file.

Several layers share one kernel instantiation (E2/E3/E4 are all igemm_kernel<bf16, CONV, 128, 128>), so the
stock `--stats` table averages them together.  A layer is identified here by (kernel name, grid size), which is
unique per layer at a fixed batch.

  summarize.py bygrid  <kernel_trace.csv>                  -> CSV on stdout: kernel, grid, calls, avg/min/max ns
  summarize.py traffic <fetch_cc.csv> <write_cc.csv> LAYER=substr:grid ...   -> JSON on stdout

`traffic` applies the gfx950 corrections of /opt/skills/guides/MI355X_MICROARCH.md (HBM section): FETCH_SIZE and
WRITE_SIZE are in KiB; FETCH_SIZE under-reports 16-B/lane streaming reads by 2x on gfx950 and is doubled;
WRITE_SIZE is taken as read.  The two counters come from separate `--pmc` passes.
"""
import csv
import json
import sys
from collections import defaultdict


def bygrid(path):
    acc = defaultdict(list)
    with open(path) as f:
        for r in csv.DictReader(f):
            grid = int(r['Grid_Size_X']) * int(r['Grid_Size_Y']) * int(r['Grid_Size_Z'])
            acc[(r['Kernel_Name'], grid)].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
    w = csv.writer(sys.stdout)
    w.writerow(['Kernel_Name', 'Grid_Size', 'Calls', 'TotalNs', 'AverageNs', 'MinNs', 'MaxNs'])
    for (k, g), v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
        w.writerow([k, g, len(v), sum(v), '%.1f' % (sum(v) / len(v)), min(v), max(v)])


def _counter(path, name):
    acc = defaultdict(list)
    with open(path) as f:
        for r in csv.DictReader(f):
            if r['Counter_Name'] == name:
                acc[(r['Kernel_Name'], int(r['Grid_Size']))].append(float(r['Counter_Value']))
    return acc


def traffic(fetch_csv, write_csv, specs):
    fe, wr = _counter(fetch_csv, 'FETCH_SIZE'), _counter(write_csv, 'WRITE_SIZE')
    out = {'config': 'bench.py bf16 B=256 D=32; rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes',
           'note': 'hbm_bytes_per_launch = 2*FETCH_SIZE_KB*1024 + WRITE_SIZE_KB*1024 (gfx950 FETCH_SIZE correction, see summarize.py)',
           'layers': {}}
    for spec in specs:
        layer, rest = spec.split('=')
        sub, grid = rest.rsplit(':', 1)
        grid = int(grid)
        fk = [k for k in fe if sub in k[0] and k[1] == grid]
        wk = [k for k in wr if sub in k[0] and k[1] == grid]
        if len(fk) != 1 or len(wk) != 1:
            raise SystemExit('layer %s: %d fetch / %d write kernels match %r grid %d' % (layer, len(fk), len(wk), sub, grid))
        f_kb = sum(fe[fk[0]]) / len(fe[fk[0]])
        w_kb = sum(wr[wk[0]]) / len(wr[wk[0]])
        out['layers'][layer] = {'kernel': fk[0][0], 'grid_threads': grid, 'launches': len(fe[fk[0]]),
                                'FETCH_SIZE_KB': f_kb, 'WRITE_SIZE_KB': w_kb,
                                'hbm_bytes_per_launch': int(2 * f_kb * 1024 + w_kb * 1024)}
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == '__main__':
    if len(sys.argv) >= 3 and sys.argv[1] == 'bygrid':
        bygrid(sys.argv[2])
    elif len(sys.argv) >= 4 and sys.argv[1] == 'traffic':
        traffic(sys.argv[2], sys.argv[3], sys.argv[4:])
    else:
        raise SystemExit(__doc__)

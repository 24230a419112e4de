"""Current-state table of DESIGN.md section 0 from the committed summaries of one round:
   python profiles/state_table.py r04  ->  markdown on stdout
layer -> kernel -> rocprofv3 average us (one batch at a time) -> achieved vs its roofline -> counter traffic vs algorithmic bytes ->
MFMA busy / wave cycles parked -> what limits it.  Every number is recomputable from profiles/<round>_*."""
import csv, json, os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), 'anytime-3d-reconstruction_amd'))
from voxvae import workload, synthetic as syn

rnd = sys.argv[1] if len(sys.argv) > 1 else 'r04'
what = sys.argv[2] if len(sys.argv) > 2 else 'eval'


def line_of(name):
    try:
        return json.loads(open(os.path.join(HERE, name)).read().strip().splitlines()[-1])
    except Exception:
        return None


if what == 'bench':
    print('| line (file under `profiles/`) | command | value | ms / step | notes |')
    print('|---|---|---|---|---|')
    spec = [('bf16_b256_bench_driver_form', '`python bench.py --steps 20 --warmup 5` (the driver\'s form)', 'reconstructions/s'),
            ('bf16_b256_bench_unprofiled', '`--steps 400 --warmup 50`', 'reconstructions/s'),
            ('ab_unfused_tail_bf16_b256_bench', '`VV_NO_POS_TAIL=1`, `--steps 400 --warmup 50` (round-3 chain, same box)', 'reconstructions/s'),
            ('ab_fused_tail_bf16_b256_bench', 'the same with the fused tail', 'reconstructions/s'),
            ('fp8_mid_b256_bench', '`--dtype fp8` (default policy `mid`: E2, E3, D3, D4 on e4m3fn)', 'reconstructions/s'),
            ('fp8_wide_b256_bench', '`--dtype fp8 --fp8-policy wide` (E2, D4)', 'reconstructions/s'),
            ('fp8_all_b256_bench', '`--dtype fp8 --fp8-policy all` (opt-in, over the IoU bar)', 'reconstructions/s'),
            ('bf16_d64_b64_bench', '`--voxel 64 --batch 64` (config 5\'s per-GPU shard, bf16)', 'reconstructions/s'),
            ('fp8_mid_d64_b64_bench', '`--voxel 64 --batch 64 --dtype fp8` (config 5, default policy `mid`)', 'reconstructions/s'),
            ('fp8_wide_d64_b64_bench', '`--voxel 64 --batch 64 --dtype fp8 --fp8-policy wide`', 'reconstructions/s'),
            ('fp8_most_d64_b64_bench', '`--voxel 64 --batch 64 --dtype fp8 --fp8-policy most` (everything but the encoder tail)', 'reconstructions/s'),
            ('fp8_all_d64_b64_bench', '`--voxel 64 --batch 64 --dtype fp8 --fp8-policy all`', 'reconstructions/s'),
            ('train_bf16_b256_bench_unprofiled', '`--mode train --steps 100 --warmup 20` (config 4\'s per-rank shape)', 'samples/s'),
            ('launcher_n1_eval', '`torch.distributed.run --nproc-per-node 1 bench.py --gpus 1` (RCCL group of one rank)', 'reconstructions/s'),
            ('launcher_n1_train', 'the same with `--mode train`', 'samples/s')]
    for key, cmd, unit in spec:
        d = line_of('%s_%s.json' % (rnd, key))
        if d is None:
            continue
        notes = []
        r = d.get('roofline')
        if r:
            notes.append('roofline %s: %.0f %s = %.3f of peak (%s, %.1f us)' % (r.get('bound'), r['achieved'], r['unit'], r['frac'], (r.get('kernel') or '')[:40], 1e3 * r.get('launch_ms', 0)))
        c = d.get('cpu_baseline')
        if c:
            notes.append('cpu_baseline %.1f %s on %d cores (%s)' % (c['value'], c['unit'], c['cores'], c['kind']))
        p_ = (d.get('parity') or {})
        if 'iou_delta' in p_:
            notes.append('IoU delta %.1e' % p_['iou_delta'])
        h = d.get('h2d_inclusive')
        if h:
            notes.append('host arrays %.0f k/s (bit-packed in %.0f k, + uint8 out %.0f k)' % (h['value'] / 1e3, h['bit_packed_input']['value'] / 1e3,
                                                                                        h['bit_packed_input']['uint8_occupancy_return']['value'] / 1e3))
        print('| `%s_%s.json` | %s | **%.1f k %s** | %.4f | %s |' % (rnd, key, cmd, d['value'] / 1e3, unit, d['ms_per_step'], '; '.join(notes)))
    sys.exit(0)

if what == 'train':
    rows = list(csv.DictReader(open(os.path.join(HERE, rnd + '_train_bf16_b256_kernel_stats.csv'))))
    steps = float(sum(int(r['Calls']) for r in rows if 'adam_multi_kernel' in r['Name']) or 25)      # one Adam launch per step
    groups = [('forward + data-gradient convolutions', ('conv_direct', 'ctw16', 'sd_kernel', 'pg_kernel', 'pg_reduce', 'first_conv', 'igemm', 'final_bce')),
              ('weight gradients', ('wgrad',)),
              ('BatchNorm: batch statistics (reduce + finalize)', ('bn_reduce_kernel<0', 'bn_stats_finalize')),
              ('BatchNorm: apply + activation', ('bn_act_fwd',)),
              ('BatchNorm: backward (reduce + finalize + apply)', ('bn_reduce_kernel<1', 'bn_bwd_finalize', 'bn_act_bwd')),
              ('Adam (one launch, 742 MB)', ('adam',)),
              ('weight packs (per step: the weights change)', ('pack', 'fold_bn')),]
    tot = sum(float(r['TotalDurationNs']) for r in rows)
    used = set()
    print('| share of the step (`%s_train_bf16_b256_kernel_stats.csv`, rocprofv3, %d steps) | ms / step | %% of GPU time | launches / step |' % (rnd, steps))
    print('|---|---|---|---|')
    for name, subs in groups:
        sel = [r for r in rows if any(s_ in r['Name'] for s_ in subs) and r['Name'] not in used]
        used |= set(r['Name'] for r in sel)
        t = sum(float(r['TotalDurationNs']) for r in sel)
        print('| %s | %.3f | %.1f | %.0f |' % (name, t / steps / 1e6, 100 * t / tot, sum(int(r['Calls']) for r in sel) / steps))
    rest = [r for r in rows if r['Name'] not in used]
    t = sum(float(r['TotalDurationNs']) for r in rest)
    print('| everything else (losses, latent, dense layers, copies) | %.3f | %.1f | %.0f |' % (t / steps / 1e6, 100 * t / tot, sum(int(r['Calls']) for r in rest) / steps))
    print('| **sum of kernel time** | **%.3f** | 100 | %.0f |' % (tot / steps / 1e6, sum(int(r['Calls']) for r in rows) / steps))
    sys.exit(0)

B = 256
cfg = syn.make_config(32, 64, True)
macs = {n: v for n, v, _ in workload.layer_macs(cfg)}
stats = {r['Name']: float(r['AverageNs']) / 1e3 for r in csv.DictReader(open(os.path.join(HERE, rnd + '_bf16_b256_kernel_stats.csv')))}
ctr = {}
for r in csv.DictReader(open(os.path.join(HERE, rnd + '_bf16_b256_counters.csv'))):
    ctr.setdefault(r['Kernel_Name'], r)
traffic = json.load(open(os.path.join(HERE, rnd + '_pmc_traffic.json')))['layers']

def find(d, sub):
    ks = [k for k in d if sub in k]
    if len(ks) != 1:
        raise SystemExit('%r matches %d kernels' % (sub, len(ks)))
    return d[ks[0]]

MB = 1e6
act = lambda side, ch: B * side ** 3 * ch * 2            # bf16 activation bytes
# layer, kernel substrings, bound, algorithmic bytes (input + output + weights read once), limiter
rows = [
    ('E1', ['first_conv_chain_kernel'], 'hbm', B * 32 ** 3 * 4 + act(16, 64), 'HBM: reads the float32 grid, writes the largest encoder activation'),
    ('E2', ['conv_direct16_kernel'], 'mfma', act(16, 64) + act(8, 128) + 64 * 64 * 128 * 2, 'matrix pipe at the clock the chip holds under load; one 8-wave workgroup per CU in lockstep (barrier per tap chunk, epilogue with the pipe idle)'),
    ('E3', ['sd_kernel<0>'], 'mfma', act(8, 128) + act(4, 256) + 64 * 128 * 256 * 2, 'its own MFMA + fragment-read stream (81 % of the kernel, r03 ablations); 23 % of the dense taps skipped'),
    ('E4', ['pg_kernel<0>'], 'mfma', act(4, 256) + 64 * 256 * 512 * 2, 'split-K: 29 MB of float32 partial sums written here and read back by the tail; K loop is 10 of the 26 us'),
    ('E4 sum + E5', ['lt_e5x_kernel'], 'latency', 8 * 512 * 128 * 2 + B * 128 * 4 * 16, 'one memory round trip over the 29 MB of partial sums'),
    ('a3-a7 + D0 + D1', ['lt_mid_kernel'], 'latency', 0, 'dependent round trips: 16 slabs summed, two K = 64 dense layers'),
    ('D2', ['pg_kernel<1>', 'pg_reduce_kernel<1'], 'mfma', act(2, 512) + act(4, 256) + 64 * 512 * 256 * 2, 'split-K slabs + the reduce launch (r03 ablations: two thirds of the layer)'),
    ('D3', ['sd_kernel<1>'], 'mfma', act(4, 256) + act(8, 128) + 64 * 256 * 128 * 2, 'as E3 (91 % of the kernel is its MFMA + fragment-read stream); 11 us of prologue + epilogue'),
    ('D4', ['ctw16_kernel'], 'mfma', act(8, 128) + act(16, 64) + 64 * 128 * 64 * 2, 'as E2: 207 k cycles per workgroup against 131 k of MFMA issue (eight parity epilogues 35 k, prologue 11 k, barriers); the four-wave tile was slower (r04_c4_ablations.json)'),
    ('D5 + a10 + a11', ['final_bce_sweepw_kernel'], 'hbm', act(16, 64) + 2 * B * 32 ** 3 * 4, 'HBM (memory floor 31-35 us) + three transcendentals per voxel'),
    ('a11 batch means', ['final_reduce_metrics_kernel'], 'latency', 0, 'launch latency'),
]
lname = {'E2': 'E2', 'E3': 'E3', 'E4': 'E4', 'D2': 'D2', 'D3': 'D3', 'D4': 'D4'}
tkey = {'E1': 'E1', 'E2': 'E2', 'E3': 'E3', 'E4': 'E4', 'E4 sum + E5': 'E4sum_E5', 'D2': 'D2', 'D3': 'D3', 'D4': 'D4', 'D5 + a10 + a11': 'D5'}
print('| layer (SURVEY 8a) | kernel(s) | us / launch (rocprofv3 average) | achieved | of roofline | counter traffic MB (algorithmic MB, ratio) | MFMA busy % | parked % (wait_any) | what limits it |')
print('|---|---|---|---|---|---|---|---|---|')
total = 0.0
for layer, subs, bound, abytes, why in rows:
    us = sum(find(stats, s) for s in subs)
    total += us
    c = find(ctr, subs[0])
    if bound == 'mfma':
        fl = 2.0 * macs[lname[layer]] * B
        ach, frac = '%.0f TFLOP/s' % (fl / us / 1e6), '%.3f of 2.5 PFLOP/s' % (fl / us / 1e6 / 2500)
    elif bound == 'hbm':
        ach, frac = '%.2f TB/s' % (abytes / us / 1e6), '%.2f of 8 TB/s (%.2f of the 6.29 TB/s copy rate)' % (abytes / us / 1e6 / 8, abytes / us / 1e6 / 6.29)
    else:
        ach, frac = '-', 'latency'
    tr = traffic.get(tkey.get(layer, ''), None)
    if tr and abytes:
        t = '%.1f (%.1f, %.2fx)' % (tr['hbm_bytes_per_launch'] / MB, abytes / MB, tr['hbm_bytes_per_launch'] / abytes)
    elif tr:
        t = '%.1f' % (tr['hbm_bytes_per_launch'] / MB)
    else:
        t = '-'
    print('| %s | `%s` | %s | %s | %s | %s | %.1f | %.1f | %s |' % (layer, '` + `'.join(subs), ' + '.join('%.1f' % find(stats, s) for s in subs), ach, frac, t,
                                                                  float(c['mfma_busy_pct']), float(c['wait_any_pct']), why))
fl_rec, _ = workload.flops_per_reconstruction(cfg)
print('| **whole path** | 12 launches | **%.1f** (sum of the averages, one batch at a time) | %.0f TFLOP/s | %.3f of 2.5 PFLOP/s | | | | three batches in flight on three streams overlap this to the `ms_per_step` of the bench line |'
      % (total, fl_rec * B / total / 1e6, fl_rec * B / total / 1e6 / 2500))

#!/usr/bin/env python
"""bench.py -- 32^3 voxel reconstructions/sec at batch 256 on MI355X (BASELINE.json metric).

One "step" = one pass of the hot path over one batch: encoder -> clip|reparam|KL -> decoder -> sigmoid/BCE/TP/FP/FN,
i.e. getEval(missing_prob=0) (reference nolbo.py:1463-1501), inputs resident in HBM.  Workload = BASELINE.json
configs[1] (ModelNet40 VAE, 32^3, batch 256, bf16); synthetic voxels + random-init weights (no dataset/weights exist).

    python bench.py --gpus N --steps K --warmup W [--dtype bf16|f32|fp8]
For N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
Eval shards the batch dimension: rank r runs its own 256 reconstructions (weak scaling), no data-path collective.

Prints ONE JSON line (rank 0) with the driver's contract fields plus `roofline` (dominant kernel, HIP events in
the timed region) and `cpu_baseline` (the fp32 C oracle, rank 0 at N=1 only, bounded sample).
"""
import argparse
import contextlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'anytime-3d-reconstruction_amd'))

import numpy as np  # noqa: E402
import torch  # noqa: E402

PEAK = {'bf16': 2.5e15, 'f32': 157.3e12, 'fp8': 5.0e15}   # dense MFMA peaks, /opt/skills/guides/MI355X_MICROARCH.md


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=400)
    ap.add_argument('--warmup', type=int, default=50)
    ap.add_argument('--dtype', default='bf16', choices=['bf16', 'f32', 'fp8'])
    ap.add_argument('--batch', type=int, default=256)
    ap.add_argument('--voxel', type=int, default=32)
    ap.add_argument('--latent', type=int, default=64)
    ap.add_argument('--cpu-samples', type=int, default=256, help='samples per CPU-baseline pass (0 = skip)')
    ap.add_argument('--no-breakdown', action='store_true')
    ap.add_argument('--mode', default='eval', choices=['eval', 'train'],
                    help="'eval' = the BASELINE.json headline (default); 'train' = fit() steps (f32, gradients all-reduced over RCCL for N>1)")
    return ap.parse_args()


def cpu_baseline(cfg, ep, dp, x, eps, n, min_seconds=10.0, max_passes=50):
    """The oracle as the CPU 'port' baseline (the reference's TF-CPU path cannot exist here: no TensorFlow).
    Bounded sample: passes over the first n samples until >= min_seconds of CPU work."""
    from oracle import c_oracle as co
    co.build()
    n = min(n, x.shape[0])
    co.vae_eval_forward(cfg, ep, dp, x[:1], x[:1], eps[:1])            # warm-up (page in weights, spin up OpenMP)
    passes, dt, r = 0, 0.0, None
    while passes < max_passes and dt < min_seconds:
        t0 = time.perf_counter()
        r = co.vae_eval_forward(cfg, ep, dp, x[:n], x[:n], eps[:n])
        dt += time.perf_counter() - t0
        passes += 1
    return r, {'value': n * passes / dt, 'unit': 'reconstructions/s', 'cores': co.num_threads(), 'kind': 'port',
               'sample': '%d pass(es) over %d of the %d synthetic 32^3 samples, fp32 C oracle (oracle/voxvae_oracle.c, OpenMP), %.1f s'
                         % (passes, n, x.shape[0], dt)}


def cpu_baseline_torch(cfg, ep, dp, x, eps, ref, n=64, min_seconds=5.0, max_passes=20):
    """Secondary CPU bracket: the same graph on torch-CPU float32 ops (oneDNN convolutions, the library family
    TF-CPU dispatches to), bounded like the C leg; also cross-checks the two CPU statements against each other."""
    from oracle import torch_oracle as to
    n = min(n, x.shape[0], ref['logits'].shape[0])
    lg, bce, tp, fp, fn = to.eval_forward_f32(cfg, ep, dp, x[:n], eps[:n])          # warm-up + agreement with the C oracle
    agree = float(np.abs(lg - ref['logits'][:n]).max())
    passes, dt = 0, 0.0
    while passes < max_passes and dt < min_seconds:
        t0 = time.perf_counter()
        to.eval_forward_f32(cfg, ep, dp, x[:n], eps[:n])
        dt += time.perf_counter() - t0
        passes += 1
    return {'value': n * passes / dt, 'unit': 'reconstructions/s', 'cores': torch.get_num_threads(), 'kind': 'port',
            'max_logit_diff_vs_c_oracle': agree,
            'sample': '%d pass(es) over %d samples, torch-CPU float32 (F.conv3d / F.conv_transpose3d), %.1f s' % (passes, n, dt)}


def bench_train(a, model, x, eps, world, rank, dev, dist):
    """Training-step throughput (BASELINE.json configs[3]: batch sharded over the ranks, gradients summed by RCCL)."""
    from voxvae import train as T
    tr = T.Trainer(model._enc_eng, model._dec_eng, True, 1e-4, world_size=world)
    for _ in range(a.warmup):
        tr.step(x, x, eps)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        kl, stats, metrics = tr.step(x, x, eps)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    el = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([el], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        el = float(tt.item())
    if rank == 0:
        print(json.dumps({'metric': '32^3 voxel VAE training samples/sec (fit: fwd + bwd + Adam)', 'value': world * a.batch * a.steps / el,
                          'unit': 'samples/s', 'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup, 'ms_per_step': 1e3 * el / a.steps,
                          'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': a.dtype, 'data': 'synthetic',
                          'config': {'workload': 'ModelNet40 VAE fit(), %d^3 voxels, latent %d, batch %d per GPU (BASELINE.json configs[3])'
                                                 % (a.voxel, a.latent, a.batch), 'global_batch': a.batch * world,
                                     'parallelism': 'dp%d, bucketed RCCL all-reduce of gradients, per-rank BatchNorm' % world},
                          'final_loss_shape': float(metrics[0]), 'final_loss_kl': float(kl.mean())}))
    if dist is not None:
        dist.destroy_process_group()


def main():
    a = parse()
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world != a.gpus and world > 1:
        raise SystemExit('--gpus %d but WORLD_SIZE=%d' % (a.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X (no CPU fallback)')
    torch.cuda.set_device(local)
    dev = 'cuda:%d' % local
    dist = None
    if world > 1 or 'RANK' in os.environ:       # under torch.distributed.run (also with one rank: same code path as N > 1)
        import torch.distributed as dist
        dist.init_process_group('nccl', device_id=torch.device(dev))   # RCCL on ROCm

    import voxvae
    from voxvae import engine as E
    from voxvae import synthetic as syn
    from voxvae import workload
    voxvae.set_default_dtype(a.dtype)
    voxvae.set_default_device(dev)
    import src.module.nolbo as nolbo

    cfg = syn.make_config(a.voxel, a.latent, True)
    ep, dp = syn.make_encoder_params(cfg['encoder']), syn.make_decoder_params(cfg['decoder'])
    with contextlib.redirect_stdout(sys.stderr):      # the model classes print build messages like the reference's; stdout carries ONE JSON line
        model = nolbo.nolboSingleObject_modelnet_category_VAE(nolbo_structure=cfg)
    model._encoder.set_weights_dict(ep)
    model._decoder.set_weights_dict(dp)
    xh = syn.make_voxels(a.batch, a.voxel, seed=1234 + rank)
    epsh = syn.make_eps(a.batch, a.latent, seed=7 + rank)
    x = torch.from_numpy(xh).to(dev)
    eps = torch.from_numpy(epsh).to(dev)

    def step():
        return model.eval_forward_device(x, x, eps)

    if a.mode == 'train':
        return bench_train(a, model, x, eps, world, rank, dev, dist)

    # ---- per-layer breakdown (outside the timed region) -> dominant kernel
    lm = {n: v for n, v, _ in workload.layer_macs(cfg)}
    dominant, breakdown = 'D4', None
    if not a.no_breakdown:
        for _ in range(3):                      # weight packing, allocator growth and clock ramp happen here
            step()
        torch.cuda.synchronize()
        t = E.LayerTimer()
        model._enc_eng.timer = model._dec_eng.timer = t
        for _ in range(10):
            step()
        torch.cuda.synchronize()
        breakdown = {k: round(float(np.median([e0.elapsed_time(e1) for e0, e1 in v])), 4) for k, v in t.events.items()}
        dominant = max(breakdown, key=breakdown.get)
    tm = E.LayerTimer(only=dominant)
    model._enc_eng.timer = model._dec_eng.timer = tm

    # ---- timed region
    for _ in range(a.warmup):
        step()
    tm.events.clear()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    el = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([el], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        el = float(tt.item())
    nl, kms = tm.summary_ms()[dominant]

    # ---- parity gate + CPU baseline (rank 0, N == 1 only): AFTER the timed region -- the oracle's OpenMP team
    # spin-waits on every host core and would starve the launch thread
    cpu, cpu_torch, iou_delta, logit_err = None, None, None, None
    model._enc_eng.timer = model._dec_eng.timer = None
    pred, stats, metrics, kl = step()
    torch.cuda.synchronize()
    if rank == 0 and world == 1 and a.cpu_samples > 0:
        ref, cpu = cpu_baseline(cfg, ep, dp, xh, epsh, a.cpu_samples)
        cpu_torch = cpu_baseline_torch(cfg, ep, dp, xh, epsh, ref)
        n = ref['bce'].shape[0]
        s = stats[:n].cpu().numpy().astype(np.float64)
        iou_g = s[:, 1] / np.maximum(s[:, 1] + s[:, 2] + s[:, 3], 1)
        iou_c = ref['tp'] / np.maximum(ref['tp'] + ref['fp'] + ref['fn'], 1)
        iou_delta = float(abs(iou_g.mean() - iou_c.mean()))
        _, z_act, _ = model._encode_latent(x[:n], eps[:n])
        _, lg, _ = model._dec_eng.forward(z_act, x[:n], want_logits=True)
        logit_err = float(np.abs(lg.cpu().numpy() - ref['logits']).max())


    traffic = None
    tf = os.path.join(ROOT, 'profiles', 'r01_pmc_traffic.json')   # rocprofv3 PMC passes (cannot be collected live)
    if os.path.exists(tf) and a.dtype == 'bf16' and a.batch == 256 and a.voxel == 32:
        traffic = json.load(open(tf))['layers'].get(dominant, {}).get('hbm_bytes_per_launch')

    def layer_dtype(name):
        if a.dtype != 'fp8':
            return a.dtype
        eng = model._enc_eng if name.startswith('E') else model._dec_eng
        return 'fp8' if eng.packed.get('q%d' % (int(name[1:]) - 1)) else 'bf16'

    if rank == 0:
        fl_rec, fl_dense = workload.flops_per_reconstruction(cfg)
        flops = 2.0 * lm[dominant] * a.batch                     # algorithmic (valid-tap) FLOPs of one launch
        es = 4 if a.dtype == 'f32' else 2
        nlast = len(cfg['decoder']['filter_num_list'])
        half = (a.voxel // 2) ** 3
        if dominant == 'D%d' % nlast:        # decoder tail + losses: reads the widest activation + target, writes probabilities
            abytes = a.batch * (half * cfg['decoder']['filter_num_list'][-2] * es + 2 * a.voxel ** 3 * 4)
            roof = {'bound': 'hbm', 'kernel': 'final_bce kernel (%s, layer %s)' % (a.dtype, dominant), 'achieved': abytes / (kms * 1e-3) / 1e9,
                    'peak': 8000.0, 'unit': 'GB/s', 'frac': abytes / (kms * 1e-3) / 8e12, 'algorithmic_bytes_per_launch': abytes}
        elif dominant == 'E1':               # first conv: reads the f32 occupancy grid, writes the widest encoder activation
            abytes = a.batch * (a.voxel ** 3 * 4 + half * cfg['encoder']['filter_num_list'][0] * es)
            roof = {'bound': 'hbm', 'kernel': 'igemm_kernel MODE_FIRST (%s, layer E1)' % a.dtype, 'achieved': abytes / (kms * 1e-3) / 1e9,
                    'peak': 8000.0, 'unit': 'GB/s', 'frac': abytes / (kms * 1e-3) / 8e12, 'algorithmic_bytes_per_launch': abytes}
        else:
            achieved = flops / (kms * 1e-3)
            kdt = layer_dtype(dominant)           # in 'fp8' mode only the Cin % 128 == 0 layers run fp8 operands; the rest are bf16 kernels
            roof = {'bound': 'mfma', 'kernel': 'conv kernel (%s, layer %s)' % (kdt, dominant), 'achieved': achieved / 1e12,
                    'peak': PEAK[kdt] / 1e12, 'unit': 'TFLOP/s', 'frac': achieved / PEAK[kdt], 'algorithmic_flops_per_launch': flops}
        roof.update({'traffic': traffic, 'launch_ms': kms, 'launches_timed': nl})
        # the heaviest MFMA layer as well, whatever is dominant
        mf = max((k for k in (breakdown or {}) if k not in ('E1', 'D%d' % nlast)), key=lambda k: (breakdown or {}).get(k, 0), default=None)
        mfma_layer = None
        if mf is not None and mf in lm:
            mfma_layer = {'layer': mf, 'ms': breakdown[mf], 'TFLOPs': 2.0 * lm[mf] * a.batch / (breakdown[mf] * 1e-3) / 1e12,
                          'frac_of_mfma_peak': 2.0 * lm[mf] * a.batch / (breakdown[mf] * 1e-3) / PEAK[layer_dtype(mf)]}
        out = {
            'metric': '32^3 voxel reconstructions/sec at batch=256; IoU delta vs reference',
            'value': world * a.batch * a.steps / el,
            'unit': 'reconstructions/s',
            'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup,
            'ms_per_step': 1e3 * el / a.steps,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': a.dtype, 'data': 'synthetic',
            'config': {'workload': 'ModelNet40 VAE getEval(missing_prob=0), %d^3 voxels, latent %d, batch %d per GPU, '
                                   'encoder+reparam/KL+decoder+BCE/TP/FP/FN (BASELINE.json configs[1])' % (a.voxel, a.latent, a.batch),
                       'batch_per_gpu': a.batch, 'global_batch': a.batch * world, 'parallelism': 'batch-sharded x%d, no collective' % world},
            'iou_delta': iou_delta, 'max_logit_err_vs_cpu_oracle': logit_err,
            'whole_path': {'algorithmic_flops_per_reconstruction': fl_rec, 'dense_flops_per_reconstruction': fl_dense,
                           'achieved_TFLOPs_per_gpu': fl_rec * a.batch * a.steps / el / 1e12,
                           'frac_of_mfma_peak': fl_rec * a.batch * a.steps / el / PEAK[a.dtype]},
            'roofline': roof,
            'heaviest_mfma_layer': mfma_layer,
            'layer_ms': breakdown,
            'cpu_baseline': cpu, 'cpu_baseline_torch': cpu_torch,
        }
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()

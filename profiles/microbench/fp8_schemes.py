"""fp8 quantisation schemes at the TRAINED operating point, evaluated on the real kernels before any kernel is written for them.

VERDICT r03 item 1: config 5 (64^3, fp8 MFMA) must hold mean IoU within 1e-3 of the reference at a trained operating point, on
>= 256 samples, per layer, and with real per-block E8M0 activation scales tried.  An e4m3fn x e4m3fn product is exact in float32
and the MFMA accumulates in float32, so an fp8 layer is reproduced EXACTLY (up to summation order) by the bf16 kernel of the same
layer fed with operands that were rounded to the e4m3fn grid first: weights at pack time, activations by `engine.LAYER_INPUT_HOOK`.
That lets every scheme below run through the product's own path (all other layers bf16, as in the real 'fp8' mode):

  real    the shipped fp8 kernels (policy 'wide', E2 only, D4 only, 'all')            -- validates the simulation
  sim     the same quantisation simulated (per-output-channel weight scale, unscaled activations)
  w / a   weights only / activations only                                             -- which operand costs the IoU
  blk     activations with a power-of-two scale per 32-channel block (the MX / E8M0 form of mfma_scale_f32_32x32x64_f8f6f4)
  shape   weights rounded with error diffusion over the taps of one (cin, cout) pair (the taps one output sums: all 64 for the conv,
          the 8 of a parity class for the transposed conv), so that the rounding errors of a pair sum to ~0 instead of ~sqrt(n) ulps
          -- the component of the weight error that a locally constant activation does not average out

Reference = the float32 mode of the same path (within 1e-5 of the C oracle in the logits; IoU identical).  Run on the GPU box:
    python profiles/microbench/fp8_schemes.py [32|64] ...
Writes gpurun_out/fp8_schemes_<D>.json.
"""
import json
import os
import sys
import time

import numpy as np
import torch

_R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, _R)
sys.path.insert(0, os.path.join(_R, 'anytime-3d-reconstruction_amd'))
import voxvae  # noqa: E402
from voxvae import engine as E  # noqa: E402
from voxvae import synthetic as syn  # noqa: E402
from voxvae import trained as tr  # noqa: E402

DEV = 'cuda:0'
BN_EPS = 1e-3


def round_e4m3(x):
    """float tensor -> nearest e4m3fn value (round half to even, saturating at 448), in float32 arithmetic."""
    x = x.float()
    ax = x.abs().clamp(max=448.0)
    _, e = torch.frexp(ax)                    # ax = m 2^e, m in [0.5, 1)
    e = (e - 1).clamp(min=-6)                 # binade exponent; subnormals share 2^-6
    step = torch.exp2((e - 3).float())
    q = (torch.round(ax / step) * step).clamp(max=448.0)
    return torch.where(x < 0, -q, q)


def perchan_scale(w, cout_axis):
    red = [d for d in range(w.dim()) if d != cout_axis]
    s = w.abs().amax(dim=red).clamp_min(1e-20) / 256.0
    shape = [1] * w.dim()
    shape[cout_axis] = -1
    return s, s.view(shape)


def quant_w(w, cout_axis, groups=None):
    """-> (e4m3 values of w / s, s).  groups: lists of flat tap indices; error diffusion runs inside each group."""
    w = torch.as_tensor(w, dtype=torch.float32)
    s, sv = perchan_scale(w, cout_axis)
    ws = w / sv
    if groups is None:
        return round_e4m3(ws), s
    flat = ws.reshape(64, w.shape[3], w.shape[4]).clone()
    q = torch.empty_like(flat)
    for g in groups:
        carry = torch.zeros_like(flat[0])
        for t in g:
            v = flat[t] + carry
            q[t] = round_e4m3(v)
            carry = v - q[t]
    return q.reshape(w.shape), s


CONV_GROUPS = [list(range(64))]
CONVT_GROUPS = [[(kd * 4 + kh) * 4 + kw for kd in range(4) for kh in range(4) for kw in range(4)
                 if (kd % 2, kh % 2, kw % 2) == (pd, ph, pw)] for pd in range(2) for ph in range(2) for pw in range(2)]


def with_quant_weights(p, kernel, bn, cout_axis, groups=None):
    """Params with `kernel` replaced by its e4m3 image (unscaled) and the BatchNorm moving statistics re-expressed for it:
    BN(c / s) with mean / s and (var + eps) / s^2 - eps is BN(c)."""
    p = dict(p)
    q, s = quant_w(p[kernel], cout_axis, groups)
    p[kernel] = q.numpy()
    s = s.numpy()
    p[bn + '/moving_mean'] = (p[bn + '/moving_mean'] / s).astype(np.float32)
    p[bn + '/moving_variance'] = ((p[bn + '/moving_variance'] + BN_EPS) / s ** 2 - BN_EPS).astype(np.float32)
    return p


def act_hook(layers, block=False):
    def hook(name, h):
        if name not in layers or h.dtype != torch.bfloat16:
            return h
        x = h.float()
        if block:
            C = x.shape[-1]
            xb = x.view(*x.shape[:-1], C // 32, 32)
            amax = xb.abs().amax(dim=-1, keepdim=True).clamp_min(2.0 ** -40)
            sc = torch.exp2(torch.ceil(torch.log2(amax / 448.0)))
            return (round_e4m3(xb / sc) * sc).view_as(x).to(torch.bfloat16).contiguous()
        return round_e4m3(x).to(torch.bfloat16).contiguous()
    return hook


def main(D):
    import src.module.nolbo as nolbo
    kw = dict(voxel=D, latent=64, device=DEV)
    if D == 64:
        kw.update(batch=32, pool=256, dtype='bf16', max_steps=3000)
    t0 = time.time()
    cfg, ep, dp, info = tr.train_operating_point(**kw)
    print('fit: %d steps, %.1f s, reached %s, IoU(eval, gpu) %.3f' % (info['steps'], time.time() - t0, info['reached'], info['iou_eval_mode_gpu']), flush=True)
    x = np.concatenate([syn.make_voxels(256, D, seed=4321)[:192], syn.make_voxels(64, D, seed=777)], axis=0)
    eps = syn.make_eps(256, 64, seed=70)
    xd, ed = torch.from_numpy(x).to(DEV), torch.from_numpy(eps).to(DEV)
    nE, nD = 'E2', 'D4'                      # the two layers of policy 'wide' at both geometries
    kE, bE, kD, bD = 'conv1/kernel', 'bn1', 'convT3/kernel', 'bnT3'
    rows = []

    def run(label, dtype, encp=ep, decp=dp, hook=None, env=None, policy='wide'):
        for k in ('VV_FP8_OFF', 'VV_FP8_E2', 'VV_FP8_LAST'):
            os.environ.pop(k, None)
        os.environ.update(env or {})
        voxvae.set_default_dtype(dtype)
        voxvae.set_default_device(DEV)
        voxvae.set_fp8_policy(policy)
        m = nolbo.nolboSingleObject_modelnet_category_VAE(nolbo_structure=cfg)
        m._encoder.set_weights_dict(encp)
        m._decoder.set_weights_dict(decp)
        E.LAYER_INPUT_HOOK = hook
        ious, lgs = [], []
        try:
            for lo in range(0, 256, 64):
                _, z_act, _ = m._encode_latent(xd[lo:lo + 64], ed[lo:lo + 64])
                _, lg, st = m._dec_eng.forward(z_act, xd[lo:lo + 64], want_logits=True)
                s = st.double().cpu().numpy()
                ious.append(s[:, 1] / np.maximum(s[:, 1] + s[:, 2] + s[:, 3], 1))
                lgs.append(lg.reshape(64, -1))
        finally:
            E.LAYER_INPUT_HOOK = None
        iou, lg = np.concatenate(ious), torch.cat(lgs)
        del m
        return label, iou, lg

    _, iou_r, lg_r = run('f32', 'f32')
    print('reference (f32 mode): IoU %.4f (seen %.4f / unseen %.4f), logits in [%.1f, %.1f]' % (iou_r.mean(), iou_r[:192].mean(), iou_r[192:].mean(),
                                                                                              float(lg_r.min()), float(lg_r.max())), flush=True)

    def report(res):
        label, iou, lg = res
        d = lg - lg_r
        fl = int(((lg >= 0) != (lg_r >= 0)).sum())
        # the mean over 256 samples of a per-sample difference: its standard error says how far the number can be trusted
        diff = iou - iou_r
        row = {'scheme': label, 'iou_delta_signed': float(diff.mean()), 'stderr': float(diff.std(ddof=1) / np.sqrt(len(diff))),
               'max_per_sample': float(np.abs(diff).max()), 'flips': fl, 'rms_dlogit': float(d.pow(2).mean().sqrt()), 'max_dlogit': float(d.abs().max())}
        rows.append(row)
        print('%-34s IoU delta %+.2e +- %.1e  max/sample %.2e  flips %7d  rms dlogit %.4f' % (label, row['iou_delta_signed'], row['stderr'],
                                                                                          row['max_per_sample'], fl, row['rms_dlogit']), flush=True)

    report(run('bf16', 'bf16'))
    report(run('real fp8 wide (E2+D4)', 'fp8'))
    report(run('real fp8 E2 only', 'fp8', env={'VV_FP8_OFF': 'D4'}))
    report(run('real fp8 D4 only', 'fp8', env={'VV_FP8_OFF': 'E2'}))
    report(run('real fp8 all', 'fp8', policy='all'))
    epq, dpq = with_quant_weights(ep, kE, bE, 4), with_quant_weights(dp, kD, bD, 3)
    eps_, dps_ = with_quant_weights(ep, kE, bE, 4, CONV_GROUPS), with_quant_weights(dp, kD, bD, 3, CONVT_GROUPS)
    both = act_hook({nE, nD})
    report(run('sim wide', 'bf16', epq, dpq, both))
    report(run('sim E2 only', 'bf16', epq, dp, act_hook({nE})))
    report(run('sim D4 only', 'bf16', ep, dpq, act_hook({nD})))
    report(run('sim wide, weights only', 'bf16', epq, dpq))
    report(run('sim wide, activations only', 'bf16', ep, dp, both))
    report(run('sim wide, block-scaled act', 'bf16', epq, dpq, act_hook({nE, nD}, block=True)))
    report(run('sim wide, shaped weights', 'bf16', eps_, dps_, both))
    report(run('sim wide, shaped weights only', 'bf16', eps_, dps_))
    report(run('sim E2 only, shaped', 'bf16', eps_, dp, act_hook({nE})))
    report(run('sim D4 only, shaped', 'bf16', ep, dps_, act_hook({nD})))
    report(run('sim wide, shaped + block act', 'bf16', eps_, dps_, act_hook({nE, nD}, block=True)))
    out = {'voxel': D, 'fit': {k: v for k, v in info.items() if k != 'history'}, 'iou_ref': float(iou_r.mean()), 'samples': 256, 'rows': rows}
    os.makedirs(os.path.join(_R, 'gpurun_out'), exist_ok=True)
    json.dump(out, open(os.path.join(_R, 'gpurun_out', 'fp8_schemes_%d.json' % D), 'w'), indent=1)


if __name__ == '__main__':
    for a in (sys.argv[1:] or ['32']):
        main(int(a))

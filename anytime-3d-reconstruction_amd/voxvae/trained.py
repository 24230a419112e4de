"""A TRAINED operating point for parity checks, made with the repo's own `fit` (reference nolbo.py:1411-1447).

Seeded Glorot weights put the path at a near-random predictor: logits within a few units of zero, IoU ~ 0.1.  An
IoU delta or an occupancy flip count measured there says little (flips cancel, and the +-15.94 region where the
reference's `clip(sigmoid(l), 1e-7, 1 - 1e-7)` saturates is never reached end to end).  No trained weights exist in
the reference, so this module makes some: the 32^3 VAE fitted for a few hundred float32 steps
on a fixed pool of the seeded synthetic shapes until it reconstructs them (IoU >= `min_iou`) and emits logits
beyond the clip (max |logit| >= `min_abs_logit`, at least `min_saturated` of the voxels past +-15.94), checked on the GPU in evaluation mode (moving statistics).

    cfg, enc_p, dec_p, info = train_operating_point()      # ~10 s on one MI355X

The returned dicts are plain float32 numpy arrays in Keras layouts: the same weights can be handed to any model
(`set_weights_dict`) in any dtype mode and to the CPU oracle in the tests.  Used by tests/test_gpu_trained.py and
bench.py's `parity.trained` block; deterministic for a given device generation (seeded data, seeded torch RNG).
"""
import contextlib
import sys

import numpy as np
import torch

from . import synthetic as syn


def train_operating_point(voxel=32, latent=64, batch=64, pool=256, lr=1e-3, seed=0, min_iou=0.6, min_abs_logit=17.0,
                          min_saturated=0.01, check_every=100, max_steps=3000, device='cuda:0', dtype='f32', verbose=False, variational=True):
    """-> (config, encoder params, decoder params, info).  Trains with `dtype` arithmetic ('f32' = the reference's)."""
    import voxvae
    import src.module.nolbo as nolbo
    prev_dt, prev_dev = voxvae.default_dtype(), voxvae.default_device()
    voxvae.set_default_dtype(dtype)
    voxvae.set_default_device(device)
    try:
        cfg = syn.make_config(voxel, latent, variational)
        with contextlib.redirect_stdout(sys.stderr):
            cls = nolbo.nolboSingleObject_modelnet_category_VAE if variational else nolbo.nolboSingleObject_modelnet_category_AE
            model = cls(nolbo_structure=cfg, learning_rate=lr)
        xs = torch.from_numpy(syn.make_voxels(pool, voxel, seed=4321 + seed)).to(device)
        gen = torch.Generator(device=device)
        gen.manual_seed(1000 + seed)
        nb = pool // batch
        hist, step, ok = [], 0, False
        probe = xs[:batch].contiguous()
        probe_eps = torch.randn(batch, latent, device=device, generator=gen)
        while step < max_steps and not ok:
            for _ in range(check_every):
                x = xs[(step % nb) * batch:(step % nb + 1) * batch]
                eps = torch.randn(batch, latent, device=device, generator=gen)
                model.fit((x, x), _eps=eps) if variational else model.fit((x, x))
                step += 1
            # evaluation mode (moving statistics), on device: IoU and the logit range
            _, z_act, _ = model._encode_latent(probe, probe_eps if variational else None)
            _, logits, stats = model._dec_eng.forward(z_act, probe, want_logits=True)
            s = stats.double()
            iou = float((s[:, 1] / torch.clamp(s[:, 1] + s[:, 2] + s[:, 3], min=1.0)).mean())
            amax = float(logits.abs().max())
            sat = float((logits.abs() > 15.94).float().mean())      # beyond the reference's clip(p, 1e-7, 1 - 1e-7)
            hist.append((step, iou, amax, sat))
            if verbose:
                print('trained operating point: step %d  IoU %.3f  max|logit| %.1f  saturated %.2f %%' % (step, iou, amax, 100 * sat), file=sys.stderr)
            ok = iou >= min_iou and amax >= min_abs_logit and sat >= min_saturated
        ep, dp = model._encoder.get_weights_dict(), model._decoder.get_weights_dict()
        info = {'steps': step, 'batch': batch, 'pool': pool, 'lr': lr, 'fit_dtype': dtype, 'history': hist, 'reached': ok,
                'iou_eval_mode_gpu': hist[-1][1], 'max_abs_logit_gpu': hist[-1][2],
                'saturated_fraction_gpu': hist[-1][3]}
        del model
        return cfg, ep, dp, info
    finally:
        voxvae.set_default_dtype(prev_dt)
        voxvae.set_default_device(prev_dev)

"""Generates tests/golden/*.npz from the fp64 numpy oracle (oracle/numpy_oracle.py) on seeded
synthetic inputs (voxvae/synthetic.py).  The reference holds no golden vectors and cannot run
here (no TensorFlow), so these fixtures pin the build against ITS OWN definition-level oracle:
"parity unpinned" with respect to TensorFlow itself.

Weights are not stored (26.5 M parameters): a fixture stores the generator arguments and the
expected outputs; tests rebuild the weights from the same seeds.

Run:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'anytime-3d-reconstruction_amd'))

from oracle import numpy_oracle as no  # noqa: E402
from voxvae import synthetic as syn  # noqa: E402

CASES = {
    # name: (voxel, latent, variational, batch, classes)
    'vae_d32_l64_b2': (32, 64, True, 2, 40),
    'ae_d32_l64_b2': (32, 64, False, 2, 40),
    'vae_d16_l64_b3': (16, 64, True, 3, 40),
    'vae_d64_l16_b1': (64, 16, True, 1, 12),
}


def make_case(name):
    D, L, var, B, C = CASES[name]
    cfg = syn.make_config(D, L, var)
    ep = syn.make_encoder_params(cfg['encoder'], seed=42)
    dp = syn.make_decoder_params(cfg['decoder'], seed=43)
    x = syn.make_voxels(B, D, seed=1234)
    oh = syn.make_onehot(B, C, seed=5)
    cats = syn.make_category_vectors(C, L, seed=11)
    eps = syn.make_eps(B, L, seed=7)
    eps2 = syn.make_eps(B, L, seed=8)
    mask = syn.make_mask(B, L, 0.5, seed=13)
    out = {}
    for tag, mp in (('p0', 0.0), ('p5', 0.5)):
        res, det = no.vae_get_eval(cfg, ep, dp, (x, x, oh), cats, eps, missing_prob=mp, mask=mask, eps2=eps2,
                                   dtype=np.float64, variational=var, details=True)
        out[tag + '_scalars'] = np.array([float(v) for v in res[1:5]] + [float(v) for v in res[6:10]], np.float64)
        for k in ('enc_out', 'z', 'bce', 'tp', 'fp', 'fn') + (('kl',) if var else ()):
            out['%s_%s' % (tag, k)] = np.asarray(det[k], np.float64)
        out[tag + '_logits'] = det['logits'].astype(np.float32)
        if mp > 0:
            for k in ('z_corr', 'bce_c', 'tp_c', 'fp_c', 'fn_c', 'argmin_masked'):
                out['%s_%s' % (tag, k)] = np.asarray(det[k], np.float64)
            out[tag + '_logits_c'] = det['logits_c'].astype(np.float32)
    out['latent'] = no.vae_get_latent(cfg, ep, x, eps, variational=var)
    out['meta'] = np.array([D, L, int(var), B, C, 42, 43, 1234, 5, 11, 7, 8, 13], np.int64)
    return out


if __name__ == '__main__':
    for name in CASES:
        np.savez_compressed(os.path.join(HERE, name + '.npz'), **make_case(name))
        print('wrote', name)

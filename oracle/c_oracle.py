"""TEST INFRASTRUCTURE ONLY -- ctypes front end of oracle/libvoxvae_oracle.so (the fp32 C
restatement) composed into the same whole-path functions oracle/numpy_oracle.py offers.

PARITY UNPINNED (see numpy_oracle.py).  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this; the product never does.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, 'libvoxvae_oracle.so')
_lib = None

_ACT = {'None': 0, None: 0, 'linear': 0, 'elu': 1, 'relu': 2, 'lrelu': 3}


def build(force=False):
    if force or not os.path.exists(_LIB_PATH):
        subprocess.check_call(['make', '-C', _HERE, 'libvoxvae_oracle.so'] + (['-B'] if force else []))
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_LIB_PATH)
        _lib.vvo_num_threads.restype = ctypes.c_int
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def num_threads():
    return int(lib().vvo_num_threads())


def conv3d_same(x, w, stride):
    x, w = _f32(x), _f32(w)
    B, D, Ci = x.shape[0], x.shape[1], x.shape[4]
    k, Co = w.shape[0], w.shape[4]
    O = -(-D // stride)
    y = np.empty((B, O, O, O, Co), np.float32)
    lib().vvo_conv3d_same(_p(x), _p(w), _p(y), B, D, Ci, Co, k, stride)
    return y


def conv3d_transpose_same(x, w, stride):
    x, w = _f32(x), _f32(w)
    B, D, Ci = x.shape[0], x.shape[1], x.shape[4]
    k, Co = w.shape[0], w.shape[3]
    N = D * stride
    y = np.empty((B, N, N, N, Co), np.float32)
    lib().vvo_conv3d_transpose_same(_p(x), _p(w), _p(y), B, D, Ci, Co, k, stride)
    return y


def bn_act_(x, p, prefix, act):
    C = x.shape[-1]
    rows = x.size // C
    g, b = _f32(p[prefix + '/gamma']), _f32(p[prefix + '/beta'])
    m, v = _f32(p[prefix + '/moving_mean']), _f32(p[prefix + '/moving_variance'])
    lib().vvo_bn_act(_p(x), ctypes.c_long(rows), C, _p(g), _p(b), _p(m), _p(v), ctypes.c_float(1e-3), _ACT[act])
    return x


def encoder3D_forward(structure, p, x):
    """autoencoder3D.py:72-102, inference mode, average or max pool."""
    h = _f32(x)
    fl, st = structure['filter_num_list'], structure['strides_list']
    for i in range(len(fl) - 1):
        h = conv3d_same(h, p['conv%d/kernel' % i], st[i])
        bn_act_(h, p, 'bn%d' % i, structure['activation'])
    i = len(fl) - 1
    h = conv3d_same(h, p['conv%d/kernel' % i], st[i])
    if structure['final_pool'] == 'max':                  # autoencoder3D.py:92-93 (tf.reduce_max over the spatial axes)
        return np.ascontiguousarray(h.max(axis=(1, 2, 3)), dtype=np.float32)
    assert structure['final_pool'] == 'average'
    B, S, C = h.shape[0], h.shape[1] ** 3, h.shape[4]
    y = np.empty((B, C), np.float32)
    lib().vvo_mean_pool(_p(h), _p(y), B, S, C)
    return y


def decoder3D_logits(structure, p, z):
    """autoencoder3D.py:104-132 up to the logits, inference mode."""
    side = int(structure['output_shape'][0] // int(np.prod(structure['strides_list'])))
    ch = max(int(structure['filter_num_list'][0] // 64), 8)
    z = _f32(z)
    W, b = _f32(p['dense/kernel']), _f32(p['dense/bias'])
    B = z.shape[0]
    h = np.empty((B, W.shape[1]), np.float32)
    lib().vvo_dense(_p(z), _p(W), _p(b), _p(h), B, W.shape[0], W.shape[1])
    bn_act_(h, p, 'bn_dense', structure['activation'])
    h = h.reshape(B, side, side, side, ch)
    fl, st = structure['filter_num_list'], structure['strides_list']
    for i in range(len(fl) - 1):
        h = conv3d_transpose_same(h, p['convT%d/kernel' % i], st[i])
        bn_act_(h, p, 'bnT%d' % i, structure['activation'])
    i = len(fl) - 1
    return conv3d_transpose_same(h, p['convT%d/kernel' % i], st[i])


def reparam_kl(enc_out, eps, L):
    enc_out, eps = _f32(enc_out), _f32(eps)
    B = enc_out.shape[0]
    z = np.empty((B, L), np.float32)
    kl = np.empty(B, np.float32)
    lib().vvo_reparam_kl(_p(enc_out), _p(eps), _p(z), _p(kl), B, L)
    return z, kl


def sigmoid_bce_counts(logits, target, gamma=0.6, epsilon=1e-7):
    B = logits.shape[0]
    lg = _f32(logits).reshape(B, -1)
    tg = _f32(target).reshape(B, -1)
    V = lg.shape[1]
    probs = np.empty_like(lg)
    bce, tp, fp, fn = (np.empty(B, np.float32) for _ in range(4))
    lib().vvo_sigmoid_bce_counts(_p(lg), _p(tg), _p(probs), _p(bce), _p(tp), _p(fp), _p(fn), B, ctypes.c_long(V),
                                 ctypes.c_float(gamma), ctypes.c_float(epsilon))
    return probs.reshape(logits.shape), bce, tp, fp, fn


def vae_eval_forward(config, enc_p, dec_p, x, y, eps, variational=True):
    """getEval(missing_prob=0) core -- nolbo.py:1463-1501: the unit bench.py counts as one reconstruction."""
    L = config['z_category_dim']
    enc_out = encoder3D_forward(config['encoder'], enc_p, x)
    if variational:
        z, kl = reparam_kl(enc_out, eps, L)
    else:
        z, kl = enc_out, None
    logits = decoder3D_logits(config['decoder'], dec_p, z)
    probs, bce, tp, fp, fn = sigmoid_bce_counts(logits, y)
    return {'enc_out': enc_out, 'z': z, 'kl': kl, 'logits': logits, 'probs': probs, 'bce': bce, 'tp': tp, 'fp': fp,
            'fn': fn}


def vae_get_eval(config, enc_p, dec_p, x, y, onehot, category_vectors, eps, missing_prob, mask, eps2):
    """getEval with missing latents -- nolbo.py:1449-1528 -- composed from the fp32 C pieces above (both decoder passes) and
    the [B, L] latent algebra in float32 numpy (the reference's arithmetic type): the form the trained-weights parity test
    uses at batch sizes where the float64 definition-level oracle (numpy_oracle.vae_get_eval) takes minutes.
    Returns a dict: first pass `logits, probs, bce, tp, fp, fn, z, acc`, corrected pass the same keys with suffix `_c`."""
    from . import numpy_oracle as no
    L = config['z_category_dim']
    cats = _f32(category_vectors)
    enc_out = encoder3D_forward(config['encoder'], enc_p, x)
    z, _ = reparam_kl(enc_out, eps, L)                                       # :1464-1470
    m = _f32(mask)
    if missing_prob > 0:
        z = z * m                                                            # :1477
        z = np.where(z == 0, cats.mean(axis=0, dtype=np.float32)[None, :] * np.ones_like(z), z).astype(np.float32)   # :1481-1482
    _, acc = no._nearest_category_acc(z, cats, onehot)
    logits = decoder3D_logits(config['decoder'], dec_p, z)
    probs, bce, tp, fp, fn = sigmoid_bce_counts(logits, y)
    out = {'z': z, 'acc': acc, 'logits': logits, 'probs': probs, 'bce': bce, 'tp': tp, 'fp': fp, 'fn': fn}
    if missing_prob == 0:
        return out
    idx, _ = no._nearest_category_acc(z, cats, onehot, mask=m)               # :1505-1506
    z_prior = (cats[idx] + _f32(eps2)).astype(np.float32)                     # sampling(mu, logVar = 0): sqrt(exp(0)) = 1 (:1507-1509)
    z_corr = np.where(m == 0, z_prior, z).astype(np.float32)                 # :1510
    _, acc_c = no._nearest_category_acc(z_corr, cats, onehot)
    logits_c = decoder3D_logits(config['decoder'], dec_p, z_corr)
    probs_c, bce_c, tp_c, fp_c, fn_c = sigmoid_bce_counts(logits_c, y)
    out.update({'z_c': z_corr, 'acc_c': acc_c, 'logits_c': logits_c, 'probs_c': probs_c, 'bce_c': bce_c, 'tp_c': tp_c,
                'fp_c': fp_c, 'fn_c': fn_c, 'argmin_masked': idx})
    return out

"""Drop-in for the loss/latent ops of the reference's src/module/function.py (lines 35-38, 40-71, 73-115), each a
hand-written HIP kernel behind the C ABI (include/voxvae.h).  Same names, argument order and meaning:
note that binary_loss takes (xPred, xTarget) while voxelPrecisionRecall takes (xTarget, xPred), and that
xPred is a PROBABILITY in both, exactly as in the reference.  Inputs may be numpy arrays, DeviceArrays or
torch tensors; results are DeviceArrays (np.array()-able)."""
import ctypes

import numpy as np
import torch

import voxvae
from voxvae import lib as _L
from voxvae.tensor import DeviceArray, as_device_f32


def _st():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _dev(x):
    return as_device_f32(x, voxvae.default_device())


def sampling(mu, logVar, epsilon=None):
    """reference function.py:35-38.  `epsilon` (extension) injects the N(0,1) draw; default: drawn on device."""
    mu, logVar = _dev(mu), _dev(logVar)
    eps = torch.randn_like(mu) if epsilon is None else _dev(epsilon)
    out = torch.empty_like(mu)
    _L.call('vv_sampling', _L.ptr(mu), _L.ptr(logVar), _L.ptr(eps), _L.ptr(out), mu.numel(), _st())
    return DeviceArray(out)


def binary_loss(xPred, xTarget, epsilon=1e-7, gamma=0.5, b_range=False):
    """reference function.py:73-82 -> per-sample loss [B]."""
    p, t = _dev(xPred), _dev(xTarget)
    B = p.shape[0]
    V = p.numel() // B
    if t.numel() != p.numel():
        raise ValueError('xPred %s and xTarget %s differ in size' % (tuple(p.shape), tuple(t.shape)))
    out = torch.empty(B, dtype=torch.float32, device=p.device)
    _L.call('vv_binary_loss', _L.ptr(p), _L.ptr(t), float(epsilon), float(gamma), float(b_range), _L.ptr(out), B, V, _st())
    return DeviceArray(out)


def kl_loss(mean, logVar, mean_target, logVar_target):
    """reference function.py:84-98 -> [B]."""
    m, lv, mt, lvt = _dev(mean), _dev(logVar), _dev(mean_target), _dev(logVar_target)
    B, Lz = m.shape[0], m.numel() // m.shape[0]
    out = torch.empty(B, dtype=torch.float32, device=m.device)
    _L.call('vv_kl_loss', _L.ptr(m), _L.ptr(lv), _L.ptr(mt), _L.ptr(lvt), _L.ptr(out), B, Lz, _st())
    return DeviceArray(out)


def regulizer_loss(z_mean, z_logVar, dist_in_z_space, class_input=None):
    """reference function.py:40-71 -> [B]: pairwise hinge that keeps latent means at least `dist_in_z_space` apart
    (scaled L1), optionally only between samples of the same class."""
    m, lv = _dev(z_mean), _dev(z_logVar)
    B, Lz = m.shape[0], m.numel() // m.shape[0]
    c = None if class_input is None else _dev(class_input)
    out = torch.empty(B, dtype=torch.float32, device=m.device)
    _L.call('vv_regulizer_loss', _L.ptr(m), _L.ptr(lv), _L.ptr(c), float(dist_in_z_space), _L.ptr(out), B, Lz,
            0 if c is None else c.numel() // B, _st())
    return DeviceArray(out)


def voxelPrecisionRecall(xTarget, xPred, prob=0.5):
    """reference function.py:100-115 -> (TP, FP, FN), each [B]."""
    t, p = _dev(xTarget), _dev(xPred)
    B = p.shape[0]
    V = p.numel() // B
    tp, fp, fn = (torch.empty(B, dtype=torch.float32, device=p.device) for _ in range(3))
    _L.call('vv_voxel_precision_recall', _L.ptr(t), _L.ptr(p), float(prob), _L.ptr(tp), _L.ptr(fp), _L.ptr(fn), B, V, _st())
    return DeviceArray(tp), DeviceArray(fp), DeviceArray(fn)

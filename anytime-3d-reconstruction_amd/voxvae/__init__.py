"""voxvae: MI355X-native runtime under the reference-compatible `src.*` modules.

    lib        ctypes binding of lib/libvoxvae.so (C ABI: include/voxvae.h); no fallback
    engine     encoder / decoder layer chains over that ABI
    synthetic  seeded configs, weights and voxel batches (no dataset ships with the reference)
    tensor     DeviceArray, the np.array()-able handle the model API returns
"""
import os

_DEFAULTS = {'dtype': os.environ.get('VOXVAE_DTYPE', 'f32'), 'device': os.environ.get('VOXVAE_DEVICE', 'cuda:0')}


def set_default_dtype(dtype):
    """'f32' (exact-f32 MFMA, the reference's arithmetic type; default), 'bf16' (bf16 MFMA, f32 accumulate) or 'fp8'
    (inference only: the MFMA layers with Cin % 128 == 0 on e4m3fn operands with per-channel weight scales, the rest bf16)."""
    if dtype not in ('f32', 'bf16', 'fp8'):
        raise ValueError(dtype)
    _DEFAULTS['dtype'] = dtype


def set_default_device(device):
    _DEFAULTS['device'] = device


def default_dtype():
    return _DEFAULTS['dtype']


def default_device():
    return _DEFAULTS['device']

"""voxvae: MI355X-native runtime under the reference-compatible `src.*` modules.

    lib        ctypes binding of lib/libvoxvae.so (C ABI: include/voxvae.h); no fallback
    engine     encoder / decoder layer chains over that ABI
    synthetic  seeded configs, weights and voxel batches (no dataset ships with the reference)
    tensor     DeviceArray, the np.array()-able handle the model API returns
"""
import os

_DEFAULTS = {'dtype': os.environ.get('VOXVAE_DTYPE', 'f32'), 'device': os.environ.get('VOXVAE_DEVICE', 'cuda:0'),
             'fp8_policy': os.environ.get('VV_FP8_POLICY', 'wide')}


def set_default_dtype(dtype):
    """'f32' (exact-f32 MFMA, the reference's arithmetic type; default), 'bf16' (bf16 MFMA, f32 accumulate) or 'fp8'
    (inference only: MFMA layers chosen by set_fp8_policy() on e4m3fn operands with per-channel weight scales, the rest bf16)."""
    if dtype not in ('f32', 'bf16', 'fp8'):
        raise ValueError(dtype)
    _DEFAULTS['dtype'] = dtype


def set_fp8_policy(policy):
    """Which MFMA layers run on e4m3fn operands in 'fp8' mode (engines built afterwards).

    'wide' (default): the layers that have a direct fp8 kernel -- the widest encoder and decoder layer (E2 / D4 at 32^3:
           62 % of the path's FLOPs and nearly all of the time fp8 saves).  Measured at the TRAINED operating points
           (tests/test_gpu_trained.py; 256 samples each): mean IoU within 5.0e-4 (32^3) / 4.3e-4 (64^3, BASELINE config 5's geometry)
           of the float32 oracle -- north_star's bar is 1e-3.  The weight images are rounded with error diffusion over the taps an
           output sums (engine.quant_fp8), which removes the weight rounding's share of that cost.
    'all':  every layer whose Cin is a multiple of 128 (and E2 through tap-pair rows), as in rounds 1-2.  Fastest, but each
           fp8 layer adds ~1-3 % of noise to its pre-activations (3 mantissa bits on both operands), and at a trained operating
           point the sum costs 1.44e-3 (32^3) / 1.23e-3 (64^3) of mean IoU -- beyond the bar; the layers between E3 and D3 also
           gain little time."""
    if policy not in ('wide', 'all'):
        raise ValueError(policy)
    _DEFAULTS['fp8_policy'] = policy


def fp8_policy():
    return _DEFAULTS['fp8_policy']


def set_default_device(device):
    _DEFAULTS['device'] = device


def default_dtype():
    return _DEFAULTS['dtype']


def default_device():
    return _DEFAULTS['device']

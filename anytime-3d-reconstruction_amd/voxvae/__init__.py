"""voxvae: MI355X-native runtime under the reference-compatible `src.*` modules.

    lib        ctypes binding of lib/libvoxvae.so (C ABI: include/voxvae.h); no fallback
    engine     encoder / decoder layer chains over that ABI
    synthetic  seeded configs, weights and voxel batches (no dataset ships with the reference)
    tensor     DeviceArray, the np.array()-able handle the model API returns
"""
import os

_DEFAULTS = {'dtype': os.environ.get('VOXVAE_DTYPE', 'f32'), 'device': os.environ.get('VOXVAE_DEVICE', 'cuda:0'),
             'fp8_policy': os.environ.get('VV_FP8_POLICY', 'mid')}


def set_default_dtype(dtype):
    """'f32' (exact-f32 MFMA, the reference's arithmetic type; default), 'bf16' (bf16 MFMA, f32 accumulate) or 'fp8'
    (inference only: MFMA layers chosen by set_fp8_policy() on e4m3fn operands with per-channel weight scales, the rest bf16)."""
    if dtype not in ('f32', 'bf16', 'fp8'):
        raise ValueError(dtype)
    _DEFAULTS['dtype'] = dtype


def set_fp8_policy(policy):
    """Which MFMA layers run on e4m3fn operands in 'fp8' mode (engines built afterwards).

    Measured at the TRAINED operating points (256 samples each, profiles/r04_fp8_policy_mid_{32,64}.jsonl; north_star's bar: mean IoU within
    1e-3 of the float32 oracle); ms = one 256-batch at 32^3 / one 64-sample shard at 64^3, one stream:

      policy   fp8 layers                          32^3: IoU delta, ms      64^3 (BASELINE config 5): IoU delta, ms
      'wide'   E2, D4 (the direct fp8 kernels)     5.0e-4   0.439           4.3e-4   0.934
      'mid'    E2, E3, D3, D4   (DEFAULT)          5.8e-4   0.419           6.5e-4   0.820
      'most'   everything but the encoder tail     1.10e-3  0.409  (over)   7.4e-4   0.747   (inside, 3 standard errors reach the bar)
      'all'    every eligible layer                1.44e-3  0.410  (over)   1.23e-3  0.745   (over)

    'mid' is the widest policy that is inside the bar at BOTH points with three standard errors to spare.  The weight images are rounded
    with error diffusion over the taps an output sums (engine.quant_fp8), which removes the weight rounding's share of the cost; what is
    left is the 3-bit mantissa of the activations, and the encoder tail's share (+4.9e-4 at 64^3) is systematic: it moves z."""
    if policy not in ('wide', 'mid', 'most', 'all'):
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

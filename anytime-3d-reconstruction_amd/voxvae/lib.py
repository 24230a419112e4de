"""ctypes binding of lib/libvoxvae.so (the C ABI declared in include/voxvae.h).

There is NO fallback: if the HIP library is missing or a call returns a non-zero status this
module raises.  Nothing here (or anywhere in the package) imports oracle/.
"""
import ctypes
import os

import torch  # noqa: F401  -- FIRST: libvoxvae must bind to the HIP runtime torch already loaded (one runtime per
#                              process; loading /opt/rocm's copy beside torch's leaves ours without a device)

PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.path.join(PKG, 'lib', 'libvoxvae.so')
# The release library reads no environment variable.  The kernel-form overrides of the A/B tests and microbenchmarks live in a second
# build of the same sources (lib/libvoxvae_hooks.so, -DVV_TEST_HOOKS); a process that sets VOXVAE_TEST_HOOKS=1 (tests/conftest.py, the
# scripts under profiles/microbench/) has its calls routed there WHILE one of the hook variables below is set, and to the release
# library otherwise -- so everything that does not ask for an override still runs the product.
HOOKS_LIB_PATH = os.path.join(PKG, 'lib', 'libvoxvae_hooks.so')
HOOK_VARS = ('VV_CD_SHAPE', 'VV_CDH_STAGGER', 'VV_CDH_ABL', 'VV_DIRECT_MT', 'VV_LT_KSLICE', 'VV_SPLIT_TARGET', 'VV_SPLIT_MINCHUNKS', 'VV_NO_KHALVES', 'VV_POSMAJOR_CONV_SIDE',
             'VV_POSMAJOR_CONVT_SIDE', 'VV_STAGES', 'VV_NO_FIRSTCONV', 'VV_WGRAD_F32', 'VV_PG_TARGET', 'VV_CTW_PS', 'VV_CTW_SHAPE',
             'VV_NO_WGRAD_PHASE', 'VV_BN_NB', 'VV_BN_SWEEP', 'VV_FINAL_BCE', 'VV_FIRSTCONV_GATHER', 'VV_FIRSTCONV_WGS', 'VV_FIRSTCONV_NOCHAIN', 'VV_CHUNK_SAMPLES')

VV_F32, VV_BF16, VV_FP8 = 0, 1, 2
ACT = {None: 0, 'None': 0, 'linear': 0, 'elu': 1, 'relu': 2, 'lrelu': 3}
DTYPES = {'f32': VV_F32, 'fp32': VV_F32, 'float32': VV_F32, 'bf16': VV_BF16, 'bfloat16': VV_BF16, 'fp8': VV_FP8}

_vp, _i, _f, _sz, _l = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_size_t, ctypes.c_long

# name -> (restype, argtypes); mirrors include/voxvae.h one to one (tests/test_abi.py checks the header)
SIGNATURES = {
    'vv_abi_version': (_i, []),
    'vv_status_string': (ctypes.c_char_p, [_i]),
    'vv_last_hip_error': (ctypes.c_char_p, []),
    'vv_pack_conv_k4': (_i, [_vp, _vp, _i, _i, _i, _vp]),
    'vv_pack_convT_k4s2': (_i, [_vp, _vp, _i, _i, _i, _vp]),
    'vv_pack_conv_k4s1_meanpool': (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    'vv_pack_conv_k4s1_full': (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    'vv_max_over_positions': (_i, [_vp, _vp, _i, _i, _i, _vp]),
    'vv_sigmoid_f32': (_i, [_vp, _vp, _l, _vp]),
    'vv_pack_convT_k4s1_dense': (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    'vv_pack_dense': (_i, [_vp, _vp, _i, _i, _i, _vp]),
    'vv_fold_bn': (_i, [_vp, _vp, _vp, _vp, _vp, _f, _vp, _vp, _i, _i, _vp]),
    'vv_conv3d_first_fwd': (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    'vv_conv3d_first_fwd_io': (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    'vv_conv3d_k4s2_workspace_bytes': (_sz, [_i, _i, _i, _i, _i]),
    'vv_conv3d_k4s2_fwd': (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _sz, _vp]),
    'vv_convT3d_k4s2_workspace_bytes': (_sz, [_i, _i, _i, _i, _i]),
    'vv_convT3d_k4s2_fwd': (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _sz, _vp]),
    'vv_conv3d_k4s2_fwd_io': (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _sz, _vp]),
    'vv_convT3d_k4s2_fwd_io': (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _sz, _vp]),
    'vv_conv3d_k4s2_direct_supported': (_i, [_i, _i, _i, _i]),
    'vv_conv3d_k4s2_direct_fwd': (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    'vv_conv3d_k4s2_direct_fwd_io': (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    'vv_conv3d_k4s2_direct_fp8_supported': (_i, [_i, _i, _i]),
    'vv_conv3d_k4s2_direct_fp8_fwd': (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    'vv_convT3d_k4s2_direct_supported': (_i, [_i, _i, _i, _i]),
    'vv_pack_convT_k4s2_frag': (_i, [_vp, _vp, _i, _i, _vp]),
    'vv_convT3d_k4s2_direct_fwd': (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    'vv_convT3d_k4s2_direct_fp8_supported': (_i, [_i, _i, _i]),
    'vv_pack_convT_k4s2_frag_fp8': (_i, [_vp, _vp, _i, _i, _vp]),
    'vv_convT3d_k4s2_direct_fp8_fwd': (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    'vv_conv3d_k4s2_skip_supported': (_i, [_i, _i, _i, _i]),
    'vv_convT3d_k4s2_skip_supported': (_i, [_i, _i, _i, _i]),
    'vv_pack_conv_k4_skip': (_i, [_vp, _vp, _i, _i, _vp]),
    'vv_pack_skip_images': (_i, [_vp, _vp, _vp, _vp, _vp, _i, _vp]),
    'vv_pack_convT_k4s2_skip': (_i, [_vp, _vp, _i, _i, _vp]),
    'vv_conv3d_k4s2_skip_fwd': (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    'vv_convT3d_k4s2_skip_fwd': (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    'vv_convT3d_k4s2_whole_supported': (_i, [_i, _i, _i, _i]),
    'vv_convT3d_k4s2_whole_fwd': (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    'vv_conv3d_k4s2_pos_supported': (_i, [_i, _i, _i, _i]),
    'vv_convT3d_k4s2_pos_supported': (_i, [_i, _i, _i, _i]),
    'vv_conv3d_k4s2_pos_workspace_bytes': (_sz, [_i, _i, _i]),
    'vv_convT3d_k4s2_pos_workspace_bytes': (_sz, [_i, _i, _i]),
    'vv_conv3d_k4s2_pos_fwd': (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _sz, _vp]),
    'vv_convT3d_k4s2_pos_fwd': (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _sz, _vp]),
    'vv_dense_workspace_bytes': (_sz, [_i, _i, _i, _i]),
    'vv_dense_fwd': (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _sz, _vp]),
    'vv_latent_tail_supported': (_i, [_i, _i, _i, _i, _i, _i, _i]),
    'vv_latent_tail_workspace_bytes': (_sz, [_i, _i, _i, _i]),
    'vv_latent_tail_fwd': (_i, [_vp] * 15 + [_i] * 9 + [_vp, _sz, _vp]),
    'vv_conv_pos_latent_tail_supported': (_i, [_i] * 8),
    'vv_conv_pos_latent_tail_workspace_bytes': (_sz, [_i, _i, _i, _i]),
    'vv_conv_pos_latent_tail_fwd': (_i, [_vp] * 4 + [_i, _i] + [_vp] * 14 + [_i] * 8 + [_vp, _sz, _vp]),
    'vv_reparam_kl_fwd': (_i, [_vp, _vp, _vp, _f, _vp, _vp, _i, _vp, _vp, _vp, _i, _i, _vp]),
    'vv_convT3d_final_bce_workspace_bytes': (_sz, [_i, _i]),
    'vv_convT3d_final_bce_fwd': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _f, _f, _i, _vp, _sz, _vp]),
    'vv_convT3d_final_bce_metrics_fwd': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _f, _f, _i, _vp, _sz, _vp]),
    'vv_shape_metrics': (_i, [_vp, _vp, _i, _vp]),
    'vv_latent_mask_fill': (_i, [_vp, _vp, _vp, _i, _vp, _vp, _i, _i, _i, _vp]),
    'vv_nearest_category': (_i, [_vp, _vp, _vp, _i, _vp, _i, _i, _vp]),
    'vv_latent_correct': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    'vv_category_accuracy': (_i, [_vp, _vp, _i, _vp, _i, _vp]),
    'vv_binary_loss': (_i, [_vp, _vp, _f, _f, _f, _vp, _i, ctypes.c_long, _vp]),
    'vv_voxel_precision_recall': (_i, [_vp, _vp, _f, _vp, _vp, _vp, _i, ctypes.c_long, _vp]),
    'vv_kl_loss': (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _vp]),
    'vv_unpack_bits_gather': (_i, [_vp, _vp, _vp, _i, _l, _vp]),
    'vv_pack_bits': (_i, [_vp, _vp, _f, _l, _vp]),
    'vv_adam_step_multi': (_i, [_vp, _i, _f, _f, _f, _f, _vp]),
    'vv_convert': (_i, [_vp, _vp, _l, _i, _i, _vp]),
    'vv_regulizer_loss': (_i, [_vp, _vp, _vp, _f, _vp, _i, _i, _i, _vp]),
    'vv_sampling': (_i, [_vp, _vp, _vp, _vp, ctypes.c_long, _vp]),
    'vv_bn_workspace_bytes': (_sz, [_l, _i]),
    'vv_bn_train_stats': (_i, [_vp, _l, _i, _vp, _vp, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _sz, _vp]),
    'vv_bn_finalize_stats': (_i, [_vp, _i, _l, _i, _vp, _vp, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    'vv_convT3d_k4s2_whole_stats_blocks': (_i, [_i]),
    'vv_convT3d_k4s2_whole_stats_fwd': (_i, [_vp, _vp, _vp, _vp, _sz, _i, _i, _i, _i, _i, _vp]),
    'vv_bn_act_fwd': (_i, [_vp, _vp, _vp, _vp, _l, _i, _i, _i, _vp]),
    'vv_bn_act_bwd': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _l, _i, _i, _i, _vp, _sz, _vp]),
    'vv_wgrad_workspace_bytes': (_sz, [_l, _i, _i]),
    'vv_wgrad_dense': (_i, [_vp, _vp, _vp, _l, _i, _i, _i, _i, _i, _vp, _sz, _vp]),
    'vv_wgrad_conv_k4s2': (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _sz, _vp]),
    'vv_unpack_meanpool_grad': (_i, [_vp, _vp, _i, _i, _i, _vp]),
    'vv_unpack_convT_dense_grad': (_i, [_vp, _vp, _i, _i, _i, _vp]),
    'vv_transpose_f32': (_i, [_vp, _vp, _i, _i, _vp]),
    'vv_colsum': (_i, [_vp, _vp, _l, _i, _i, _vp]),
    'vv_bce_bwd': (_i, [_vp, _vp, _vp, _i, _l, _f, _f, _f, _vp]),
    'vv_reparam_kl_bwd': (_i, [_vp, _vp, _vp, _vp, _f, _vp, _i, _i, _f, _vp]),
    'vv_adam_step': (_i, [_vp, _vp, _vp, _vp, _l, _f, _f, _f, _f, _vp]),
}


class VoxVaeError(RuntimeError):
    pass


_lib = None
_hooks_lib = None


def _open(path, hooks):
    if not os.path.exists(path):
        raise VoxVaeError('HIP library missing: %s (run `python __graft_entry__.py` / voxvae/build.py); '
                          'there is no CPU fallback' % path)
    # a library built from other sources than the ones in the tree is refused (content hash written by voxvae/build.py):
    # a stale .so would otherwise run silently after a checkout that resets mtimes
    from . import build as _build
    if os.path.isdir(_build.CSRC) and os.path.exists(path + '.srchash') and not _build.is_current(hooks):
        raise VoxVaeError('%s is stale: csrc/ or include/voxvae.h changed since it was built (run `python __graft_entry__.py`)' % path)
    lib = ctypes.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)      # AttributeError if the symbol is not exported
        fn.restype, fn.argtypes = res, args
    return lib


def hooks_requested():
    """True while this process opted into the test-hook build AND one of its variables is set."""
    env = os.environ
    return env.get('VOXVAE_TEST_HOOKS') == '1' and any(v in env for v in HOOK_VARS)


def load():
    """The library the next call goes to (loaded once each): the release build, or -- only in a process that set
    VOXVAE_TEST_HOOKS=1, and only while a hook variable is set -- the -DVV_TEST_HOOKS build.  Raises VoxVaeError when the
    library has not been built -- by design."""
    global _lib, _hooks_lib
    if hooks_requested():
        if _hooks_lib is None:
            _hooks_lib = _open(HOOKS_LIB_PATH, True)
        return _hooks_lib
    if _lib is None:
        _lib = _open(LIB_PATH, False)
    return _lib


def check(status, what):
    if status != 0:
        detail = ' [%s]' % load().vv_last_hip_error().decode() if status == -6 else ''
        raise VoxVaeError('%s failed: %s (%d)%s' % (what, load().vv_status_string(status).decode(), status, detail))


def call(name, *args):
    check(getattr(load(), name)(*args), name)


def ptr(t):
    """Device pointer of a torch tensor (or None)."""
    return None if t is None else ctypes.c_void_p(t.data_ptr())

"""GPU parity tests, per kernel, through the C ABI (voxvae.lib) against the oracle.
f32 mode: exact-f32 MFMA -> tight tolerances.  bf16 mode: inputs/weights rounded to bf16 first so the
comparison isolates the kernel (accumulation is f32 in both)."""
import ctypes

import numpy as np
import pytest
import torch

from oracle import numpy_oracle as no

pytestmark = pytest.mark.gpu

DEV = 'cuda:0'


@pytest.fixture(scope='module')
def L():
    from voxvae import lib
    lib.load()
    assert torch.cuda.is_available()
    return lib


def _st():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _dev(a, dt=torch.float32):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(DEV).to(dt).contiguous()


def _bf16_round(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(torch.bfloat16).to(torch.float32).numpy()


def _tol(dtname):
    return (2e-5, 1e-5) if dtname == 'f32' else (2e-2, 2e-2)   # (rtol on max|ref|, atol)


def _check(got, ref, dtname, what):
    got = got.float().cpu().numpy().astype(np.float64)
    rt, at = _tol(dtname)
    err = np.abs(got - ref).max()
    bound = rt * np.abs(ref).max() + at
    assert err <= bound, '%s %s: max err %.3e > %.3e (max|ref| %.3e)' % (what, dtname, err, bound, np.abs(ref).max())


@pytest.mark.parametrize('dtname', ['f32', 'bf16'])
# B >= 32 switches small grids to position-major rows with padded-tap skipping
@pytest.mark.parametrize('B,side,cin,cout', [(2, 8, 64, 128), (3, 4, 128, 64), (1, 16, 64, 128), (5, 2, 256, 512),
                                             (32, 4, 128, 64), (37, 8, 64, 128), (64, 2, 64, 64)])
def test_conv3d_k4s2(L, dtname, B, side, cin, cout):
    rng = np.random.default_rng(B * 1000 + side)
    dt, tdt = L.DTYPES[dtname], (torch.float32 if dtname == 'f32' else torch.bfloat16)
    x = rng.standard_normal((B, side, side, side, cin)).astype(np.float32)
    w = (rng.standard_normal((4, 4, 4, cin, cout)) / np.sqrt(64 * cin)).astype(np.float32)
    scale = rng.uniform(0.5, 1.5, cout).astype(np.float32)
    shift = rng.normal(0, 0.3, cout).astype(np.float32)
    if dtname == 'bf16':
        x, w = _bf16_round(x), _bf16_round(w)
    ref = no.activation(no.conv3d_same(x.astype(np.float64), w.astype(np.float64), 2) * scale + shift, 'elu')
    xd, wd, scd, shd = _dev(x, tdt), _dev(w), _dev(scale), _dev(shift)
    wp = torch.empty(cout, 64 * cin, dtype=tdt, device=DEV)
    L.call('vv_pack_conv_k4', L.ptr(wd), L.ptr(wp), cin, cout, dt, _st())
    nb = L.load().vv_conv3d_k4s2_workspace_bytes(B, side, cin, cout, dt)
    ws = torch.empty(max(nb, 16), dtype=torch.uint8, device=DEV)
    y = torch.empty(B, side // 2, side // 2, side // 2, cout, dtype=tdt, device=DEV)
    L.call('vv_conv3d_k4s2_fwd', L.ptr(xd), L.ptr(wp), L.ptr(scd), L.ptr(shd), L.ptr(y), B, side, cin, cout,
           1, dt, L.ptr(ws), ws.numel(), _st())
    torch.cuda.synchronize()
    _check(y, ref, dtname, 'conv3d_k4s2')


@pytest.mark.parametrize('dtname', ['f32', 'bf16'])
@pytest.mark.parametrize('B,side,cin,cout', [(2, 4, 128, 64), (3, 2, 512, 256), (1, 8, 128, 64), (2, 1, 64, 128),
                                             (33, 2, 64, 128), (40, 4, 64, 64), (64, 1, 64, 64)])
def test_convT3d_k4s2(L, dtname, B, side, cin, cout):
    rng = np.random.default_rng(B * 77 + side)
    dt, tdt = L.DTYPES[dtname], (torch.float32 if dtname == 'f32' else torch.bfloat16)
    x = rng.standard_normal((B, side, side, side, cin)).astype(np.float32)
    w = (rng.standard_normal((4, 4, 4, cout, cin)) / np.sqrt(8 * cin)).astype(np.float32)
    scale = rng.uniform(0.5, 1.5, cout).astype(np.float32)
    shift = rng.normal(0, 0.3, cout).astype(np.float32)
    if dtname == 'bf16':
        x, w = _bf16_round(x), _bf16_round(w)
    ref = no.activation(no.conv3d_transpose_same(x.astype(np.float64), w.astype(np.float64), 2) * scale + shift, 'elu')
    xd, wd, scd, shd = _dev(x, tdt), _dev(w), _dev(scale), _dev(shift)
    wp = torch.empty(8, cout, 8 * cin, dtype=tdt, device=DEV)
    L.call('vv_pack_convT_k4s2', L.ptr(wd), L.ptr(wp), cin, cout, dt, _st())
    nb = L.load().vv_convT3d_k4s2_workspace_bytes(B, side, cin, cout, dt)
    ws = torch.empty(max(nb, 16), dtype=torch.uint8, device=DEV)
    y = torch.empty(B, 2 * side, 2 * side, 2 * side, cout, dtype=tdt, device=DEV)
    L.call('vv_convT3d_k4s2_fwd', L.ptr(xd), L.ptr(wp), L.ptr(scd), L.ptr(shd), L.ptr(y), B, side, cin, cout,
           1, dt, L.ptr(ws), ws.numel(), _st())
    torch.cuda.synchronize()
    _check(y, ref, dtname, 'convT3d_k4s2')


# whole samples resident in LDS, padded taps skipped per MFMA row tile (skip_direct.hip); B = 5 / 9 leave a ragged last quad
@pytest.mark.parametrize('act', [1, 0])
@pytest.mark.parametrize('B,cin,cout', [(4, 128, 256), (5, 64, 64), (9, 256, 512), (33, 64, 128)])
def test_conv3d_k4s2_skip(L, B, cin, cout, act):
    rng = np.random.default_rng(B * 31 + cin)
    x = _bf16_round(rng.standard_normal((B, 8, 8, 8, cin)).astype(np.float32))
    w = _bf16_round((rng.standard_normal((4, 4, 4, cin, cout)) / np.sqrt(64 * cin)).astype(np.float32))
    scale = rng.uniform(0.5, 1.5, cout).astype(np.float32)
    shift = rng.normal(0, 0.3, cout).astype(np.float32)
    assert L.load().vv_conv3d_k4s2_skip_supported(8, cin, cout, L.VV_BF16)
    ref = no.activation(no.conv3d_same(x.astype(np.float64), w.astype(np.float64), 2) * scale + shift, 'elu' if act else None)
    xd, wd, scd, shd = _dev(x, torch.bfloat16), _dev(w), _dev(scale), _dev(shift)
    wp = torch.empty(64 * cin * cout, dtype=torch.bfloat16, device=DEV)
    L.call('vv_pack_conv_k4_skip', L.ptr(wd), L.ptr(wp), cin, cout, _st())
    y = torch.full((B, 4, 4, 4, cout), float('nan'), dtype=torch.bfloat16, device=DEV)
    L.call('vv_conv3d_k4s2_skip_fwd', L.ptr(xd), L.ptr(wp), L.ptr(scd), L.ptr(shd), L.ptr(y), B, 8, cin, cout, act, L.VV_BF16, _st())
    torch.cuda.synchronize()
    _check(y, ref, 'bf16', 'conv3d_k4s2_skip')


@pytest.mark.parametrize('act', [1, 0])
@pytest.mark.parametrize('B,cin,cout', [(4, 256, 128), (3, 64, 128), (6, 512, 256), (33, 128, 128)])
def test_convT3d_k4s2_skip(L, B, cin, cout, act):
    rng = np.random.default_rng(B * 17 + cin)
    x = _bf16_round(rng.standard_normal((B, 4, 4, 4, cin)).astype(np.float32))
    w = _bf16_round((rng.standard_normal((4, 4, 4, cout, cin)) / np.sqrt(8 * cin)).astype(np.float32))
    scale = rng.uniform(0.5, 1.5, cout).astype(np.float32)
    shift = rng.normal(0, 0.3, cout).astype(np.float32)
    assert L.load().vv_convT3d_k4s2_skip_supported(4, cin, cout, L.VV_BF16)
    ref = no.activation(no.conv3d_transpose_same(x.astype(np.float64), w.astype(np.float64), 2) * scale + shift, 'elu' if act else None)
    xd, wd, scd, shd = _dev(x, torch.bfloat16), _dev(w), _dev(scale), _dev(shift)
    wp = torch.empty(64 * cin * cout, dtype=torch.bfloat16, device=DEV)
    L.call('vv_pack_convT_k4s2_skip', L.ptr(wd), L.ptr(wp), cin, cout, _st())
    y = torch.full((B, 8, 8, 8, cout), float('nan'), dtype=torch.bfloat16, device=DEV)
    L.call('vv_convT3d_k4s2_skip_fwd', L.ptr(xd), L.ptr(wp), L.ptr(scd), L.ptr(shd), L.ptr(y), B, 4, cin, cout, act, L.VV_BF16, _st())
    torch.cuda.synchronize()
    _check(y, ref, 'bf16', 'convT3d_k4s2_skip')


# whole-sample transposed convolution 8^3 x 128 -> 16^3 x 64 (convt_whole.hip): every parity split, every activation,
# batches that are not a multiple of anything, null scale / shift; compared against the float64 definition
@pytest.mark.parametrize('shape', [16, 32, 4])         # 16 (default) / 32: the eight-wave kernel on MFMA 16x16x32 / 32x32x16; 4: four waves, one per SIMD, epilogue in the MFMA gaps (round 4: correct, slower)
@pytest.mark.parametrize('ps', [0, 1, 2, 4, 8])
@pytest.mark.parametrize('B,act', [(3, 1), (7, 0), (33, 2), (5, 3)])
def test_convT3d_k4s2_whole(L, B, act, ps, shape, monkeypatch):
    monkeypatch.setenv('VV_CTW_SHAPE', str(shape))
    cin, cout = 128, 64
    rng = np.random.default_rng(B * 23 + act)
    x = _bf16_round(rng.standard_normal((B, 8, 8, 8, cin)).astype(np.float32))
    w = _bf16_round((rng.standard_normal((4, 4, 4, cout, cin)) / np.sqrt(8 * cin)).astype(np.float32))
    scale = rng.uniform(0.5, 1.5, cout).astype(np.float32)
    shift = rng.normal(0, 0.3, cout).astype(np.float32)
    assert L.load().vv_convT3d_k4s2_whole_supported(8, cin, cout, L.VV_BF16)
    assert not L.load().vv_convT3d_k4s2_whole_supported(16, cin, cout, L.VV_BF16)
    ref = no.activation(no.conv3d_transpose_same(x.astype(np.float64), w.astype(np.float64), 2) * scale + shift,
                        {0: None, 1: 'elu', 2: 'relu', 3: 'lrelu'}[act])
    xd, wd, scd, shd = _dev(x, torch.bfloat16), _dev(w), _dev(scale), _dev(shift)
    wp = torch.empty(64 * cin * cout, dtype=torch.bfloat16, device=DEV)
    L.call('vv_pack_convT_k4s2_skip', L.ptr(wd), L.ptr(wp), cin, cout, _st())
    y = torch.full((B, 16, 16, 16, cout), float('nan'), dtype=torch.bfloat16, device=DEV)
    if ps:
        monkeypatch.setenv('VV_CTW_PS', str(ps))
    else:
        monkeypatch.delenv('VV_CTW_PS', raising=False)
    L.call('vv_convT3d_k4s2_whole_fwd', L.ptr(xd), L.ptr(wp), L.ptr(scd), L.ptr(shd), L.ptr(y), B, 8, cin, cout, act, L.VV_BF16, _st())
    torch.cuda.synchronize()
    _check(y, ref, 'bf16', 'convT3d_k4s2_whole')
    if ps in (0, 8):                                  # null scale / shift = identity
        y2 = torch.full_like(y, float('nan'))
        L.call('vv_convT3d_k4s2_whole_fwd', L.ptr(xd), L.ptr(wp), None, None, L.ptr(y2), B, 8, cin, cout, 0, L.VV_BF16, _st())
        torch.cuda.synchronize()
        _check(y2, no.conv3d_transpose_same(x.astype(np.float64), w.astype(np.float64), 2), 'bf16', 'convT3d_k4s2_whole (no BN)')


@pytest.mark.parametrize('B', [3, 64, 256])
def test_convT3d_whole_stats_form(L, B):
    """The training-mode forward of the widest decoder layer (round 4): the kernel that leaves the per-workgroup column sums of its own
    output.  Output bit-identical to the plain raw form; the column sums are those of the stored bf16 values; finalised, they are the
    batch statistics vv_bn_train_stats computes from the tensor (mean / variance / folded scale and shift / moving statistics)."""
    g = torch.Generator(device=DEV).manual_seed(B)
    cin, cout = 128, 64
    x = torch.randn(B, 8, 8, 8, cin, device=DEV, generator=g).to(torch.bfloat16)
    w = (torch.randn(4, 4, 4, cout, cin, device=DEV, generator=g) / (8 * cin) ** 0.5).contiguous()
    wk = torch.empty(64 * cin * cout, dtype=torch.bfloat16, device=DEV)
    L.call('vv_pack_convT_k4s2_skip', L.ptr(w), L.ptr(wk), cin, cout, _st())
    lib = L.load()
    y0 = torch.full((B, 16, 16, 16, cout), float('nan'), dtype=torch.bfloat16, device=DEV)
    L.call('vv_convT3d_k4s2_whole_fwd', L.ptr(x), L.ptr(wk), None, None, L.ptr(y0), B, 8, cin, cout, 0, L.VV_BF16, _st())
    nblk = lib.vv_convT3d_k4s2_whole_stats_blocks(B)
    part = torch.full((nblk, 2, cout), float('nan'), dtype=torch.float32, device=DEV)
    y1 = torch.full_like(y0, float('nan'))
    L.call('vv_convT3d_k4s2_whole_stats_fwd', L.ptr(x), L.ptr(wk), L.ptr(y1), L.ptr(part), part.numel() * 4, B, 8, cin, cout, L.VV_BF16, _st())
    torch.cuda.synchronize()
    assert torch.equal(y0, y1)
    yf = y1.double().reshape(-1, cout)
    s1, s2 = part[:, 0].double().sum(0), part[:, 1].double().sum(0)
    R = yf.shape[0]
    assert torch.allclose(s1, yf.sum(0), rtol=0, atol=2e-5 * float(yf.abs().sum(0).max()))
    assert torch.allclose(s2, (yf * yf).sum(0), rtol=2e-5, atol=0)
    # finalise: against vv_bn_train_stats on the tensor
    gamma, beta = torch.rand(cout, device=DEV, generator=g) + 0.5, torch.randn(cout, device=DEV, generator=g)
    outs = []
    for form in (0, 1):
        mean, var, rstd, scale, shift = (torch.empty(cout, device=DEV) for _ in range(5))
        mm, mv = torch.zeros(cout, device=DEV), torch.ones(cout, device=DEV)
        if form == 0:
            ws = torch.empty(max(lib.vv_bn_workspace_bytes(R, cout), 16), dtype=torch.uint8, device=DEV)
            L.call('vv_bn_train_stats', L.ptr(y1), R, cout, L.ptr(gamma), L.ptr(beta), 1e-3, 0.99, L.ptr(mean), L.ptr(var), L.ptr(rstd), L.ptr(scale),
                   L.ptr(shift), L.ptr(mm), L.ptr(mv), L.VV_BF16, L.ptr(ws), ws.numel(), _st())
        else:
            L.call('vv_bn_finalize_stats', L.ptr(part), nblk, R, cout, L.ptr(gamma), L.ptr(beta), 1e-3, 0.99, L.ptr(mean), L.ptr(var), L.ptr(rstd),
                   L.ptr(scale), L.ptr(shift), L.ptr(mm), L.ptr(mv), _st())
        torch.cuda.synchronize()
        outs.append([t.clone() for t in (mean, var, rstd, scale, shift, mm, mv)])
    for a, b in zip(*outs):
        assert torch.allclose(a, b, rtol=2e-5, atol=2e-6), (a - b).abs().max()
    assert lib.vv_convT3d_k4s2_whole_stats_fwd(L.ptr(x), L.ptr(wk), L.ptr(y1), L.ptr(part), 16, B, 8, cin, cout, L.VV_BF16, _st()) == -5


def test_pack_skip_images_batched_equals_single_calls(L):
    """vv_pack_skip_images (round 4: the training step's nine skip / position / whole-sample weight images in two launches) writes the
    bits of the single vv_pack_conv_k4_skip / vv_pack_convT_k4s2_skip calls -- mixed kinds, more than eight jobs of one kind, bad
    arguments refused."""
    import ctypes
    g = torch.Generator(device=DEV).manual_seed(9)
    shapes = [(0, 128, 256), (1, 512, 256), (0, 256, 512), (1, 256, 128), (1, 128, 64), (0, 64, 64), (1, 64, 128), (0, 128, 64), (1, 64, 64),
              (1, 128, 128), (1, 192, 64), (1, 64, 192), (1, 256, 64), (1, 64, 256)]          # 10 of kind 1: two launches of that kind
    ws = [torch.randn(4, 4, 4, (cin if k == 0 else cout), (cout if k == 0 else cin), device=DEV, generator=g).contiguous() for k, cin, cout in shapes]
    ref = []
    for (k, cin, cout), w in zip(shapes, ws):
        o = torch.full((64 * cin * cout,), float('nan'), dtype=torch.bfloat16, device=DEV)
        L.call('vv_pack_conv_k4_skip' if k == 0 else 'vv_pack_convT_k4s2_skip', L.ptr(w), L.ptr(o), cin, cout, _st())
        ref.append(o)
    outs = [torch.full_like(r, float('nan')) for r in ref]
    n = len(shapes)
    kinds = (ctypes.c_int * n)(*[k for k, _, _ in shapes])
    wp = (ctypes.c_void_p * n)(*[w.data_ptr() for w in ws])
    op = (ctypes.c_void_p * n)(*[o.data_ptr() for o in outs])
    ci = (ctypes.c_int * n)(*[c for _, c, _ in shapes])
    co = (ctypes.c_int * n)(*[c for _, _, c in shapes])
    L.call('vv_pack_skip_images', kinds, wp, op, ci, co, n, _st())
    torch.cuda.synchronize()
    for (k, cin, cout), a, b in zip(shapes, ref, outs):
        assert torch.equal(a.view(torch.int16), b.view(torch.int16)), (k, cin, cout)
    lib = L.load()
    bad = (ctypes.c_int * 1)(2)
    assert lib.vv_pack_skip_images(bad, wp, op, ci, co, 1, _st()) == -2
    assert lib.vv_pack_skip_images(kinds, wp, op, ci, co, 0, _st()) == -2
    assert lib.vv_pack_skip_images(None, wp, op, ci, co, 1, _st()) == -1


# position-major split-K GEMM of the 4^3 <-> 2^3 layers (posgemm.hip): ragged batches, channel tails, several sample tiles
@pytest.mark.parametrize('act', [1, 0])
@pytest.mark.parametrize('B,cin,cout', [(5, 64, 64), (37, 256, 512), (256, 128, 136), (300, 64, 128)])
def test_conv3d_k4s2_pos(L, B, cin, cout, act):
    rng = np.random.default_rng(B * 13 + cin)
    x = _bf16_round(rng.standard_normal((B, 4, 4, 4, cin)).astype(np.float32))
    w = _bf16_round((rng.standard_normal((4, 4, 4, cin, cout)) / np.sqrt(27 * cin)).astype(np.float32))
    scale = rng.uniform(0.5, 1.5, cout).astype(np.float32)
    shift = rng.normal(0, 0.3, cout).astype(np.float32)
    assert L.load().vv_conv3d_k4s2_pos_supported(4, cin, cout, L.VV_BF16)
    ref = no.activation(no.conv3d_same(x.astype(np.float64), w.astype(np.float64), 2) * scale + shift, 'elu' if act else None)
    xd, wd, scd, shd = _dev(x, torch.bfloat16), _dev(w), _dev(scale), _dev(shift)
    wp = torch.empty(64 * cin * cout, dtype=torch.bfloat16, device=DEV)
    L.call('vv_pack_conv_k4_skip', L.ptr(wd), L.ptr(wp), cin, cout, _st())
    ws = torch.empty(max(L.load().vv_conv3d_k4s2_pos_workspace_bytes(B, cin, cout), 16), dtype=torch.uint8, device=DEV)
    y = torch.full((B, 2, 2, 2, cout), float('nan'), dtype=torch.bfloat16, device=DEV)
    L.call('vv_conv3d_k4s2_pos_fwd', L.ptr(xd), L.ptr(wp), L.ptr(scd), L.ptr(shd), L.ptr(y), B, 4, cin, cout, act, L.VV_BF16,
           L.ptr(ws), ws.numel(), _st())
    torch.cuda.synchronize()
    _check(y, ref, 'bf16', 'conv3d_k4s2_pos')


@pytest.mark.parametrize('act', [1, 0])
@pytest.mark.parametrize('B,cin,cout', [(3, 64, 64), (40, 512, 256), (256, 128, 72), (290, 64, 128)])
def test_convT3d_k4s2_pos(L, B, cin, cout, act):
    rng = np.random.default_rng(B * 19 + cin)
    x = _bf16_round(rng.standard_normal((B, 2, 2, 2, cin)).astype(np.float32))
    w = _bf16_round((rng.standard_normal((4, 4, 4, cout, cin)) / np.sqrt(4 * cin)).astype(np.float32))
    scale = rng.uniform(0.5, 1.5, cout).astype(np.float32)
    shift = rng.normal(0, 0.3, cout).astype(np.float32)
    assert L.load().vv_convT3d_k4s2_pos_supported(2, cin, cout, L.VV_BF16)
    ref = no.activation(no.conv3d_transpose_same(x.astype(np.float64), w.astype(np.float64), 2) * scale + shift, 'elu' if act else None)
    xd, wd, scd, shd = _dev(x, torch.bfloat16), _dev(w), _dev(scale), _dev(shift)
    wp = torch.empty(64 * cin * cout, dtype=torch.bfloat16, device=DEV)
    L.call('vv_pack_convT_k4s2_skip', L.ptr(wd), L.ptr(wp), cin, cout, _st())
    ws = torch.empty(max(L.load().vv_convT3d_k4s2_pos_workspace_bytes(B, cin, cout), 16), dtype=torch.uint8, device=DEV)
    y = torch.full((B, 4, 4, 4, cout), float('nan'), dtype=torch.bfloat16, device=DEV)
    L.call('vv_convT3d_k4s2_pos_fwd', L.ptr(xd), L.ptr(wp), L.ptr(scd), L.ptr(shd), L.ptr(y), B, 2, cin, cout, act, L.VV_BF16,
           L.ptr(ws), ws.numel(), _st())
    torch.cuda.synchronize()
    _check(y, ref, 'bf16', 'convT3d_k4s2_pos')


@pytest.mark.parametrize('dtname', ['f32', 'bf16'])
@pytest.mark.parametrize('M,N,K', [(4, 64, 64), (7, 128, 16), (256, 128, 4096), (130, 4096, 64), (2, 64, 8200)])
def test_dense(L, dtname, M, N, K):
    rng = np.random.default_rng(M + N + K)
    dt, tdt = L.DTYPES[dtname], (torch.float32 if dtname == 'f32' else torch.bfloat16)
    x = rng.standard_normal((M, K)).astype(np.float32)
    w = (rng.standard_normal((K, N)) / np.sqrt(K)).astype(np.float32)
    if dtname == 'bf16':
        x, w = _bf16_round(x), _bf16_round(w)
    scale = rng.uniform(0.5, 1.5, N).astype(np.float32)
    shift = rng.normal(0, 0.3, N).astype(np.float32)
    ref = no.activation(x.astype(np.float64) @ w.astype(np.float64) * scale + shift, 'elu')
    wp = torch.empty(N, K, dtype=tdt, device=DEV)
    wd, xd, scd, shd = _dev(w), _dev(x, tdt), _dev(scale), _dev(shift)
    L.call('vv_pack_dense', L.ptr(wd), L.ptr(wp), K, N, dt, _st())
    nb = L.load().vv_dense_workspace_bytes(M, N, K, dt)
    ws = torch.empty(max(nb, 16), dtype=torch.uint8, device=DEV)
    y = torch.empty(M, N, dtype=torch.float32, device=DEV)
    L.call('vv_dense_fwd', L.ptr(xd), L.ptr(wp), L.ptr(scd), L.ptr(shd), L.ptr(y), M, N, K, 1, dt,
           L.VV_F32, L.ptr(ws), ws.numel(), _st())
    torch.cuda.synchronize()
    _check(y, ref, 'f32' if dtname == 'f32' else 'bf16', 'dense')


@pytest.mark.parametrize('dtname', ['f32', 'bf16'])
# D >= 32 (bf16) takes the plane-form kernel, smaller grids the gather form; B = 7 leaves a ragged last workgroup
@pytest.mark.parametrize('B,D', [(2, 32), (1, 16), (3, 8), (7, 32), (1, 64), (1, 128)])
def test_conv3d_first(L, dtname, B, D):
    from voxvae import synthetic as syn
    rng = np.random.default_rng(D)
    dt, tdt = L.DTYPES[dtname], (torch.float32 if dtname == 'f32' else torch.bfloat16)
    x = syn.make_voxels(B, D, seed=D)
    w = (rng.standard_normal((4, 4, 4, 1, 64)) / 8).astype(np.float32)
    scale = rng.uniform(0.5, 1.5, 64).astype(np.float32)
    shift = rng.normal(0, 0.3, 64).astype(np.float32)
    ref = no.activation(no.conv3d_same(x.astype(np.float64), w.astype(np.float64), 2) * scale + shift, 'elu')
    y = torch.empty(B, D // 2, D // 2, D // 2, 64, dtype=tdt, device=DEV)
    xd, wd, scd, shd = _dev(x), _dev(w), _dev(scale), _dev(shift)
    wp = torch.empty(64, 64, dtype=tdt, device=DEV)
    L.call('vv_pack_conv_k4', L.ptr(wd), L.ptr(wp), 1, 64, dt, _st())
    L.call('vv_conv3d_first_fwd', L.ptr(xd), L.ptr(wp), L.ptr(scd), L.ptr(shd), L.ptr(y), B, D, 64,
           1, dt, _st())
    torch.cuda.synchronize()
    rt = 'f32' if dtname == 'f32' else 'bf16'
    got = y.float().cpu().numpy()
    tol = 1e-5 if dtname == 'f32' else 1e-2 * np.abs(ref).max()
    assert np.abs(got - ref).max() <= tol, (rt, np.abs(got - ref).max())


@pytest.mark.parametrize('dtname', ['f32', 'bf16'])
@pytest.mark.parametrize('B,side,form', [(2, 16, 'box'), (3, 4, 'box'), (1, 8, 'box'), (2, 16, 'sweep'), (3, 8, 'sweep'), (1, 32, 'sweep'),
                                         (2, 16, 'sweepp'), (3, 8, 'sweepp')])
def test_convT3d_final_bce(L, dtname, B, side, form, monkeypatch):
    # bf16 has three kernels: 'box' (4^3 cells + halo per workgroup, small batches), 'sweep' (8x8 tile swept through the depth with the w
    # direction summed inside the MFMA, large batches) and 'sweepp' (the sweep form that publishes P[cell][64 taps], kept for reference); VV_FINAL_BCE overrides the batch heuristic so both are checked at test sizes
    if dtname == 'f32' and form.startswith('sweep'):
        pytest.skip('sweep form is bf16 only')
    monkeypatch.setenv('VV_FINAL_BCE', form)
    rng = np.random.default_rng(side)
    dt, tdt = L.DTYPES[dtname], (torch.float32 if dtname == 'f32' else torch.bfloat16)
    x = rng.standard_normal((B, side, side, side, 64)).astype(np.float32)
    w = (rng.standard_normal((4, 4, 4, 1, 64)) * 0.3).astype(np.float32)
    if dtname == 'bf16':
        x, w = _bf16_round(x), _bf16_round(w)   # the bf16 path feeds MFMA: both operands are bf16
    D = 2 * side
    y = (rng.random((B, D, D, D, 1)) < 0.3).astype(np.float32)
    lg = no.conv3d_transpose_same(x.astype(np.float64), w.astype(np.float64), 2)
    pr = no.sigmoid(lg)
    bce = no.binary_loss(pr.astype(np.float32), y, gamma=0.6)
    tp, fp, fn = no.voxel_precision_recall(y, pr.astype(np.float32))
    nb = L.load().vv_convT3d_final_bce_workspace_bytes(B, side)
    ws = torch.empty(max(nb, 16), dtype=torch.uint8, device=DEV)
    probs = torch.empty(B, D, D, D, 1, dtype=torch.float32, device=DEV)
    logits = torch.empty_like(probs)
    stats = torch.empty(B, 4, dtype=torch.float32, device=DEV)
    xd, wd, yd = _dev(x, tdt), _dev(w), _dev(y)
    L.call('vv_convT3d_final_bce_fwd', L.ptr(xd), L.ptr(wd), L.ptr(yd), L.ptr(probs), L.ptr(logits),
           L.ptr(stats), B, side, 64, 0.6, 1e-7, dt, L.ptr(ws), ws.numel(), _st())
    torch.cuda.synchronize()
    glg = logits.cpu().numpy().astype(np.float64)
    assert np.abs(glg - lg).max() < 2e-5 * max(1.0, np.abs(lg).max())
    np.testing.assert_allclose(probs.cpu().numpy(), pr, rtol=0, atol=1e-5)  # dp <= 0.25 * dlogit
    s = stats.cpu().numpy().astype(np.float64)
    # clip(sigmoid(l)) then log(1-p) (function.py:79-80) is ill-conditioned near p -> 1 (one ulp of p moves a
    # saturated term by percents), so the loss FORMULA is checked on the kernel's own float32 probabilities ...
    pg = probs.cpu().numpy()
    q = np.clip(pg, np.float32(1e-7), np.float32(1.0) - np.float32(1e-7))
    om = (np.float32(1.0) - q).astype(np.float64)
    bce_self = -(0.6 * y * np.log(q.astype(np.float64)) + 0.4 * (1 - y) * np.log(om)).reshape(B, -1).sum(-1)
    np.testing.assert_allclose(s[:, 0], bce_self, rtol=2e-5)
    # ... and against the oracle's loss with the tolerance that conditioning allows
    np.testing.assert_allclose(s[:, 0], bce, rtol=2e-3)
    flips = int(((glg >= 0) != (lg >= 0)).sum())
    assert flips == 0 or np.abs(lg[(glg >= 0) != (lg >= 0)]).max() < 1e-5
    for k, r in ((1, tp), (2, fp), (3, fn)):
        assert np.abs(s[:, k] - r).max() <= flips
    out = torch.empty(4, dtype=torch.float32, device=DEV)
    L.call('vv_shape_metrics', L.ptr(stats), L.ptr(out), B, _st())
    torch.cuda.synchronize()
    o = out.cpu().numpy()
    prr, rcc = no.pr_rc(s[:, 1], s[:, 2], s[:, 3])
    np.testing.assert_allclose(o[:3], [s[:, 0].mean(), prr, rcc], rtol=1e-5)
    # the same layer with the batch metrics folded into its reduction launch: identical per-sample sums, same metrics
    stats2, out2 = torch.empty_like(stats), torch.empty(4, dtype=torch.float32, device=DEV)
    L.call('vv_convT3d_final_bce_metrics_fwd', L.ptr(xd), L.ptr(wd), L.ptr(yd), L.ptr(probs), L.ptr(logits), L.ptr(stats2), L.ptr(out2),
           B, side, 64, 0.6, 1e-7, dt, L.ptr(ws), ws.numel(), _st())
    torch.cuda.synchronize()
    # (the fused launch adds a sample's partial blocks in block order, the two-launch form lane-strided + a wave tree: the counts
    # are exact either way, the loss may differ in its last bits)
    assert torch.equal(stats2[:, 1:], stats[:, 1:])
    np.testing.assert_allclose(stats2[:, 0].cpu().numpy(), stats[:, 0].cpu().numpy(), rtol=2e-6)
    np.testing.assert_allclose(out2.cpu().numpy(), o, rtol=2e-6)


@pytest.mark.parametrize('B,K5,Lz,lin,n1,variational', [(256, 4096, 64, 64, 4096, True), (37, 4096, 64, 64, 4096, False),
                                                        (5, 1024, 32, 512, 2048, True), (19, 8200, 64, 96, 80, True)])
def test_latent_tail(L, B, K5, Lz, lin, n1, variational):
    """encoder tail -> clip | sampling | KL -> Dense + BN + act -> first decoder layer (dense panel) + BN + act in two launches
    (latent_tail.hip) against the float64 definition on the same bf16 operands."""
    rng = np.random.default_rng(B + K5)
    E = 2 * Lz if variational else Lz
    assert L.load().vv_latent_tail_supported(K5, E, Lz, lin, n1, int(variational), L.VV_BF16)
    h = _bf16_round(rng.standard_normal((B, K5)).astype(np.float32))
    w5 = _bf16_round((rng.standard_normal((E, K5)) / np.sqrt(K5)).astype(np.float32)) * 3
    w5 = _bf16_round(w5)
    eps = rng.standard_normal((B, Lz)).astype(np.float32)
    wd = _bf16_round((rng.standard_normal((lin, Lz)) / np.sqrt(Lz)).astype(np.float32))
    w1 = _bf16_round((rng.standard_normal((n1, lin)) / np.sqrt(lin)).astype(np.float32))
    scd, shd = rng.uniform(0.5, 1.5, lin).astype(np.float32), rng.normal(0, 0.3, lin).astype(np.float32)
    sc1, sh1 = rng.uniform(0.5, 1.5, n1).astype(np.float32), rng.normal(0, 0.3, n1).astype(np.float32)
    enc = h.astype(np.float64) @ w5.astype(np.float64).T
    if variational:
        mu, lv = no.split_mean_logvar(enc, Lz)
        zr = mu + np.sqrt(np.exp(lv)) * eps
        klr = no.kl_loss(mu, lv, np.zeros_like(mu), np.zeros_like(lv))
    else:
        zr, klr = enc, None
    bt = torch.bfloat16
    hd, w5d, wdd, w1d = _dev(h, bt), _dev(w5, bt), _dev(wd, bt), _dev(w1, bt)
    epsd, scdd, shdd, sc1d, sh1d = _dev(eps), _dev(scd), _dev(shd), _dev(sc1), _dev(sh1)
    enc_out = torch.empty(B, E, dtype=torch.float32, device=DEV)
    z = torch.empty(B, Lz, dtype=torch.float32, device=DEV)
    zb = torch.empty(B, Lz, dtype=bt, device=DEV)
    kl = torch.full((B,), float('nan'), dtype=torch.float32, device=DEV)
    h1 = torch.full((B, n1), float('nan'), dtype=bt, device=DEV)
    ws = torch.empty(max(L.load().vv_latent_tail_workspace_bytes(B, K5, E, n1), 16), dtype=torch.uint8, device=DEV)
    L.call('vv_latent_tail_fwd', L.ptr(hd), L.ptr(w5d), None, L.ptr(epsd) if variational else None, L.ptr(wdd), L.ptr(scdd), L.ptr(shdd),
           L.ptr(w1d), L.ptr(sc1d), L.ptr(sh1d), L.ptr(enc_out), L.ptr(z), L.ptr(zb), L.ptr(kl) if variational else None, L.ptr(h1),
           B, K5, E, Lz, lin, n1, int(variational), 1, L.VV_BF16, L.ptr(ws), ws.numel(), _st())
    torch.cuda.synchronize()
    np.testing.assert_allclose(enc_out.cpu().numpy(), enc, rtol=0, atol=2e-5 * max(1.0, np.abs(enc).max()))
    np.testing.assert_allclose(z.cpu().numpy(), zr, rtol=2e-5, atol=2e-5 * max(1.0, np.abs(zr).max()))
    if variational:
        np.testing.assert_allclose(kl.cpu().numpy(), klr, rtol=2e-5, atol=1e-4)
    assert torch.equal(zb, z.to(bt))
    # the two dense layers on the kernel's own bf16 latent (what the split path feeds them too)
    zq = zb.float().cpu().numpy().astype(np.float64)
    t = no.activation(zq @ wd.astype(np.float64).T * scd + shd, 'elu')
    tq = _bf16_round(t.astype(np.float32)).astype(np.float64)
    ref = no.activation(tq @ w1.astype(np.float64).T * sc1 + sh1, 'elu')
    _check(h1, ref, 'bf16', 'latent_tail')


# the last stride-2 encoder layer + the latent tail as ONE call (round 4): the layer's split-K partial sums are summed by the tail's
# first kernel.  Against the two calls it replaces (vv_conv3d_k4s2_pos_fwd -> vv_latent_tail_fwd: same share order, same bf16 rounding
# of the layer output, so the encoder output differs only by the float32 summation order of the tail GEMM) and, for the layer
# itself, against the float64 definition through the encoder output.
@pytest.mark.parametrize('B,cin,cout,Lz,variational', [(256, 256, 512, 64, True), (37, 64, 256, 32, True), (300, 128, 512, 64, False),
                                                       (5, 64, 256, 64, True)])
def test_conv_pos_latent_tail_fused(L, B, cin, cout, Lz, variational):
    rng = np.random.default_rng(B * 7 + cin)
    E = 2 * Lz if variational else Lz
    K5, lin, n1 = 8 * cout, 64, 4096
    assert L.load().vv_conv_pos_latent_tail_supported(cin, cout, E, Lz, lin, n1, int(variational), L.VV_BF16)
    x = _bf16_round(rng.standard_normal((B, 4, 4, 4, cin)).astype(np.float32))
    w4 = _bf16_round((rng.standard_normal((4, 4, 4, cin, cout)) / np.sqrt(27 * cin)).astype(np.float32))
    sc4, sh4 = rng.uniform(0.5, 1.5, cout).astype(np.float32), rng.normal(0, 0.3, cout).astype(np.float32)
    w5 = _bf16_round((rng.standard_normal((E, K5)) / np.sqrt(K5)).astype(np.float32) * 3)
    eps = rng.standard_normal((B, Lz)).astype(np.float32)
    wd = _bf16_round((rng.standard_normal((lin, Lz)) / np.sqrt(Lz)).astype(np.float32))
    w1 = _bf16_round((rng.standard_normal((n1, lin)) / np.sqrt(lin)).astype(np.float32))
    scd, shd = rng.uniform(0.5, 1.5, lin).astype(np.float32), rng.normal(0, 0.3, lin).astype(np.float32)
    sc1, sh1 = rng.uniform(0.5, 1.5, n1).astype(np.float32), rng.normal(0, 0.3, n1).astype(np.float32)
    bt = torch.bfloat16
    xd, w4d, sc4d, sh4d = _dev(x, bt), _dev(w4), _dev(sc4), _dev(sh4)
    w5d, wdd, w1d = _dev(w5, bt), _dev(wd, bt), _dev(w1, bt)
    epsd, scdd, shdd, sc1d, sh1d = _dev(eps), _dev(scd), _dev(shd), _dev(sc1), _dev(sh1)
    wp = torch.empty(64 * cin * cout, dtype=bt, device=DEV)
    L.call('vv_pack_conv_k4_skip', L.ptr(w4d), L.ptr(wp), cin, cout, _st())
    lib = L.load()

    def outs():
        return (torch.full((B, E), float('nan'), dtype=torch.float32, device=DEV), torch.full((B, Lz), float('nan'), dtype=torch.float32, device=DEV),
                torch.empty(B, Lz, dtype=bt, device=DEV), torch.full((B,), float('nan'), dtype=torch.float32, device=DEV),
                torch.full((B, n1), float('nan'), dtype=bt, device=DEV))

    # the two calls
    ws = torch.empty(max(lib.vv_conv3d_k4s2_pos_workspace_bytes(B, cin, cout), 16), dtype=torch.uint8, device=DEV)
    h4 = torch.full((B, 2, 2, 2, cout), float('nan'), dtype=bt, device=DEV)
    L.call('vv_conv3d_k4s2_pos_fwd', L.ptr(xd), L.ptr(wp), L.ptr(sc4d), L.ptr(sh4d), L.ptr(h4), B, 4, cin, cout, 1, L.VV_BF16, L.ptr(ws), ws.numel(), _st())
    e0, z0, zb0, kl0, h10 = outs()
    ws2 = torch.empty(max(lib.vv_latent_tail_workspace_bytes(B, K5, E, n1), 16), dtype=torch.uint8, device=DEV)
    L.call('vv_latent_tail_fwd', L.ptr(h4), L.ptr(w5d), None, L.ptr(epsd) if variational else None, L.ptr(wdd), L.ptr(scdd), L.ptr(shdd),
           L.ptr(w1d), L.ptr(sc1d), L.ptr(sh1d), L.ptr(e0), L.ptr(z0), L.ptr(zb0), L.ptr(kl0) if variational else None, L.ptr(h10),
           B, K5, E, Lz, lin, n1, int(variational), 1, L.VV_BF16, L.ptr(ws2), ws2.numel(), _st())
    # the fused call
    e1, z1, zb1, kl1, h11 = outs()
    need = lib.vv_conv_pos_latent_tail_workspace_bytes(B, cin, cout, E)
    assert need > 0
    ws3 = torch.empty(need, dtype=torch.uint8, device=DEV)
    L.call('vv_conv_pos_latent_tail_fwd', L.ptr(xd), L.ptr(wp), L.ptr(sc4d), L.ptr(sh4d), cin, cout, L.ptr(w5d), None,
           L.ptr(epsd) if variational else None, L.ptr(wdd), L.ptr(scdd), L.ptr(shdd), L.ptr(w1d), L.ptr(sc1d), L.ptr(sh1d), L.ptr(e1), L.ptr(z1),
           L.ptr(zb1), L.ptr(kl1) if variational else None, L.ptr(h11), B, E, Lz, lin, n1, int(variational), 1, L.VV_BF16, L.ptr(ws3), ws3.numel(), _st())
    torch.cuda.synchronize()
    en0, en1 = e0.cpu().numpy().astype(np.float64), e1.cpu().numpy().astype(np.float64)
    tol = 2e-5 * max(1.0, np.abs(en0).max())
    np.testing.assert_allclose(en1, en0, rtol=0, atol=tol)            # same operands, float32 summation order only
    # the float64 definition of the tail on the bf16 layer output the unfused kernel stored
    ref = h4.float().cpu().numpy().astype(np.float64).reshape(B, K5) @ w5.astype(np.float64).T
    np.testing.assert_allclose(en1, ref, rtol=0, atol=tol)
    np.testing.assert_allclose(z1.cpu().numpy(), z0.cpu().numpy(), rtol=2e-5, atol=tol)
    if variational:
        np.testing.assert_allclose(kl1.cpu().numpy(), kl0.cpu().numpy(), rtol=1e-4, atol=1e-3)
    assert torch.equal(zb1, z1.to(bt))
    d = (h11.float() - h10.float()).abs().max().item()
    assert d <= 0.05 * max(1.0, h10.float().abs().max().item()), d       # bf16 outputs of two dense layers on latents one float32 ulp apart
    # too small a workspace is refused, nothing is launched
    assert lib.vv_conv_pos_latent_tail_fwd(L.ptr(xd), L.ptr(wp), L.ptr(sc4d), L.ptr(sh4d), cin, cout, L.ptr(w5d), None, L.ptr(epsd), L.ptr(wdd),
                                           L.ptr(scdd), L.ptr(shdd), L.ptr(w1d), L.ptr(sc1d), L.ptr(sh1d), L.ptr(e1), L.ptr(z1), L.ptr(zb1), L.ptr(kl1),
                                           L.ptr(h11), B, E, Lz, lin, n1, int(variational), 1, L.VV_BF16, L.ptr(ws3), need - 16, _st()) == -5


def test_reparam_kl(L):
    rng = np.random.default_rng(0)
    for B, Lz in ((5, 64), (3, 16), (2, 100)):
        e = (rng.standard_normal((B, 2 * Lz)) * 6).astype(np.float32)
        eps = rng.standard_normal((B, Lz)).astype(np.float32)
        mask = (rng.random((B, Lz)) > 0.3).astype(np.float32)
        mu, lv = no.split_mean_logvar(e.astype(np.float64), Lz)
        for use_mask in (False, True):
            z = torch.empty(B, Lz, dtype=torch.float32, device=DEV)
            zb = torch.empty(B, Lz, dtype=torch.bfloat16, device=DEV)
            kl = torch.empty(B, dtype=torch.float32, device=DEV)
            mo, lo = torch.empty_like(z), torch.empty_like(z)
            scale = 1.0 / (1.0 - 0.3) if use_mask else 1.0
            ed, epsd, md = _dev(e), _dev(eps), _dev(mask)
            L.call('vv_reparam_kl_fwd', L.ptr(ed), L.ptr(epsd), L.ptr(md) if use_mask else None, scale, L.ptr(z),
                   L.ptr(zb), L.VV_BF16, L.ptr(kl), L.ptr(mo), L.ptr(lo), B, Lz, _st())
            torch.cuda.synchronize()
            zr = no.sampling(mu, lv, eps)
            if use_mask:
                zr = zr * mask * scale
            np.testing.assert_allclose(z.cpu().numpy(), zr, rtol=2e-5, atol=2e-5)
            np.testing.assert_allclose(kl.cpu().numpy(), no.kl_loss(mu, lv, 0 * mu, 0 * lv), rtol=2e-5)
            np.testing.assert_allclose(lo.cpu().numpy(), lv, rtol=0, atol=0)
            assert torch.equal(zb, z.to(torch.bfloat16))


def test_error_codes(L):
    lib = L.load()
    x = torch.zeros(16, device=DEV)
    assert lib.vv_conv3d_k4s2_fwd(None, None, None, None, None, 1, 8, 64, 64, 1, 0, None, 0, None) == -1
    assert lib.vv_conv3d_k4s2_fwd(L.ptr(x), L.ptr(x), None, None, L.ptr(x), 1, 6, 64, 64, 1, 0, None, 0, None) == -2
    assert lib.vv_conv3d_k4s2_fwd(L.ptr(x), L.ptr(x), None, None, L.ptr(x), 1, 8, 48, 64, 1, 1, None, 0, None) == -2
    assert lib.vv_conv3d_k4s2_fwd(L.ptr(x), L.ptr(x), None, None, L.ptr(x), 1, 8, 64, 64, 1, 7, None, 0, None) == -3
    assert lib.vv_dense_fwd(L.ptr(x), L.ptr(x), None, None, L.ptr(x), 2, 64, 8192, 0, 1, 0, None, 0, None) == -5
    with pytest.raises(L.VoxVaeError):
        L.call('vv_shape_metrics', None, None, 1, None)


@pytest.mark.parametrize('variant', ['8', '4', '2'])
@pytest.mark.parametrize('B,side', [(2, 8), (1, 16), (3, 8)])
def test_convT3d_k4s2_direct(L, B, side, variant, monkeypatch):
    """LDS-resident input-tile variant of the widest decoder layer (bf16, 128 -> 64): 8 waves x 1 parity (default), and the
    two 4-wave x 2-parity forms."""
    monkeypatch.setenv('VV_DIRECT_MT', variant)
    cin, cout = 128, 64
    assert L.load().vv_convT3d_k4s2_direct_supported(side, cin, cout, L.VV_BF16) == 1
    assert L.load().vv_convT3d_k4s2_direct_supported(4, cin, cout, L.VV_BF16) == 0
    rng = np.random.default_rng(side + B)
    x = _bf16_round(rng.standard_normal((B, side, side, side, cin)).astype(np.float32))
    w = _bf16_round((rng.standard_normal((4, 4, 4, cout, cin)) / np.sqrt(8 * cin)).astype(np.float32))
    scale = rng.uniform(0.5, 1.5, cout).astype(np.float32)
    shift = rng.normal(0, 0.3, cout).astype(np.float32)
    ref = no.activation(no.conv3d_transpose_same(x.astype(np.float64), w.astype(np.float64), 2) * scale + shift, 'elu')
    xd, wd, scd, shd = _dev(x, torch.bfloat16), _dev(w), _dev(scale), _dev(shift)
    wf = torch.empty(64 * cin * cout, dtype=torch.bfloat16, device=DEV)
    L.call('vv_pack_convT_k4s2_frag', L.ptr(wd), L.ptr(wf), cin, cout, _st())
    y = torch.full((B, 2 * side, 2 * side, 2 * side, cout), -7.0, dtype=torch.bfloat16, device=DEV)
    L.call('vv_convT3d_k4s2_direct_fwd', L.ptr(xd), L.ptr(wf), L.ptr(scd), L.ptr(shd), L.ptr(y), B, side, cin, cout, 1, L.VV_BF16, _st())
    torch.cuda.synchronize()
    _check(y, ref, 'bf16', 'convT3d_k4s2_direct')


@pytest.mark.parametrize('shape', [16, 32, 8])         # MFMA shape: 16x16x32 (default) / 32x32x16 / 8 = 16x16x32 as two 4-wave workgroups per CU
@pytest.mark.parametrize('B,side,act', [(2, 16, 1), (1, 32, 1), (3, 16, 0), (5, 16, 2)])
def test_conv3d_k4s2_direct(L, B, side, act, shape, monkeypatch):
    monkeypatch.setenv('VV_CD_SHAPE', str(shape))
    """LDS-resident phase-tile variant of the widest encoder layer (bf16, 64 -> 128); also bit-compared with the
    implicit-GEMM kernel's result on the same inputs (same bf16 operands, different summation order)."""
    cin, cout = 64, 128
    assert L.load().vv_conv3d_k4s2_direct_supported(side, cin, cout, L.VV_BF16) == 1
    assert L.load().vv_conv3d_k4s2_direct_supported(8, cin, cout, L.VV_BF16) == 0
    assert L.load().vv_conv3d_k4s2_direct_supported(side, cin, cout, L.VV_F32) == 0
    rng = np.random.default_rng(side + 10 * B)
    x = _bf16_round(rng.standard_normal((B, side, side, side, cin)).astype(np.float32))
    w = _bf16_round((rng.standard_normal((4, 4, 4, cin, cout)) / np.sqrt(64 * cin)).astype(np.float32))
    scale = rng.uniform(0.5, 1.5, cout).astype(np.float32)
    shift = rng.normal(0, 0.3, cout).astype(np.float32)
    actname = {0: 'none', 1: 'elu', 2: 'relu'}[act]
    ref = no.activation(no.conv3d_same(x.astype(np.float64), w.astype(np.float64), 2) * scale + shift, actname)
    xd, wd, scd, shd = _dev(x, torch.bfloat16), _dev(w), _dev(scale), _dev(shift)
    wp = torch.empty(cout, 64 * cin, dtype=torch.bfloat16, device=DEV)
    L.call('vv_pack_conv_k4', L.ptr(wd), L.ptr(wp), cin, cout, L.VV_BF16, _st())
    so = side // 2
    y = torch.full((B, so, so, so, cout), -7.0, dtype=torch.bfloat16, device=DEV)
    L.call('vv_conv3d_k4s2_direct_fwd', L.ptr(xd), L.ptr(wp), L.ptr(scd), L.ptr(shd), L.ptr(y), B, side, cin, cout, act, L.VV_BF16, _st())
    torch.cuda.synchronize()
    _check(y, ref, 'bf16', 'conv3d_k4s2_direct')
    nb = L.load().vv_conv3d_k4s2_workspace_bytes(B, side, cin, cout, L.VV_BF16)
    ws = torch.empty(max(nb, 16), dtype=torch.uint8, device=DEV)
    y2 = torch.empty_like(y)
    L.call('vv_conv3d_k4s2_fwd', L.ptr(xd), L.ptr(wp), L.ptr(scd), L.ptr(shd), L.ptr(y2), B, side, cin, cout, act, L.VV_BF16,
           L.ptr(ws), ws.numel(), _st())
    torch.cuda.synchronize()
    d = (y.float() - y2.float()).abs().max().item()
    assert d <= 2e-2, 'direct vs implicit-GEMM: %.3e' % d


def _wgrad_conv_ref(src, g):
    """dW[t][ci][co] = sum over (b, o) of src[b, 2o-1+t, ci] * g[b, o, co] (zero padding), float64."""
    B, S, _, _, cin = src.shape
    o, cout = S // 2, g.shape[-1]
    p = np.zeros((B, S + 2, S + 2, S + 2, cin))
    p[:, 1:-1, 1:-1, 1:-1] = src
    dw = np.zeros((4, 4, 4, cin, cout))
    for td in range(4):
        for th in range(4):
            for tw in range(4):
                win = p[:, td:td + 2 * o:2, th:th + 2 * o:2, tw:tw + 2 * o:2]          # [B,o,o,o,cin]
                dw[td, th, tw] = np.einsum('bdhwi,bdhwo->io', win, g)
    return dw


@pytest.mark.parametrize('adt,gdt', [('f32', 'f32'), ('bf16', 'bf16'), ('bf16', 'f32')])
def test_wgrad_kernels(L, adt, gdt, monkeypatch):
    """Weight gradients dW = A^T G: float32 operands on the exact-f32 MFMA, bf16 operands on the bf16 MFMA with
    transposed LDS reads (ds_read_b64_tr_b16), mixed operands widened onto the f32 path.  Dense (ragged M / N / rows) and
    the strided-conv gather (cin = 64: a 128-column tile spans two taps; cin = 128)."""
    rng = np.random.default_rng(7)
    T = {'f32': torch.float32, 'bf16': torch.bfloat16}
    rnd = lambda a, d: _bf16_round(a) if d == 'bf16' else a.astype(np.float32)
    for rows, m, n in ((300, 160, 96), (64, 128, 4096), (1000, 32, 32)):
        a = rnd(rng.standard_normal((rows, m)).astype(np.float32), adt)
        g = rnd(rng.standard_normal((rows, n)).astype(np.float32), gdt)
        ref = a.astype(np.float64).T @ g.astype(np.float64)
        ad, gd = _dev(a, T[adt]), _dev(g, T[gdt])
        out = torch.full((m, n), 7.0, dtype=torch.float32, device=DEV)
        ws = torch.empty(L.load().vv_wgrad_workspace_bytes(rows, m, n), dtype=torch.uint8, device=DEV)
        L.call('vv_wgrad_dense', L.ptr(ad), L.ptr(gd), L.ptr(out), rows, m, n, m, L.DTYPES[adt], L.DTYPES[gdt], L.ptr(ws), ws.numel(), _st())
        torch.cuda.synchronize()
        err = np.abs(out.cpu().numpy() - ref).max()
        assert err <= 2e-5 * np.abs(ref).max() + 1e-5, ('dense', rows, m, n, err)
    for B, side, cin, cout in ((3, 8, 64, 128), (2, 4, 128, 64), (5, 8, 64, 32)):
        src = rnd(rng.standard_normal((B, side, side, side, cin)).astype(np.float32), adt)
        o = side // 2
        g = rnd(rng.standard_normal((B, o, o, o, cout)).astype(np.float32), gdt)
        ref = _wgrad_conv_ref(src.astype(np.float64), g.astype(np.float64))
        sd, gd = _dev(src, T[adt]), _dev(g, T[gdt])
        out = torch.full((4, 4, 4, cin, cout), 7.0, dtype=torch.float32, device=DEV)
        ws = torch.empty(L.load().vv_wgrad_workspace_bytes(B * o ** 3, 64 * cin, cout), dtype=torch.uint8, device=DEV)
        L.call('vv_wgrad_conv_k4s2', L.ptr(sd), L.ptr(gd), L.ptr(out), B, side, cin, cout, L.DTYPES[adt], L.DTYPES[gdt], L.ptr(ws),
               ws.numel(), _st())
        torch.cuda.synchronize()
        err = np.abs(out.cpu().numpy() - ref).max()
        assert err <= 2e-5 * np.abs(ref).max() + 1e-5, ('conv', B, side, cin, cout, err)
    last = (sd, gd, out, ws, B, side, cin, cout, np.abs(ref).max())
    # single input channel (first conv / last transposed conv): float32 source grid, g float32 or bf16 (bf16 im2col path)
    for B, side, cout in ((3, 16, 64), (2, 8, 32)):
        src = (rng.random((B, side, side, side, 1)) < 0.3).astype(np.float32) * rng.standard_normal((B, side, side, side, 1)).astype(np.float32)
        if gdt == 'bf16':
            src = _bf16_round(src)                       # the im2col rows are bf16
        o = side // 2
        g = rnd(rng.standard_normal((B, o, o, o, cout)).astype(np.float32), gdt)
        ref = _wgrad_conv_ref(src.astype(np.float64), g.astype(np.float64))
        sd, gd = _dev(src), _dev(g, T[gdt])
        out = torch.full((4, 4, 4, 1, cout), 7.0, dtype=torch.float32, device=DEV)
        ws = torch.empty(L.load().vv_wgrad_workspace_bytes(B * o ** 3, 64, cout), dtype=torch.uint8, device=DEV)
        L.call('vv_wgrad_conv_k4s2', L.ptr(sd), L.ptr(gd), L.ptr(out), B, side, 1, cout, L.VV_F32, L.DTYPES[gdt], L.ptr(ws), ws.numel(), _st())
        torch.cuda.synchronize()
        err = np.abs(out.cpu().numpy() - ref).max()
        assert err <= 2e-5 * np.abs(ref).max() + 1e-5, ('conv cin=1', B, side, cout, err)
    if adt == gdt == 'bf16':      # same operands through the widening f32 kernel: the two paths agree to float32 rounding
        sd, gd, out, ws, B, side, cin, cout, refmax = last
        monkeypatch.setenv('VV_WGRAD_F32', '1')
        out2 = torch.empty_like(out)
        L.call('vv_wgrad_conv_k4s2', L.ptr(sd), L.ptr(gd), L.ptr(out2), B, side, cin, cout, L.DTYPES[adt], L.DTYPES[gdt], L.ptr(ws),
               ws.numel(), _st())
        torch.cuda.synchronize()
        assert (out - out2).abs().max().item() <= 2e-5 * refmax


@pytest.mark.parametrize('dtname', ['f32', 'bf16'])
@pytest.mark.parametrize('rows,C', [(4096, 64), (2048, 512), (777, 128), (5, 256), (300, 1024), (20000, 64)])
def test_batchnorm_train_ops(L, dtname, rows, C):
    """Training-mode BatchNorm + activation: batch statistics (biased variance, Keras momentum update), forward and the
    three-output backward, against float64 numpy (BatchNormalization semantics as restated in oracle/torch_oracle.py)."""
    rng = np.random.default_rng(rows + C)
    dt, tdt = L.DTYPES[dtname], (torch.float32 if dtname == 'f32' else torch.bfloat16)
    x = (rng.standard_normal((rows, C)) * rng.uniform(0.5, 2.0, C) + rng.normal(0, 1, C)).astype(np.float32)
    dy = rng.standard_normal((rows, C)).astype(np.float32)
    if dtname == 'bf16':
        x, dy = _bf16_round(x), _bf16_round(dy)
    gamma, beta = rng.uniform(0.5, 1.5, C).astype(np.float32), rng.normal(0, 0.3, C).astype(np.float32)
    mm0, mv0 = rng.normal(0, 1, C).astype(np.float32), rng.uniform(0.5, 2, C).astype(np.float32)
    eps, mom, act = 1e-3, 0.99, 1
    x64 = x.astype(np.float64)
    mean, var = x64.mean(0), x64.var(0)
    rstd = 1.0 / np.sqrt(var + eps)
    scale, shift = gamma * rstd, beta - mean * gamma * rstd
    xd, dyd = _dev(x, tdt), _dev(dy, tdt)
    gd, bd, mmd, mvd = _dev(gamma), _dev(beta), _dev(mm0), _dev(mv0)
    o = {n: torch.empty(C, dtype=torch.float32, device=DEV) for n in ('mean', 'var', 'rstd', 'scale', 'shift', 'dgamma', 'dbeta')}
    nb = L.load().vv_bn_workspace_bytes(rows, C)
    ws = torch.empty(max(nb, 16), dtype=torch.uint8, device=DEV)
    L.call('vv_bn_train_stats', L.ptr(xd), rows, C, L.ptr(gd), L.ptr(bd), eps, mom, L.ptr(o['mean']), L.ptr(o['var']), L.ptr(o['rstd']),
           L.ptr(o['scale']), L.ptr(o['shift']), L.ptr(mmd), L.ptr(mvd), dt, L.ptr(ws), ws.numel(), _st())
    torch.cuda.synchronize()
    for name, ref in (('mean', mean), ('var', var), ('rstd', rstd), ('scale', scale), ('shift', shift)):
        got = o[name].cpu().numpy()
        assert np.abs(got - ref).max() <= 2e-5 * max(1.0, np.abs(ref).max()), name
    assert np.abs(mmd.cpu().numpy() - (mm0 * mom + mean * (1 - mom))).max() <= 1e-5
    assert np.abs(mvd.cpu().numpy() - (mv0 * mom + var * (1 - mom))).max() <= 1e-5
    # forward / backward with the device's own statistics
    sc, sh, mu, rs = (o[n].cpu().numpy().astype(np.float64) for n in ('scale', 'shift', 'mean', 'rstd'))
    u = x64 * sc + sh
    y = torch.empty_like(xd)
    L.call('vv_bn_act_fwd', L.ptr(xd), L.ptr(o['scale']), L.ptr(o['shift']), L.ptr(y), rows, C, act, dt, _st())
    torch.cuda.synchronize()
    _check(y, no.activation(u, 'elu'), dtname, 'bn_act_fwd')
    du = dy.astype(np.float64) * np.where(u > 0, 1.0, np.exp(np.minimum(u, 0)))
    xh = (x64 - mu) * rs
    dbeta, dgamma = du.sum(0), (du * xh).sum(0)
    dx_ref = sc * (du - dbeta / rows - xh * dgamma / rows)
    dx = torch.empty_like(xd)
    L.call('vv_bn_act_bwd', L.ptr(xd), L.ptr(dyd), L.ptr(o['scale']), L.ptr(o['shift']), L.ptr(o['mean']), L.ptr(o['rstd']),
           L.ptr(o['dgamma']), L.ptr(o['dbeta']), L.ptr(dx), rows, C, act, dt, L.ptr(ws), ws.numel(), _st())
    torch.cuda.synchronize()
    tol = 1e-4 if dtname == 'f32' else 2e-3       # bf16 mode evaluates exp with the hardware approximation; sums are float32
    assert np.abs(o['dbeta'].cpu().numpy() - dbeta).max() <= tol * max(1.0, np.abs(dbeta).max())
    assert np.abs(o['dgamma'].cpu().numpy() - dgamma).max() <= tol * max(1.0, np.abs(dgamma).max())
    _check(dx, dx_ref, dtname, 'bn_act_bwd')
    if dtname == 'bf16':
        assert L.load().vv_bn_act_fwd(L.ptr(xd), L.ptr(o['scale']), L.ptr(o['shift']), L.ptr(y), rows, 12, act, dt, _st()) == -2      # VV_ERR_SHAPE: bf16 rows are swept 8 channels per lane


@pytest.mark.parametrize('dtname', ['f32', 'bf16'])
@pytest.mark.parametrize('rows,C', [(256, 256), (3, 64), (1000, 40)])
def test_colsum(L, dtname, rows, C):
    rng = np.random.default_rng(rows)
    tdt = torch.float32 if dtname == 'f32' else torch.bfloat16
    x = rng.standard_normal((rows, C)).astype(np.float32)
    if dtname == 'bf16':
        x = _bf16_round(x)
    out = torch.empty(C, dtype=torch.float32, device=DEV)
    L.call('vv_colsum', L.ptr(_dev(x, tdt)), L.ptr(out), rows, C, L.DTYPES[dtname], _st())
    torch.cuda.synchronize()
    assert np.abs(out.cpu().numpy() - x.astype(np.float64).sum(0)).max() <= 1e-4


@pytest.mark.parametrize('B,side,cin,cout', [(2, 16, 64, 128), (5, 16, 64, 128), (16, 8, 128, 256), (17, 8, 64, 128), (1, 32, 64, 128),
                                             (4, 16, 128, 128)])
def test_wgrad_conv_phase_form(L, B, side, cin, cout, monkeypatch):
    """Phase-form weight gradient (wgrad_phase.hip: bf16, cin % 64 == 0, cout % 128 == 0, output side 4 / 8 / 16): one
    staged parity sub-grid tile feeds the 8 taps of the parity.  Against the float64 definition and against the
    reduction-GEMM kernel on the same bf16 operands (the two differ by float32 summation order only); odd batches leave
    the last box half empty."""
    rng = np.random.default_rng(B * 100 + side)
    src = _bf16_round(rng.standard_normal((B, side, side, side, cin)).astype(np.float32))
    o = side // 2
    g = _bf16_round(rng.standard_normal((B, o, o, o, cout)).astype(np.float32))
    ref = _wgrad_conv_ref(src.astype(np.float64), g.astype(np.float64))
    sd, gd = _dev(src, torch.bfloat16), _dev(g, torch.bfloat16)
    ws = torch.empty(L.load().vv_wgrad_workspace_bytes(B * o ** 3, 64 * cin, cout), dtype=torch.uint8, device=DEV)
    outs = []
    for off in (False, True):
        if off:
            monkeypatch.setenv('VV_NO_WGRAD_PHASE', '1')
        out = torch.full((4, 4, 4, cin, cout), 7.0, dtype=torch.float32, device=DEV)
        L.call('vv_wgrad_conv_k4s2', L.ptr(sd), L.ptr(gd), L.ptr(out), B, side, cin, cout, L.VV_BF16, L.VV_BF16, L.ptr(ws), ws.numel(), _st())
        torch.cuda.synchronize()
        outs.append(out.cpu().numpy())
        err = np.abs(outs[-1] - ref).max()
        assert err <= 2e-5 * np.abs(ref).max() + 1e-5, ('phase off' if off else 'phase on', err)
    assert np.abs(outs[0] - outs[1]).max() <= 2e-5 * np.abs(ref).max() + 1e-5


# ---------------------------------------------------------------------------------------------- fp8 (OCP e4m3fn) MFMA path
F8 = getattr(torch, 'float8_e4m3fn', None)


def _fp8_round(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(F8).float().numpy()


def _check_fp8_out(got_f8, ref, what):
    got = got_f8.float().cpu().numpy().astype(np.float64)
    # e4m3: 3 mantissa bits (half an ulp = 2^-4 relative), subnormal spacing 2^-9
    bad = np.abs(got - ref) > 0.0635 * np.abs(ref) + 2.5e-3
    assert not bad.any(), '%s: %d of %d beyond fp8 rounding, worst %.3e' % (what, bad.sum(), bad.size, np.abs(got - ref).max())


def test_launch_chunking_over_sample_ranges(L, monkeypatch):
    """Tensors past 2 GiB go out as several launches over sample ranges (32-bit buffer offsets).  VV_CHUNK_SAMPLES forces the
    same code path at a small batch: results are bit-identical to the single launch for the E2 kernel, the halo-tile D4 kernel
    and the sweep-form last layer (logits, probabilities and the four loss sums)."""
    B = 11
    g = torch.Generator(device=DEV).manual_seed(3)
    sc = torch.rand(128, device=DEV, generator=g) + 0.5
    sh = torch.randn(128, device=DEV, generator=g) * 0.3

    def both(fn):
        monkeypatch.delenv('VV_CHUNK_SAMPLES', raising=False)
        a = fn()
        monkeypatch.setenv('VV_CHUNK_SAMPLES', '4')
        b = fn()
        monkeypatch.delenv('VV_CHUNK_SAMPLES')
        torch.cuda.synchronize()
        for u, v in zip(a, b):
            assert torch.equal(u, v)

    x2 = torch.randn(B, 16, 16, 16, 64, device=DEV, generator=g).to(torch.bfloat16)
    w2 = (torch.randn(4, 4, 4, 64, 128, device=DEV, generator=g) / 64).contiguous()
    wp = torch.empty(128, 64 * 64, dtype=torch.bfloat16, device=DEV)
    L.call('vv_pack_conv_k4', L.ptr(w2), L.ptr(wp), 64, 128, L.VV_BF16, _st())

    def e2():
        y = torch.full((B, 8, 8, 8, 128), float('nan'), dtype=torch.bfloat16, device=DEV)
        L.call('vv_conv3d_k4s2_direct_fwd', L.ptr(x2), L.ptr(wp), L.ptr(sc), L.ptr(sh), L.ptr(y), B, 16, 64, 128, 1, L.VV_BF16, _st())
        torch.cuda.synchronize()
        return (y,)
    both(e2)

    x4 = torch.randn(B, 8, 8, 8, 128, device=DEV, generator=g).to(torch.bfloat16)
    w4 = (torch.randn(4, 4, 4, 64, 128, device=DEV, generator=g) / 32).contiguous()
    wf = torch.empty(64 * 128 * 64, dtype=torch.bfloat16, device=DEV)
    L.call('vv_pack_convT_k4s2_frag', L.ptr(w4), L.ptr(wf), 128, 64, _st())

    def d4():
        y = torch.full((B, 16, 16, 16, 64), float('nan'), dtype=torch.bfloat16, device=DEV)
        L.call('vv_convT3d_k4s2_direct_fwd', L.ptr(x4), L.ptr(wf), L.ptr(sc), L.ptr(sh), L.ptr(y), B, 8, 128, 64, 1, L.VV_BF16, _st())
        torch.cuda.synchronize()
        return (y,)
    both(d4)

    Bs = 37                                            # sweep form needs batch * tiles >= 128
    xa = torch.randn(Bs, 16, 16, 16, 64, device=DEV, generator=g).to(torch.bfloat16)
    w5 = (torch.randn(4, 4, 4, 1, 64, device=DEV, generator=g) / 16).contiguous()
    tgt = (torch.rand(Bs, 32, 32, 32, 1, device=DEV, generator=g) < 0.1).float().contiguous()
    ws = torch.empty(max(L.load().vv_convT3d_final_bce_workspace_bytes(Bs, 16), 16), dtype=torch.uint8, device=DEV)
    monkeypatch.setenv('VV_FINAL_BCE', 'sweep')

    def d5():
        lg = torch.full((Bs, 32, 32, 32, 1), float('nan'), device=DEV)
        pr = torch.full_like(lg, float('nan'))
        st = torch.empty(Bs, 4, device=DEV)
        L.call('vv_convT3d_final_bce_fwd', L.ptr(xa), L.ptr(w5), L.ptr(tgt), L.ptr(pr), L.ptr(lg), L.ptr(st), Bs, 16, 64, 0.6, 1e-7, L.VV_BF16,
               L.ptr(ws), ws.numel(), _st())
        torch.cuda.synchronize()
        return lg, pr, st
    both(d5)


def test_widest_layers_full_batch_forms_agree(L, monkeypatch):
    """B = 256: the whole-sample D4 kernel in both MFMA shapes and both parity splits equals the halo-tile kernel bit for bit, and
    the two MFMA shapes of the E2 kernel equal each other (same bf16 operands, same float32 accumulation order per output up to
    the MFMA's internal tree -- measured identical)."""
    B = 256
    g = torch.Generator(device=DEV).manual_seed(11)
    sc = torch.rand(128, device=DEV, generator=g) + 0.5
    sh = torch.randn(128, device=DEV, generator=g) * 0.3
    # D4: 8^3 x 128 -> 16^3 x 64
    x = torch.randn(B, 8, 8, 8, 128, device=DEV, generator=g).to(torch.bfloat16)
    w = (torch.randn(4, 4, 4, 64, 128, device=DEV, generator=g) / 32).contiguous()
    wf = torch.empty(64 * 128 * 64, dtype=torch.bfloat16, device=DEV)
    wk = torch.empty_like(wf)
    L.call('vv_pack_convT_k4s2_frag', L.ptr(w), L.ptr(wf), 128, 64, _st())
    L.call('vv_pack_convT_k4s2_skip', L.ptr(w), L.ptr(wk), 128, 64, _st())
    y0 = torch.full((B, 16, 16, 16, 64), float('nan'), dtype=torch.bfloat16, device=DEV)
    L.call('vv_convT3d_k4s2_direct_fwd', L.ptr(x), L.ptr(wf), L.ptr(sc), L.ptr(sh), L.ptr(y0), B, 8, 128, 64, 1, L.VV_BF16, _st())
    for shape in ('4', '16', '32'):
        for ps in ('1', '2'):
            monkeypatch.setenv('VV_CTW_SHAPE', shape)
            monkeypatch.setenv('VV_CTW_PS', ps)
            y1 = torch.full_like(y0, float('nan'))
            L.call('vv_convT3d_k4s2_whole_fwd', L.ptr(x), L.ptr(wk), L.ptr(sc), L.ptr(sh), L.ptr(y1), B, 8, 128, 64, 1, L.VV_BF16, _st())
            torch.cuda.synchronize()
            assert torch.equal(y0, y1), 'whole-sample D4, MFMA shape %s, %s parity split(s)' % (shape, ps)
    # E2: 16^3 x 64 -> 8^3 x 128
    x2 = torch.randn(B, 16, 16, 16, 64, device=DEV, generator=g).to(torch.bfloat16)
    w2 = (torch.randn(4, 4, 4, 64, 128, device=DEV, generator=g) / 64).contiguous()
    wp = torch.empty(128, 64 * 64, dtype=torch.bfloat16, device=DEV)
    L.call('vv_pack_conv_k4', L.ptr(w2), L.ptr(wp), 64, 128, L.VV_BF16, _st())
    outs = []
    for shape in ('16', '32', '8'):
        monkeypatch.setenv('VV_CD_SHAPE', shape)
        y = torch.full((B, 8, 8, 8, 128), float('nan'), dtype=torch.bfloat16, device=DEV)
        L.call('vv_conv3d_k4s2_direct_fwd', L.ptr(x2), L.ptr(wp), L.ptr(sc), L.ptr(sh), L.ptr(y), B, 16, 64, 128, 1, L.VV_BF16, _st())
        torch.cuda.synchronize()
        outs.append(y)
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2]) and not torch.isnan(outs[0].float()).any()


def test_first_and_last_layer_full_batch_cross_forms(L, monkeypatch):
    """B = 256 (BASELINE configs[1]: several items / steps per persistent workgroup, two workgroups per CU): the plane form of the
    first layer equals its gather form bit for bit, and the depth-sweep forms of the last layer equal the 4^3-box form (logits to
    1e-5, loss sums to float32 summation error) and the float64 definition of the four loss sums on their own logits -- run twice (a race in a
    look-ahead load showed up only here: wrong logits in the last output plane, different from run to run)."""
    B = 256
    g = torch.Generator(device=DEV).manual_seed(5)
    x = (torch.rand(B, 32, 32, 32, 1, device=DEV, generator=g) < 0.1).float().contiguous()
    w = (torch.randn(4, 4, 4, 1, 64, device=DEV, generator=g) / 8).contiguous()
    sc = torch.rand(64, device=DEV, generator=g) + 0.5
    sh = torch.randn(64, device=DEV, generator=g) * 0.3
    wp = torch.empty(64, 64, dtype=torch.bfloat16, device=DEV)
    L.call('vv_pack_conv_k4', L.ptr(w), L.ptr(wp), 1, 64, L.VV_BF16, _st())

    def e1():
        y = torch.full((B, 16, 16, 16, 64), float('nan'), dtype=torch.bfloat16, device=DEV)
        L.call('vv_conv3d_first_fwd', L.ptr(x), L.ptr(wp), L.ptr(sc), L.ptr(sh), L.ptr(y), B, 32, 64, 1, L.VV_BF16, _st())
        torch.cuda.synchronize()
        return y
    y0, y1 = e1(), e1()                                # the chained plane form (four consecutive output planes per workgroup)
    monkeypatch.setenv('VV_FIRSTCONV_GATHER', '1')
    yg = e1()
    monkeypatch.delenv('VV_FIRSTCONV_GATHER')
    monkeypatch.setenv('VV_FIRSTCONV_NOCHAIN', '1')
    yp = e1()                                          # the plane form that loads all four input planes per item
    monkeypatch.delenv('VV_FIRSTCONV_NOCHAIN')
    assert torch.equal(y0, yg) and torch.equal(y1, yg) and torch.equal(yp, yg)
    # chains of two (batch 70: 1,120 items) and of three (batch 130: they cross sample boundaries, so an item in the middle of a
    # chain has no predecessor in its sample and loads both halves) on a slice of the same input
    for Bc in (70, 130):
        def e1c():
            y = torch.full((Bc, 16, 16, 16, 64), float('nan'), dtype=torch.bfloat16, device=DEV)
            L.call('vv_conv3d_first_fwd', L.ptr(x), L.ptr(wp), L.ptr(sc), L.ptr(sh), L.ptr(y), Bc, 32, 64, 1, L.VV_BF16, _st())
            torch.cuda.synchronize()
            return y
        assert torch.equal(e1c(), yg[:Bc])

    xa = torch.randn(B, 16, 16, 16, 64, device=DEV, generator=g).to(torch.bfloat16)
    w5 = (torch.randn(4, 4, 4, 1, 64, device=DEV, generator=g) / 16).contiguous()
    tgt = (torch.rand(B, 32, 32, 32, 1, device=DEV, generator=g) < 0.1).float().contiguous()
    ws = torch.empty(max(L.load().vv_convT3d_final_bce_workspace_bytes(B, 16), 16), dtype=torch.uint8, device=DEV)

    def d5():
        lg = torch.full((B, 32, 32, 32, 1), float('nan'), device=DEV)
        pr = torch.empty_like(lg)
        st = torch.empty(B, 4, device=DEV)
        L.call('vv_convT3d_final_bce_fwd', L.ptr(xa), L.ptr(w5), L.ptr(tgt), L.ptr(pr), L.ptr(lg), L.ptr(st), B, 16, 64, 0.6, 1e-7, L.VV_BF16,
               L.ptr(ws), ws.numel(), _st())
        torch.cuda.synchronize()
        return lg, st
    runs = [d5(), d5()]
    monkeypatch.setenv('VV_FINAL_BCE', 'box')
    lb, sb = d5()
    monkeypatch.delenv('VV_FINAL_BCE')
    t64 = tgt.double().view(B, -1)

    def sums(logit):                                   # the float64 definition of the four sums on a given set of logits
        l64 = logit.double().view(B, -1)
        q = torch.sigmoid(l64).clamp(1e-7, 1 - 1e-7)
        yh = (l64 >= 0).double()
        return torch.stack([-(0.6 * t64 * q.log() + 0.4 * (1 - t64) * (1 - q).log()).sum(1), (t64 * yh).sum(1), ((1 - t64) * yh).sum(1),
                            (t64 * (1 - yh)).sum(1)], 1)
    ref_box = sums(lb)
    # The sweep form sums the two w terms of an output inside the MFMA (K = 128) and the h / d terms afterwards, the box form all
    # eight in tap order: the logits agree to float32 rounding (1e-5), so an occupancy decision may differ only where the logit is
    # inside that band -- the counts are exact against the sweep form's OWN logits, and within the band population of the box form's.
    band = (lb.double().view(B, -1).abs() <= 1e-5).double().sum(1)
    for lg, st in runs:
        assert (lg - lb).abs().max().item() <= 1e-5
        assert torch.equal(lg, runs[0][0])
        d = (st.double() - sums(lg)).abs().max(0).values
        assert d[0].item() <= 1e-2 and d[1:].max().item() == 0.0, d.tolist()     # loss sum ~2e4 per sample; the counts are exact
        assert bool(((st.double() - ref_box)[:, 1:].abs().sum(1) <= 2 * band).all())
    monkeypatch.setenv('VV_FINAL_BCE', 'sweepp')       # the form that publishes P[halo cell][64 taps] (tap order, as the box form)
    lp, sp_ = d5()
    monkeypatch.delenv('VV_FINAL_BCE')
    assert (lp - lb).abs().max().item() <= 1e-5
    d = (sp_.double() - sums(lp)).abs().max(0).values
    assert d[0].item() <= 1e-2 and d[1:].max().item() == 0.0, d.tolist()


@pytest.mark.skipif(F8 is None, reason='torch.float8_e4m3fn not available')
@pytest.mark.parametrize('B,side,cin,cout,odt', [(2, 8, 128, 256, 'bf16'), (2, 8, 128, 256, 'fp8'), (37, 4, 256, 512, 'fp8'), (1, 16, 128, 64, 'f32'),
                                                 (2, 16, 64, 128, 'fp8'), (3, 8, 64, 128, 'bf16'), (33, 4, 64, 64, 'fp8')])   # Cin 64: tap-pair rows
def test_conv3d_k4s2_fp8(L, B, side, cin, cout, odt):
    """Stride-2 Conv3D on the fp8 MFMA (operands e4m3fn, float32 accumulation, per-channel scales in `scale`) against the
    float64 definition on the same fp8-representable operands; outputs bf16 / fp8 / f32."""
    rng = np.random.default_rng(B + side)
    x = _fp8_round(rng.standard_normal((B, side, side, side, cin)))
    w = _fp8_round(rng.standard_normal((4, 4, 4, cin, cout)))
    scale = (rng.uniform(0.5, 1.5, cout) / np.sqrt(64 * cin)).astype(np.float32)
    shift = rng.normal(0, 0.3, cout).astype(np.float32)
    ref = no.activation(no.conv3d_same(x.astype(np.float64), w.astype(np.float64), 2) * scale + shift, 'elu')
    xd = _dev(x).to(F8)
    wp = torch.empty(cout, 64 * cin, dtype=F8, device=DEV)
    L.call('vv_pack_conv_k4', L.ptr(_dev(w)), L.ptr(wp), cin, cout, L.VV_FP8, _st())
    torch.cuda.synchronize()
    assert np.array_equal(wp.float().cpu().numpy(), w.reshape(64 * cin, cout).T)          # packing fp8-representable values is exact
    so = side // 2
    tout = {'bf16': torch.bfloat16, 'fp8': F8, 'f32': torch.float32}[odt]
    y = torch.zeros(B, so, so, so, cout, dtype=tout, device=DEV)
    ws = torch.empty(max(L.load().vv_conv3d_k4s2_workspace_bytes(B, side, cin, cout, L.VV_FP8), 16), dtype=torch.uint8, device=DEV)
    sd, hd = _dev(scale), _dev(shift)
    L.call('vv_conv3d_k4s2_fwd_io', L.ptr(xd), L.ptr(wp), L.ptr(sd), L.ptr(hd), L.ptr(y), B, side, cin, cout, 1, L.VV_FP8, L.DTYPES[odt],
           L.ptr(ws), ws.numel(), _st())
    torch.cuda.synchronize()
    if odt == 'fp8':
        _check_fp8_out(y, ref, 'conv3d fp8->fp8')
    else:
        _check(y, ref, odt, 'conv3d fp8->' + odt)
    # Cin must be 64 (tap-pair rows) or a multiple of 128
    assert L.load().vv_conv3d_k4s2_fwd_io(L.ptr(xd), L.ptr(wp), None, None, L.ptr(y), B, side, 32, cout, 0, L.VV_FP8, L.DTYPES[odt],
                                          L.ptr(ws), ws.numel(), _st()) == -2


@pytest.mark.skipif(F8 is None, reason='torch.float8_e4m3fn not available')
@pytest.mark.parametrize('B,side,cin,cout,odt', [(2, 4, 256, 128, 'fp8'), (3, 8, 128, 64, 'bf16'), (33, 2, 512, 256, 'fp8')])
def test_convT3d_k4s2_fp8(L, B, side, cin, cout, odt):
    rng = np.random.default_rng(B * 7 + side)
    x = _fp8_round(rng.standard_normal((B, side, side, side, cin)))
    w = _fp8_round(rng.standard_normal((4, 4, 4, cout, cin)))
    scale = (rng.uniform(0.5, 1.5, cout) / np.sqrt(8 * cin)).astype(np.float32)
    shift = rng.normal(0, 0.3, cout).astype(np.float32)
    ref = no.activation(no.conv3d_transpose_same(x.astype(np.float64), w.astype(np.float64), 2) * scale + shift, 'elu')
    xd = _dev(x).to(F8)
    wp = torch.empty(8, cout, 8 * cin, dtype=F8, device=DEV)
    L.call('vv_pack_convT_k4s2', L.ptr(_dev(w)), L.ptr(wp), cin, cout, L.VV_FP8, _st())
    tout = {'bf16': torch.bfloat16, 'fp8': F8}[odt]
    y = torch.zeros(B, 2 * side, 2 * side, 2 * side, cout, dtype=tout, device=DEV)
    ws = torch.empty(max(L.load().vv_convT3d_k4s2_workspace_bytes(B, side, cin, cout, L.VV_FP8), 16), dtype=torch.uint8, device=DEV)
    sd, hd = _dev(scale), _dev(shift)
    L.call('vv_convT3d_k4s2_fwd_io', L.ptr(xd), L.ptr(wp), L.ptr(sd), L.ptr(hd), L.ptr(y), B, side, cin, cout, 1, L.VV_FP8, L.DTYPES[odt],
           L.ptr(ws), ws.numel(), _st())
    torch.cuda.synchronize()
    if odt == 'fp8':
        _check_fp8_out(y, ref, 'convT3d fp8->fp8')
    else:
        _check(y, ref, odt, 'convT3d fp8->bf16')


@pytest.mark.skipif(F8 is None, reason='torch.float8_e4m3fn not available')
def test_dense_and_convert_fp8(L):
    rng = np.random.default_rng(5)
    m, n, k = 37, 128, 4096
    x = _fp8_round(rng.standard_normal((m, k)))
    w = _fp8_round(rng.standard_normal((k, n)))
    scale = (rng.uniform(0.5, 1.5, n) / np.sqrt(k)).astype(np.float32)
    ref = (x.astype(np.float64) @ w.astype(np.float64)) * scale
    wp = torch.empty(n, k, dtype=F8, device=DEV)
    L.call('vv_pack_dense', L.ptr(_dev(w)), L.ptr(wp), k, n, L.VV_FP8, _st())
    y = torch.zeros(m, n, dtype=torch.float32, device=DEV)
    ws = torch.empty(max(L.load().vv_dense_workspace_bytes(m, n, k, L.VV_FP8), 16), dtype=torch.uint8, device=DEV)
    x8, scd = _dev(x).to(F8), _dev(scale)                   # named: raw pointers must outlive the launch
    L.call('vv_dense_fwd', L.ptr(x8), L.ptr(wp), L.ptr(scd), None, L.ptr(y), m, n, k, 0, L.VV_FP8, L.VV_F32, L.ptr(ws),
           ws.numel(), _st())
    torch.cuda.synchronize()
    _check(y, ref, 'f32', 'dense fp8->f32')
    # bf16 operands, fp8 output (the hand-over into an fp8 stretch)
    xb, wb = _bf16_round(rng.standard_normal((m, 256)).astype(np.float32)), _bf16_round(rng.standard_normal((256, n)).astype(np.float32) / 16)
    ref2 = xb.astype(np.float64) @ wb.astype(np.float64)
    wpb = torch.empty(n, 256, dtype=torch.bfloat16, device=DEV)
    L.call('vv_pack_dense', L.ptr(_dev(wb)), L.ptr(wpb), 256, n, L.VV_BF16, _st())
    y8 = torch.zeros(m, n, dtype=F8, device=DEV)
    ws = torch.empty(max(L.load().vv_dense_workspace_bytes(m, n, 256, L.VV_BF16), 16), dtype=torch.uint8, device=DEV)
    xbd = _dev(xb, torch.bfloat16)
    L.call('vv_dense_fwd', L.ptr(xbd), L.ptr(wpb), None, None, L.ptr(y8), m, n, 256, 0, L.VV_BF16, L.VV_FP8, L.ptr(ws),
           ws.numel(), _st())
    torch.cuda.synchronize()
    _check_fp8_out(y8, ref2, 'dense bf16->fp8')
    # conversions: bf16 -> fp8 rounds to nearest (checked against torch's cast), saturates at 448; fp8 -> f32 is exact
    v = _bf16_round((rng.standard_normal(4096) * np.exp(rng.uniform(-6, 6, 4096))).astype(np.float32))
    v[:4] = [1000.0, -1000.0, 448.0, 0.0]
    vd = _dev(v, torch.bfloat16)
    o8 = torch.zeros(4096, dtype=F8, device=DEV)
    L.call('vv_convert', L.ptr(vd), L.ptr(o8), 4096, L.VV_BF16, L.VV_FP8, _st())
    back = torch.zeros(4096, dtype=torch.float32, device=DEV)
    L.call('vv_convert', L.ptr(o8), L.ptr(back), 4096, L.VV_FP8, L.VV_F32, _st())
    torch.cuda.synchronize()
    want = torch.from_numpy(np.clip(v, -448, 448)).to(F8).float().numpy()
    assert np.array_equal(back.cpu().numpy(), want)
    assert np.array_equal(o8.float().cpu().numpy(), want)


@pytest.mark.skipif(F8 is None, reason='torch.float8_e4m3fn not available')
def test_conv3d_direct_fp8_output(L):
    """The direct 64 -> 128 kernel storing e4m3fn instead of bf16 (hand-over into the fp8 layers): equals its own bf16
    output rounded once more, up to double rounding (bf16 first, then fp8: at most one fp8 ulp apart)."""
    rng = np.random.default_rng(3)
    B, side, cin, cout = 2, 16, 64, 128
    x = _bf16_round(rng.standard_normal((B, side, side, side, cin)).astype(np.float32))
    w = _bf16_round((rng.standard_normal((4, 4, 4, cin, cout)) / np.sqrt(64 * cin)).astype(np.float32))
    scale, shift = rng.uniform(0.5, 1.5, cout).astype(np.float32), rng.normal(0, 0.3, cout).astype(np.float32)
    ref = no.activation(no.conv3d_same(x.astype(np.float64), w.astype(np.float64), 2) * scale + shift, 'elu')
    xd, sd, hd = _dev(x, torch.bfloat16), _dev(scale), _dev(shift)
    wp = torch.empty(cout, 64 * cin, dtype=torch.bfloat16, device=DEV)
    L.call('vv_pack_conv_k4', L.ptr(_dev(w)), L.ptr(wp), cin, cout, L.VV_BF16, _st())
    y8 = torch.zeros(B, 8, 8, 8, cout, dtype=F8, device=DEV)
    L.call('vv_conv3d_k4s2_direct_fwd_io', L.ptr(xd), L.ptr(wp), L.ptr(sd), L.ptr(hd), L.ptr(y8), B, side, cin, cout, 1, L.VV_BF16, L.VV_FP8, _st())
    torch.cuda.synchronize()
    _check_fp8_out(y8, ref, 'conv3d direct bf16->fp8')


@pytest.mark.skipif(F8 is None, reason='torch.float8_e4m3fn not available')
@pytest.mark.parametrize('B,D', [(2, 32), (1, 64)])
def test_conv3d_first_fp8_output(L, B, D):
    """First layer (plane-form kernel) storing e4m3fn for an fp8 second layer; smaller grids have no fp8 store."""
    from voxvae import synthetic as syn
    rng = np.random.default_rng(D)
    x = syn.make_voxels(B, D, seed=D)
    w = _bf16_round((rng.standard_normal((4, 4, 4, 1, 64)) / 8).astype(np.float32))
    scale, shift = rng.uniform(0.5, 1.5, 64).astype(np.float32), rng.normal(0, 0.3, 64).astype(np.float32)
    ref = no.activation(no.conv3d_same(x.astype(np.float64), w.astype(np.float64), 2) * scale + shift, 'elu')
    wp = torch.empty(64, 64, dtype=torch.bfloat16, device=DEV)
    L.call('vv_pack_conv_k4', L.ptr(_dev(w)), L.ptr(wp), 1, 64, L.VV_BF16, _st())
    xd, sd, hd = _dev(x), _dev(scale), _dev(shift)
    y = torch.zeros(B, D // 2, D // 2, D // 2, 64, dtype=F8, device=DEV)
    L.call('vv_conv3d_first_fwd_io', L.ptr(xd), L.ptr(wp), L.ptr(sd), L.ptr(hd), L.ptr(y), B, D, 64, 1, L.VV_BF16, L.VV_FP8, _st())
    torch.cuda.synchronize()
    _check_fp8_out(y, ref, 'conv3d_first bf16->fp8')
    assert L.load().vv_conv3d_first_fwd_io(L.ptr(xd), L.ptr(wp), L.ptr(sd), L.ptr(hd), L.ptr(y), B, 16, 64, 1, L.VV_BF16, L.VV_FP8, _st()) == -3   # VV_ERR_DTYPE


@pytest.mark.parametrize('out', ['bf16', 'fp8'])
@pytest.mark.parametrize('B,D,act', [(65, 32, 1), (70, 32, 1), (130, 32, 0), (200, 32, 2), (255, 32, 1), (257, 32, 3), (9, 64, 1), (20, 64, 0), (40, 64, 1), (67, 64, 1)])
def test_conv3d_first_chained_equals_plane_form(L, monkeypatch, out, B, D, act):
    """The chained first layer (an output plane takes two of its four input planes from its predecessor in the workgroup; ring of
    half tiles, buffer-descriptor loads, register-to-memory e4m3fn stores) against the plane form that loads all four planes per
    item, bit for bit: chains of 2, 3 (they cross plane-0 boundaries: an item in the middle of a chain without a predecessor),
    4, 5 and 9 (ragged last workgroups: batches 65, 255, 257, 67), both grid sizes, both output types, every activation template."""
    if out == 'fp8' and F8 is None:
        pytest.skip('torch.float8_e4m3fn not available')
    g = torch.Generator(device=DEV).manual_seed(B + D)
    x = (torch.rand(B, D, D, D, 1, device=DEV, generator=g) < 0.15).float().contiguous()
    w = (torch.randn(4, 4, 4, 1, 64, device=DEV, generator=g) / 8).contiguous()
    sc = torch.rand(64, device=DEV, generator=g) + 0.5
    sh = torch.randn(64, device=DEV, generator=g) * 0.3
    wp = torch.empty(64, 64, dtype=torch.bfloat16, device=DEV)
    L.call('vv_pack_conv_k4', L.ptr(w), L.ptr(wp), 1, 64, L.VV_BF16, _st())
    odt, tdt = (L.VV_FP8, torch.uint8) if out == 'fp8' else (L.VV_BF16, torch.int16)

    def run():
        y = torch.full((B, D // 2, D // 2, D // 2, 64), 0x55, dtype=tdt, device=DEV)
        L.call('vv_conv3d_first_fwd_io', L.ptr(x), L.ptr(wp), L.ptr(sc), L.ptr(sh), L.ptr(y), B, D, 64, act, L.VV_BF16, odt, _st())
        torch.cuda.synchronize()
        return y
    monkeypatch.delenv('VV_FIRSTCONV_NOCHAIN', raising=False)
    a = run()
    monkeypatch.setenv('VV_FIRSTCONV_NOCHAIN', '1')
    b = run()
    monkeypatch.delenv('VV_FIRSTCONV_NOCHAIN')
    assert torch.equal(a, b)


@pytest.mark.skipif(F8 is None, reason='torch.float8_e4m3fn not available')
@pytest.mark.parametrize('B,side,act', [(1, 8, 1), (3, 8, 0), (2, 16, 1), (1, 32, 2)])
def test_convT3d_direct_fp8(L, B, side, act):
    """fp8 twin of the direct 128 -> 64 transposed layer (LDS-resident halo tile, K = 64 block-scaled MFMA, bf16 output)
    against the float64 definition on the same fp8-representable operands, and against the fp8 implicit GEMM."""
    cin, cout = 128, 64
    lib = L.load()
    assert lib.vv_convT3d_k4s2_direct_fp8_supported(side, cin, cout) == 1 and lib.vv_convT3d_k4s2_direct_fp8_supported(4, cin, cout) == 0
    rng = np.random.default_rng(side + B)
    x = _fp8_round(rng.standard_normal((B, side, side, side, cin)))
    w = _fp8_round(rng.standard_normal((4, 4, 4, cout, cin)))
    scale = (rng.uniform(0.5, 1.5, cout) / np.sqrt(8 * cin)).astype(np.float32)
    shift = rng.normal(0, 0.3, cout).astype(np.float32)
    actname = {0: 'none', 1: 'elu', 2: 'relu'}[act]
    ref = no.activation(no.conv3d_transpose_same(x.astype(np.float64), w.astype(np.float64), 2) * scale + shift, actname)
    xd, wd, sd, hd = _dev(x).to(F8), _dev(w), _dev(scale), _dev(shift)
    wf = torch.empty(64 * cin * cout, dtype=torch.uint8, device=DEV)
    L.call('vv_pack_convT_k4s2_frag_fp8', L.ptr(wd), L.ptr(wf), cin, cout, _st())
    y = torch.full((B, 2 * side, 2 * side, 2 * side, cout), -7.0, dtype=torch.bfloat16, device=DEV)
    L.call('vv_convT3d_k4s2_direct_fp8_fwd', L.ptr(xd), L.ptr(wf), L.ptr(sd), L.ptr(hd), L.ptr(y), B, side, cin, cout, act, L.VV_BF16, _st())
    y8 = torch.zeros(B, 2 * side, 2 * side, 2 * side, cout, dtype=F8, device=DEV)
    L.call('vv_convT3d_k4s2_direct_fp8_fwd', L.ptr(xd), L.ptr(wf), L.ptr(sd), L.ptr(hd), L.ptr(y8), B, side, cin, cout, act, L.VV_FP8, _st())
    torch.cuda.synchronize()
    _check_fp8_out(y8, ref, 'convT3d_direct_fp8 -> fp8')
    torch.cuda.synchronize()
    _check(y, ref, 'bf16', 'convT3d_direct_fp8')
    wp = torch.empty(8, cout, 8 * cin, dtype=F8, device=DEV)
    L.call('vv_pack_convT_k4s2', L.ptr(wd), L.ptr(wp), cin, cout, L.VV_FP8, _st())
    ws = torch.empty(max(lib.vv_convT3d_k4s2_workspace_bytes(B, side, cin, cout, L.VV_FP8), 16), dtype=torch.uint8, device=DEV)
    y2 = torch.empty_like(y)
    L.call('vv_convT3d_k4s2_fwd_io', L.ptr(xd), L.ptr(wp), L.ptr(sd), L.ptr(hd), L.ptr(y2), B, side, cin, cout, act, L.VV_FP8, L.VV_BF16,
           L.ptr(ws), ws.numel(), _st())
    torch.cuda.synchronize()
    d = (y.float() - y2.float()).abs().max().item()
    assert d <= 2e-2 * max(1.0, float(np.abs(ref).max())), 'direct vs implicit GEMM (fp8): %.3e' % d


@pytest.mark.skipif(F8 is None, reason='torch.float8_e4m3fn not available')
@pytest.mark.parametrize('B,side,act,odt', [(1, 16, 1, 'bf16'), (3, 16, 0, 'fp8'), (2, 32, 1, 'fp8'), (1, 64, 2, 'bf16')])
def test_conv3d_direct_fp8(L, B, side, act, odt):
    """fp8 twin of the direct 64 -> 128 layer (phase tiles with 64-byte rows, tap pairs per chunk, K = 64 block-scaled MFMA)
    against the float64 definition on fp8-representable operands, and against the fp8 implicit GEMM (tap-pair rows)."""
    cin, cout = 64, 128
    lib = L.load()
    assert lib.vv_conv3d_k4s2_direct_fp8_supported(side, cin, cout) == 1 and lib.vv_conv3d_k4s2_direct_fp8_supported(8, cin, cout) == 0
    rng = np.random.default_rng(side + 3 * B)
    x = _fp8_round(rng.standard_normal((B, side, side, side, cin)))
    w = _fp8_round(rng.standard_normal((4, 4, 4, cin, cout)))
    scale = (rng.uniform(0.5, 1.5, cout) / np.sqrt(64 * cin)).astype(np.float32)
    shift = rng.normal(0, 0.3, cout).astype(np.float32)
    actname = {0: 'none', 1: 'elu', 2: 'relu'}[act]
    ref = no.activation(no.conv3d_same(x.astype(np.float64), w.astype(np.float64), 2) * scale + shift, actname)
    xd, sd, hd = _dev(x).to(F8), _dev(scale), _dev(shift)
    wp = torch.empty(cout, 64 * cin, dtype=F8, device=DEV)
    L.call('vv_pack_conv_k4', L.ptr(_dev(w)), L.ptr(wp), cin, cout, L.VV_FP8, _st())
    so = side // 2
    tout = {'bf16': torch.bfloat16, 'fp8': F8}[odt]
    y = torch.zeros(B, so, so, so, cout, dtype=tout, device=DEV)
    L.call('vv_conv3d_k4s2_direct_fp8_fwd', L.ptr(xd), L.ptr(wp), L.ptr(sd), L.ptr(hd), L.ptr(y), B, side, cin, cout, act, L.DTYPES[odt], _st())
    torch.cuda.synchronize()
    if odt == 'fp8':
        _check_fp8_out(y, ref, 'conv3d_direct_fp8 -> fp8')
    else:
        _check(y, ref, 'bf16', 'conv3d_direct_fp8 -> bf16')
    ws = torch.empty(max(lib.vv_conv3d_k4s2_workspace_bytes(B, side, cin, cout, L.VV_FP8), 16), dtype=torch.uint8, device=DEV)
    y2 = torch.zeros(B, so, so, so, cout, dtype=torch.bfloat16, device=DEV)
    L.call('vv_conv3d_k4s2_fwd_io', L.ptr(xd), L.ptr(wp), L.ptr(sd), L.ptr(hd), L.ptr(y2), B, side, cin, cout, act, L.VV_FP8, L.VV_BF16,
           L.ptr(ws), ws.numel(), _st())
    torch.cuda.synchronize()
    d = (y.float() - y2.float()).abs().max().item()
    assert d <= 0.07 * max(1.0, float(np.abs(ref).max())), 'direct vs implicit GEMM (fp8): %.3e' % d


@pytest.mark.skipif(F8 is None, reason='torch.float8_e4m3fn not available')
@pytest.mark.parametrize('B,side', [(2, 16), (3, 8), (1, 32)])
def test_convT3d_final_bce_fp8_input(L, B, side):
    """Last layer (sweep form) with an e4m3fn activation: the kernel quantises the float32 Keras kernel per tap itself, so the
    float64 reference uses the same per-tap quantised weights; logits, probabilities, loss and TP/FP/FN as in the bf16 test."""
    rng = np.random.default_rng(side + B)
    x = _fp8_round(rng.standard_normal((B, side, side, side, 64)))
    w = (rng.standard_normal((4, 4, 4, 1, 64)) * 0.3 * np.exp(rng.uniform(-2, 2, (4, 4, 4, 1, 1)))).astype(np.float32)   # taps of different magnitude
    s_tap = (np.abs(w).max(axis=-1, keepdims=True) / np.float32(256.0)).astype(np.float32)
    wq = _fp8_round((w * (np.float32(1.0) / s_tap)).astype(np.float32)).astype(np.float64) * s_tap       # the kernel's arithmetic: w * (1 / s)
    D = 2 * side
    y = (rng.random((B, D, D, D, 1)) < 0.3).astype(np.float32)
    lg = no.conv3d_transpose_same(x.astype(np.float64), wq, 2)
    pr = no.sigmoid(lg)
    tp, fp, fn = no.voxel_precision_recall(y, pr.astype(np.float32))
    ws = torch.empty(max(L.load().vv_convT3d_final_bce_workspace_bytes(B, side), 16), dtype=torch.uint8, device=DEV)
    probs = torch.empty(B, D, D, D, 1, dtype=torch.float32, device=DEV)
    logits = torch.empty_like(probs)
    stats = torch.empty(B, 4, dtype=torch.float32, device=DEV)
    xd, wd, yd = _dev(x).to(F8), _dev(w), _dev(y)            # named: the raw pointers must outlive the launch
    L.call('vv_convT3d_final_bce_fwd', L.ptr(xd), L.ptr(wd), L.ptr(yd), L.ptr(probs), L.ptr(logits), L.ptr(stats), B, side, 64,
           0.6, 1e-7, L.VV_FP8, L.ptr(ws), ws.numel(), _st())
    torch.cuda.synchronize()
    glg = logits.cpu().numpy().astype(np.float64)
    assert np.abs(glg - lg).max() < 3e-5 * max(1.0, np.abs(lg).max())         # exact fp8 x fp8 products, float32 accumulation
    np.testing.assert_allclose(probs.cpu().numpy(), pr, rtol=0, atol=0.25 * 3e-5 * max(1.0, np.abs(lg).max()) + 1e-6)   # dp <= 0.25 * dlogit
    s = stats.cpu().numpy().astype(np.float64)
    bce = no.binary_loss(pr.astype(np.float32), y, gamma=0.6)
    np.testing.assert_allclose(s[:, 0], bce, rtol=2e-3)
    flips = int(((glg >= 0) != (lg >= 0)).sum())
    assert flips == 0 or np.abs(lg[(glg >= 0) != (lg >= 0)]).max() < 1e-5
    for k, r in ((1, tp), (2, fp), (3, fn)):
        assert np.abs(s[:, k] - r).max() <= flips
    assert L.load().vv_convT3d_final_bce_fwd(L.ptr(xd), L.ptr(wd), L.ptr(yd), L.ptr(probs), L.ptr(logits), L.ptr(stats), B, 4, 64,
                                             0.6, 1e-7, L.VV_FP8, L.ptr(ws), ws.numel(), _st()) == -2

"""CPU tests of the oracle itself: three independent statements must agree
(numpy definition / C fp32 restatement / torch-CPU functional ops), plus the committed golden
fixtures and the loss edge cases of SURVEY.md §8(c).  No GPU."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import c_oracle as co
from oracle import numpy_oracle as no
from voxvae import synthetic as syn

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def _torch_conv_same(x, w, s):
    # Keras [kd,kh,kw,Ci,Co] -> torch [Co,Ci,kd,kh,kw]; TF SAME for k=4: s=2 pads (1,1), s=1 pads (1,2)
    xt = torch.from_numpy(x).permute(0, 4, 1, 2, 3)
    wt = torch.from_numpy(w).permute(4, 3, 0, 1, 2)
    pad = (1, 1) * 3 if s == 2 else (1, 2) * 3
    return F.conv3d(F.pad(xt, pad), wt, stride=s).permute(0, 2, 3, 4, 1).numpy()


def _torch_convT_same(x, w, s):
    # Keras [kd,kh,kw,Co,Ci] -> torch conv_transpose weight [Ci,Co,kd,kh,kw]; padding=1, crop to n*s
    xt = torch.from_numpy(x).permute(0, 4, 1, 2, 3)
    wt = torch.from_numpy(w).permute(4, 3, 0, 1, 2)
    y = F.conv_transpose3d(xt, wt, stride=s, padding=1)
    n = x.shape[1] * s
    return y[:, :, :n, :n, :n].permute(0, 2, 3, 4, 1).numpy()


@pytest.mark.parametrize('D,Ci,Co,s', [(8, 3, 5, 2), (4, 4, 6, 1), (2, 8, 4, 1), (6, 1, 7, 2), (7, 2, 3, 2), (5, 2, 3, 1)])
def test_conv3d_same_three_statements(D, Ci, Co, s):
    rng = np.random.default_rng(D * 100 + Ci)
    x = rng.standard_normal((2, D, D, D, Ci))
    w = rng.standard_normal((4, 4, 4, Ci, Co))
    y_np = no.conv3d_same(x, w, s)
    assert y_np.shape[1] == -(-D // s)
    if D % s == 0:                                  # torch statement uses the fixed pads of even sizes
        np.testing.assert_allclose(y_np, _torch_conv_same(x, w, s), rtol=0, atol=1e-11)
    y_c = co.conv3d_same(x, w, s)
    np.testing.assert_allclose(y_c, y_np, rtol=0, atol=2e-4)


@pytest.mark.parametrize('D,Ci,Co,s', [(4, 3, 5, 2), (2, 8, 6, 1), (4, 8, 4, 1), (3, 2, 1, 2), (8, 4, 1, 2)])
def test_conv3d_transpose_same_three_statements(D, Ci, Co, s):
    rng = np.random.default_rng(D * 10 + Co)
    x = rng.standard_normal((2, D, D, D, Ci))
    w = rng.standard_normal((4, 4, 4, Co, Ci))
    y_np = no.conv3d_transpose_same(x, w, s)
    assert y_np.shape == (2, D * s, D * s, D * s, Co)
    np.testing.assert_allclose(y_np, _torch_convT_same(x, w, s), rtol=0, atol=1e-11)
    np.testing.assert_allclose(co.conv3d_transpose_same(x, w, s), y_np, rtol=0, atol=2e-4)


@pytest.mark.parametrize('D,Ci,Co,s', [(8, 2, 3, 2), (6, 1, 2, 2), (4, 3, 2, 1), (7, 2, 2, 2)])
def test_conv_and_transpose_against_scipy_correlate(D, Ci, Co, s):
    """A fourth, independent statement (round 4): scipy.signal.correlate / convolve, an N-D routine that shares no code with numpy_oracle's
    tap loops, the C restatement or torch's oneDNN path.  Conv3D SAME = 'valid' cross-correlation of the TF-padded input, strided by
    slicing; Conv3DTranspose SAME = full convolution of the zero-stuffed input with the kernel, cropped by pad_before (the gradient-of-conv
    definition TensorFlow documents)."""
    from scipy import signal
    rng = np.random.default_rng(D * 7 + Ci)
    x = rng.standard_normal((2, D, D, D, Ci))
    w = rng.standard_normal((4, 4, 4, Ci, Co))
    out_n = -(-D // s)
    pad_total = max((out_n - 1) * s + 4 - D, 0)
    pb = pad_total // 2
    xp = np.pad(x, ((0, 0),) + ((pb, pad_total - pb),) * 3 + ((0, 0),))
    y = np.zeros((2, out_n, out_n, out_n, Co))
    for b in range(2):
        for co in range(Co):
            acc = sum(signal.correlate(xp[b, ..., ci], w[..., ci, co], mode='valid') for ci in range(Ci))
            y[b, ..., co] = acc[::s, ::s, ::s][:out_n, :out_n, :out_n]
    np.testing.assert_allclose(no.conv3d_same(x, w, s), y, rtol=0, atol=1e-10)
    # transposed: Keras kernel [k,k,k,Cout_T,Cin_T]; input z [2,n,n,n,Cin_T] -> [2,n*s,n*s,n*s,Cout_T]
    n = out_n
    z = rng.standard_normal((2, n, n, n, Co))
    wt = rng.standard_normal((4, 4, 4, Ci, Co))                     # Cout_T = Ci, Cin_T = Co
    N = n * s
    pt = max((n - 1) * s + 4 - N, 0)
    ptb = pt // 2
    yt = np.zeros((2, N, N, N, Ci))
    for b in range(2):
        for co in range(Ci):
            acc = 0.0
            for ci in range(Co):
                up = np.zeros(((n - 1) * s + 1,) * 3)
                up[::s, ::s, ::s] = z[b, ..., ci]
                acc = acc + signal.convolve(up, wt[..., co, ci], mode='full')
            yt[b, ..., co] = acc[ptb:ptb + N, ptb:ptb + N, ptb:ptb + N]
    np.testing.assert_allclose(no.conv3d_transpose_same(z, wt, s), yt, rtol=0, atol=1e-10)


def test_transpose_is_adjoint_of_conv():
    """<conv(x), y> == <x, convT(y)> for the same kernel: the defining property of Conv3DTranspose."""
    rng = np.random.default_rng(3)
    for s, D in ((2, 8), (1, 4)):
        x = rng.standard_normal((1, D, D, D, 3))
        w = rng.standard_normal((4, 4, 4, 3, 5))         # forward kernel Ci=3 -> Co=5
        y = rng.standard_normal((1, D // s, D // s, D // s, 5))
        lhs = np.sum(no.conv3d_same(x, w, s) * y)
        rhs = np.sum(x * no.conv3d_transpose_same(y, w, s))   # same array read as [k,k,k,Co_T=3,Ci_T=5]
        assert abs(lhs - rhs) < 1e-9 * max(1.0, abs(lhs))


def test_bn_act_and_dense_c_vs_numpy():
    rng = np.random.default_rng(0)
    x = rng.standard_normal((3, 2, 2, 2, 16)).astype(np.float32)
    p = {'bn/gamma': rng.uniform(0.5, 1.5, 16).astype(np.float32), 'bn/beta': rng.normal(0, 0.1, 16).astype(np.float32),
         'bn/moving_mean': rng.normal(0, 0.1, 16).astype(np.float32),
         'bn/moving_variance': rng.uniform(0.5, 1.5, 16).astype(np.float32)}
    for act in ('elu', 'relu', 'lrelu', 'None'):
        ref = no.activation(no.batchnorm_inference(x.astype(np.float64), p['bn/gamma'], p['bn/beta'], p['bn/moving_mean'],
                                                   p['bn/moving_variance']), act)
        got = co.bn_act_(x.copy(), p, 'bn', act)
        np.testing.assert_allclose(got, ref, rtol=0, atol=2e-6)


@pytest.mark.parametrize('name', ['vae_d32_l64_b2', 'ae_d32_l64_b2', 'vae_d16_l64_b3'])
def test_c_oracle_matches_golden(name):
    g = np.load(os.path.join(GOLDEN, name + '.npz'))
    D, L, var, B, C = [int(v) for v in g['meta'][:5]]
    cfg = syn.make_config(D, L, bool(var))
    ep, dp = syn.make_encoder_params(cfg['encoder']), syn.make_decoder_params(cfg['decoder'])
    x, eps = syn.make_voxels(B, D), syn.make_eps(B, L)
    c = co.vae_eval_forward(cfg, ep, dp, x, x, eps, variational=bool(var))
    np.testing.assert_allclose(c['enc_out'], g['p0_enc_out'], rtol=0, atol=2e-6)
    np.testing.assert_allclose(c['z'], g['p0_z'], rtol=0, atol=5e-6)
    if var:
        np.testing.assert_allclose(c['kl'], g['p0_kl'], rtol=1e-5, atol=1e-6)
    lg = g['p0_logits']
    np.testing.assert_allclose(c['logits'], lg, rtol=0, atol=1e-4)       # well inside the 1e-3 bar
    safe = np.abs(lg) > 1e-4                                              # occupancy exact away from the threshold
    assert np.array_equal((c['logits'] >= 0)[safe], (lg >= 0)[safe])
    np.testing.assert_allclose(c['bce'], g['p0_bce'], rtol=2e-6)
    for k in ('tp', 'fp', 'fn'):
        assert np.abs(c[k] - g['p0_' + k]).max() <= np.sum(~safe)


def test_numpy_oracle_reproduces_golden_fp64():
    """The fixture generator is deterministic (seeded numpy Generators)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location('make_golden', os.path.join(GOLDEN, 'make_golden.py'))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    new = mg.make_case('vae_d16_l64_b3')
    g = np.load(os.path.join(GOLDEN, 'vae_d16_l64_b3.npz'))
    for k in g.files:
        np.testing.assert_array_equal(new[k], g[k])


def test_binary_loss_edge_cases():
    """SURVEY.md hard part 6: in float32 the upper clip 1-1e-7 is 0.99999988, i.e. logits saturate at
    +-15.9424; all-empty / all-full grids; TP+FP == 0 -> precision 0 through the 1e-10 guard."""
    assert np.float32(1.0) - np.float32(1e-7) == np.float32(0.99999988)
    lg = np.array([[-40.0, -16.0, -15.9, 0.0, 15.9, 16.0, 40.0, -1e-9]], np.float32)
    y = np.array([[1, 0, 1, 1, 0, 1, 0, 1]], np.float32)
    p32 = no.sigmoid(lg.astype(np.float32))
    ref = no.binary_loss(p32, y, gamma=0.6)
    probs, bce, tp, fp, fn = co.sigmoid_bce_counts(lg, y)
    np.testing.assert_allclose(bce, ref, rtol=1e-6)
    # saturated positives cost -0.6*log(1e-7), saturated negatives -0.4*log(1-0.99999988)
    assert abs(float(no.binary_loss(np.array([[0.0]], np.float32), np.array([[1.0]], np.float32), gamma=0.6)[0])
               - 0.6 * 16.118095) < 1e-4
    t_tp, t_fp, t_fn = no.voxel_precision_recall(y, p32)
    assert (tp[0], fp[0], fn[0]) == (t_tp[0], t_fp[0], t_fn[0])
    # threshold is >= on the float32 probability: sigmoid(-1e-9) rounds to 0.5 -> occupied
    assert probs[0, -1] == np.float32(0.5) and t_tp[0] == 3
    empty = np.zeros((1, 64), np.float32)
    tp0, fp0, fn0 = no.voxel_precision_recall(empty, np.full((1, 64), 0.1, np.float32))
    pr, rc = no.pr_rc(tp0, fp0, fn0)
    assert pr == 0.0 and rc == 0.0
    full = np.ones((1, 64), np.float32)
    tp1, fp1, fn1 = no.voxel_precision_recall(full, np.full((1, 64), 0.9, np.float32))
    assert no.pr_rc(tp1, fp1, fn1) == (pytest.approx(1.0), pytest.approx(1.0))


def test_kl_and_sampling_formulas():
    rng = np.random.default_rng(1)
    e = rng.standard_normal((4, 32)).astype(np.float32) * 6
    eps = rng.standard_normal((4, 16)).astype(np.float32)
    mu, lv = no.split_mean_logvar(e.astype(np.float64), 16)
    assert lv.min() >= -10 and lv.max() <= 10
    z, kl = co.reparam_kl(e, eps, 16)
    np.testing.assert_allclose(z, no.sampling(mu, lv, eps), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(kl, no.kl_loss(mu, lv, 0 * mu, 0 * lv), rtol=2e-5)
    # closed form against torch.distributions
    q = torch.distributions.Normal(torch.from_numpy(mu), torch.from_numpy(np.exp(0.5 * lv)))
    pr = torch.distributions.Normal(torch.zeros_like(q.loc), torch.ones_like(q.scale))
    np.testing.assert_allclose(torch.distributions.kl_divergence(q, pr).sum(-1).numpy(),
                               no.kl_loss(mu, lv, 0 * mu, 0 * lv), rtol=1e-10)


def test_missing_latent_quirks():
    """where(z == 0) also rewrites genuine zeros (nolbo.py:1481-1482); masked entries are replaced by
    the prior sample in the corrected latent (nolbo.py:1510)."""
    g = np.load(os.path.join(GOLDEN, 'vae_d16_l64_b3.npz'))
    mask = syn.make_mask(3, 64, 0.5)
    cats = syn.make_category_vectors(40, 64)
    z = g['p5_z']
    np.testing.assert_allclose(z[mask == 0], np.broadcast_to(cats.astype(np.float64).mean(0), z.shape)[mask == 0])
    zc = g['p5_z_corr']
    np.testing.assert_array_equal(zc[mask == 1], z[mask == 1])
    eps2 = syn.make_eps(3, 64, seed=8)
    idx = g['p5_argmin_masked'].astype(int)
    np.testing.assert_allclose(zc[mask == 0], (cats[idx].astype(np.float64) + eps2)[mask == 0])


def test_torch_cpu_f32_bracket_matches_c_oracle():
    """bench.py's secondary CPU leg (torch float32 ops) and the C oracle are two statements of the same forward."""
    from oracle import c_oracle as co
    from oracle import torch_oracle as to
    cfg = syn.make_config(16, 64, True)
    ep, dp = syn.make_encoder_params(cfg['encoder']), syn.make_decoder_params(cfg['decoder'])
    x, eps = syn.make_voxels(3, 16), syn.make_eps(3, 64)
    co.build()
    r = co.vae_eval_forward(cfg, ep, dp, x, x, eps)
    lg, bce, tp, fp, fn = to.eval_forward_f32(cfg, ep, dp, x, eps)
    assert np.abs(lg - r['logits']).max() < 2e-4
    np.testing.assert_allclose(bce, r['bce'], rtol=1e-4)
    np.testing.assert_array_equal([tp, fp, fn], [r['tp'], r['fp'], r['fn']])


def test_c_oracle_under_asan():
    """The C restatement under AddressSanitizer + UBSan (oracle/Makefile `asan`: every exported routine on small odd shapes
    with exactly-sized heap buffers).  GPU ASan does not exist on this pool; the sanitizers run on the CPU build."""
    import subprocess
    here = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'oracle')
    p = subprocess.run(['make', '-C', here, '-B', 'asan'], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert p.returncode == 0 and 'asan driver: ok' in p.stdout, p.stdout[-3000:]


@pytest.mark.parametrize('name', ['vae_d32_l64_b2', 'vae_d16_l64_b3'])
def test_c_oracle_two_pass_get_eval_matches_golden(name):
    """c_oracle.vae_get_eval (fp32 C conv stacks + float32 latent algebra; the checker of the trained-weights parity test)
    against the float64 definition-level fixtures of the missing-latent path (nolbo.py:1472-1528)."""
    g = np.load(os.path.join(GOLDEN, name + '.npz'))
    D, L, var, B, C = [int(v) for v in g['meta'][:5]]
    cfg = syn.make_config(D, L, True)
    ep, dp = syn.make_encoder_params(cfg['encoder']), syn.make_decoder_params(cfg['decoder'])
    x, eps, eps2 = syn.make_voxels(B, D), syn.make_eps(B, L), syn.make_eps(B, L, seed=8)
    oh, cats, mask = syn.make_onehot(B, C), syn.make_category_vectors(C, L), syn.make_mask(B, L, 0.5)
    r = co.vae_get_eval(cfg, ep, dp, x, x, oh, cats, eps, 0.5, mask, eps2)
    np.testing.assert_allclose(r['z'], g['p5_z'], rtol=0, atol=5e-6)
    np.testing.assert_allclose(r['z_c'], g['p5_z_corr'], rtol=0, atol=5e-6)
    assert np.array_equal(r['argmin_masked'], g['p5_argmin_masked'].astype(np.int64))
    np.testing.assert_allclose(r['logits'], g['p5_logits'], rtol=0, atol=1e-4)
    np.testing.assert_allclose(r['logits_c'], g['p5_logits_c'], rtol=0, atol=1e-4)
    np.testing.assert_allclose(r['bce'], g['p5_bce'], rtol=2e-6)
    np.testing.assert_allclose(r['bce_c'], g['p5_bce_c'], rtol=2e-6)
    assert abs(r['acc'] - g['p5_scalars'][3]) < 1e-12 and abs(r['acc_c'] - g['p5_scalars'][7]) < 1e-12


def test_max_pool_encoder_c_vs_numpy():
    """final_pool = 'max' (reference autoencoder3D.py:92-93): the fp32 C composition against the float64 definition."""
    cfg = syn.make_config(32, 8, True)                   # last feature map 2^3: the two pools differ
    cfg['encoder']['final_pool'] = 'max'
    cfg['encoder']['filter_num_list'] = [8, 8, 16, 16, 16]
    ep = syn.make_encoder_params(cfg['encoder'])
    x = syn.make_voxels(2, 32, seed=5)
    ref = no.encoder3D_forward(cfg['encoder'], ep, x.astype(np.float64))
    got = co.encoder3D_forward(cfg['encoder'], ep, x)
    np.testing.assert_allclose(got, ref, rtol=0, atol=5e-6)
    cfg['encoder']['final_pool'] = 'average'
    assert np.abs(no.encoder3D_forward(cfg['encoder'], ep, x.astype(np.float64)) - ref).max() > 1e-3     # the two pools differ

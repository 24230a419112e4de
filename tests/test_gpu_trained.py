"""End-to-end parity at a TRAINED operating point.

Every other end-to-end check runs on seeded Glorot weights: a near-random predictor (IoU ~ 0.1, |logit| < 6), where an IoU
delta is weak evidence and the saturation region of the reference's `clip(sigmoid(l), 1e-7, 1 - 1e-7)` (function.py:79) is
never reached.  Here the 32^3 VAE is first fitted with the repo's own float32 `fit` (voxvae/trained.py, reference
nolbo.py:1411-1447) until the CPU oracle itself reports IoU >= 0.5 and |logit| >= 16 on the evaluation batch; the trained
weights then go to the oracle and to the HIP path in every arithmetic mode:

  f32          logits within 1e-3 of the oracle, occupancy identical outside a 1e-4 band around the threshold (north_star)
  bf16 / fp8   mean IoU within 1e-3, per-sample IoU within 5e-3
  BCE          per-sample sums within tolerance although most voxels sit in the saturated region
  missing_prob = 0.9: the two-pass getEval (nolbo.py:1504-1528) against the composed oracle
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
N_EVAL = 256


@pytest.fixture(scope='module')
def trained():
    from voxvae import synthetic as syn
    from voxvae import trained as tr
    from oracle import c_oracle as co
    cfg, ep, dp, info = tr.train_operating_point(device=DEV, verbose=True)
    assert info['reached'], info
    # evaluation batch: 192 shapes of the training pool + 64 the model has never seen (256: the mean IoU delta of a
    # 64-sample batch carries ~3e-4 of sampling noise, a third of the bar)
    x = np.concatenate([syn.make_voxels(256, 32, seed=4321)[:192], syn.make_voxels(64, 32, seed=777)], axis=0)
    eps = syn.make_eps(N_EVAL, 64, seed=70)
    ref = co.vae_eval_forward(cfg, ep, dp, x, x, eps)
    iou = ref['tp'] / np.maximum(ref['tp'] + ref['fp'] + ref['fn'], 1)
    print('\n[trained] %d fit steps; oracle IoU %.4f (seen %.4f / unseen %.4f), logits in [%.1f, %.1f], %.1f %% of voxels with |logit| > 15.94'
          % (info['steps'], iou.mean(), iou[:192].mean(), iou[192:].mean(), ref['logits'].min(), ref['logits'].max(),
             100.0 * np.mean(np.abs(ref['logits']) > 15.94)))
    # the operating point the test is about, judged by the ORACLE, not by the path under test
    assert iou.mean() >= 0.5
    assert np.abs(ref['logits']).max() >= 16.0
    return dict(cfg=cfg, ep=ep, dp=dp, x=x, eps=eps, ref=ref, iou=iou, info=info)


def _model(t, dtype):
    import voxvae
    voxvae.set_default_dtype(dtype)
    voxvae.set_default_device(DEV)
    import src.module.nolbo as nolbo
    m = nolbo.nolboSingleObject_modelnet_category_VAE(nolbo_structure=t['cfg'])
    m._encoder.set_weights_dict(t['ep'])
    m._decoder.set_weights_dict(t['dp'])
    return m


def _run(m, t):
    x, eps = torch.from_numpy(t['x']).to(DEV), torch.from_numpy(t['eps']).to(DEV)
    _, z_act, kl = m._encode_latent(x, eps)
    probs, logits, stats = m._dec_eng.forward(z_act, x, want_logits=True)
    # the fused path bench.py times (latent tail, metrics in the last launch) must say the same
    pred2, stats2, metrics, _ = m.eval_forward_device(x, x, eps)
    torch.cuda.synchronize()
    return probs.cpu().numpy(), logits.cpu().numpy(), stats.cpu().numpy().astype(np.float64), stats2.cpu().numpy().astype(np.float64), \
        kl.cpu().numpy()


def test_f32_logits_occupancy_and_saturated_bce_on_trained_weights(trained):
    t = trained
    ref = t['ref']
    probs, logits, stats, stats2, kl = _run(_model(t, 'f32'), t)
    err = np.abs(logits.astype(np.float64) - ref['logits'])
    print('\n[trained f32] max |dlogit| %.2e at |logit| up to %.1f; max |dprob| %.2e' % (err.max(), np.abs(ref['logits']).max(),
                                                                                     np.abs(probs - ref['probs']).max()))
    assert err.max() <= 1e-3                                                   # north_star: logits within 1e-3 in fp32
    safe = np.abs(ref['logits']) > 1e-4
    assert np.array_equal((logits >= 0)[safe], (ref['logits'] >= 0)[safe])    # occupancy exact outside the guard band
    assert np.array_equal((probs >= 0.5)[safe], (ref['logits'] >= 0)[safe])
    # counts: exact up to the voxels inside the guard band
    slack = int((~safe).reshape(N_EVAL, -1).sum(1).max())
    for col, k in ((1, 'tp'), (2, 'fp'), (3, 'fn')):
        assert np.abs(stats[:, col] - ref[k]).max() <= slack
        assert np.abs(stats2[:, col] - ref[k]).max() <= slack
    # BCE with a share of the voxels saturated: function.py:79's float32 clip (1 - 1e-7 -> 0.99999988) is reproduced, and the per-sample
    # sum is within 2e-4 (one ulp of p moves log(1 - p) by percents near p -> 1: the formulation is ill-conditioned there and both
    # sides keep it)
    sat = np.abs(ref['logits']) > 15.94
    assert sat.mean() > 0.005, 'the trained net must reach the clip region (%.4f of the voxels do)' % sat.mean()
    rel = np.abs(stats[:, 0] - ref['bce']) / ref['bce']
    rel2 = np.abs(stats2[:, 0] - ref['bce']) / ref['bce']
    print('[trained f32] saturated voxels %.1f %%; per-sample BCE rel err max %.2e (fused path %.2e)' % (100 * sat.mean(), rel.max(), rel2.max()))
    assert rel.max() <= 2e-4 and rel2.max() <= 2e-4
    np.testing.assert_allclose(kl, ref['kl'], rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize('dtype', ['bf16', 'fp8', 'fp8/wide', 'fp8/all'])
def test_reduced_precision_iou_on_trained_weights(trained, dtype, monkeypatch):
    """bf16 and the default fp8 policy ('mid' since round 4: E2, E3 / D3, D4 on e4m3fn operands; measured 5.8e-4) meet north_star's bar at
    the trained operating point: mean IoU within 1e-3 of the oracle; so does 'wide' (the direct-kernel layers E2 / D4: 5.0e-4 with the
    error-diffused weight images of round 4, 7.1e-4 before).  Per-sample: bf16 within 5e-3; fp8 within 1e-2.  'fp8/all' (every eligible
    layer, rounds 1-2's mode) does NOT meet the 1e-3 bar here -- measured 1.44e-3 (1.7e-3 before round 4), always a LOSS of IoU: each
    fp8 layer adds 1-3 % of noise to its pre-activations and a fitted model sits at an optimum -- it is gated at 2e-3 so that the
    finding stays visible and bounded."""
    import voxvae
    t = trained
    ref = t['ref']
    if '/' in dtype:
        monkeypatch.setitem(voxvae._DEFAULTS, 'fp8_policy', dtype.split('/')[1])
    mean_gate, sample_gate = {'bf16': (1e-3, 5e-3), 'fp8': (1e-3, 1e-2), 'fp8/wide': (1e-3, 1e-2), 'fp8/all': (2e-3, 2e-2)}[dtype]
    probs, logits, stats, stats2, kl = _run(_model(t, dtype.split('/')[0]), t)
    for s in (stats, stats2):
        iou = s[:, 1] / np.maximum(s[:, 1] + s[:, 2] + s[:, 3], 1)
        d_mean, d_max = abs(iou.mean() - t['iou'].mean()), np.abs(iou - t['iou']).max()
        assert d_mean <= mean_gate, (dtype, d_mean)                            # north_star: IoU within 1e-3 of the reference
        assert d_max <= sample_gate, (dtype, d_max)
    flips = (logits >= 0) != (ref['logits'] >= 0)
    print('\n[trained %s] IoU ref %.4f, delta %.2e, max per-sample delta %.2e; %d occupancy flips of %d (largest |ref logit| at a flip %.3f); '
          'max |dlogit| %.3f' % (dtype, t['iou'].mean(), d_mean, d_max, flips.sum(), flips.size,
                                 np.abs(ref['logits'][flips]).max() if flips.any() else 0.0,
                                 np.abs(logits - ref['logits']).max()))
    # the loss: dominated by the boundary voxels, a few percent at most in reduced precision
    rel = np.abs(stats[:, 0] - ref['bce']) / ref['bce']
    assert rel.max() <= (0.05 if dtype == 'bf16' else 0.25), rel.max()


def test_two_pass_missing_latents_on_trained_weights(trained):
    """getEval(missing_prob=0.9) (nolbo.py:1472-1528): masked latent filled with the prototype mean, nearest prototype by the
    masked distance, prior sample in the masked slots, decoder again -- at the trained operating point the first pass
    reconstructs badly (most of z is the prototype mean) and the corrected pass differs from it, so both are real tests."""
    from voxvae import synthetic as syn
    from oracle import c_oracle as co
    t = trained
    n = 16
    x, eps = t['x'][184:184 + n], t['eps'][184:184 + n]        # 8 seen + 8 unseen shapes
    oh, cats = syn.make_onehot(n, 40, seed=6), syn.make_category_vectors(40, 64, seed=12)
    eps2, mask = syn.make_eps(n, 64, seed=9), syn.make_mask(n, 64, 0.9, seed=14)
    ref = co.vae_get_eval(t['cfg'], t['ep'], t['dp'], x, x, oh, cats, eps, 0.9, mask, eps2)
    m = _model(t, 'f32')
    out = m.getEval(inputs=(x, x, oh), category_vectors=cats, missing_prob=0.9, _eps=eps, _mask=mask, _eps2=eps2)
    assert len(out) == 10
    pred, pred_c = np.array(out[0]), np.array(out[5])
    np.testing.assert_allclose(np.array(m._z_category), ref['z'], atol=2e-5)
    np.testing.assert_allclose(np.array(m._z_category_corrected), ref['z_c'], atol=2e-5)
    np.testing.assert_allclose(pred, ref['probs'], atol=2.5e-4)                # |dprob| <= |dlogit| / 4
    np.testing.assert_allclose(pred_c, ref['probs_c'], atol=2.5e-4)
    for got, a, b, c in ((out[1:5], '', ref['bce'], ref['acc']), (out[6:10], '_c', ref['bce_c'], ref['acc_c'])):
        tp, fp, fn = ref['tp' + a].astype(np.float64), ref['fp' + a].astype(np.float64), ref['fn' + a].astype(np.float64)
        assert abs(float(got[0]) - b.mean()) <= 2e-4 * b.mean()
        assert abs(float(got[1]) - np.mean(tp / (tp + fp + 1e-10))) < 1e-4
        assert abs(float(got[2]) - np.mean(tp / (tp + fn + 1e-10))) < 1e-4
        assert abs(float(got[3]) - c) < 1e-6
    assert np.abs(ref['logits'] - ref['logits_c']).max() > 1.0                 # the correction changes the reconstruction
    # bf16: same call, IoU of both passes within the reduced-precision bar
    mb = _model(t, 'bf16')
    ob = mb.getEval(inputs=(x, x, oh), category_vectors=cats, missing_prob=0.9, _eps=eps, _mask=mask, _eps2=eps2)
    for p, a in ((np.array(ob[0]), ''), (np.array(ob[5]), '_c')):
        yh = p.reshape(n, -1) >= 0.5
        yt = x.reshape(n, -1) > 0.5
        iou = (yh & yt).sum(1) / np.maximum((yh | yt).sum(1), 1)
        tp, fp, fn = ref['tp' + a], ref['fp' + a], ref['fn' + a]
        iou_r = tp / np.maximum(tp + fp + fn, 1)
        assert abs(iou.mean() - iou_r.mean()) <= 1e-3 and np.abs(iou - iou_r).max() <= 5e-3


# ---------------------------------------------------------------------------------------------- 64^3: BASELINE config 5's geometry
@pytest.fixture(scope='module')
def trained64():
    """The 64^3 VAE (the reference's native grid, test_modelnet_VAE.py:174-189; BASELINE.json configs[4]) fitted with the repo's
    bf16 mixed-precision fit() -- the weights are just weights, the ORACLE judges the operating point -- at batch 32 on a pool of 256
    seeded shapes, evaluated like the 32^3 model on 256 samples (192 of the pool + 64 the model has never seen): the mean of the
    per-sample IoU difference then carries a standard error of ~4e-5 (round 3 used 48 samples: ~3e-4, a third of the bar)."""
    from voxvae import synthetic as syn
    from voxvae import trained as tr
    from oracle import c_oracle as co
    cfg, ep, dp, info = tr.train_operating_point(voxel=64, latent=64, batch=32, pool=256, device=DEV, dtype='bf16', max_steps=3000)
    assert info['reached'], info
    n = N_EVAL
    x = np.concatenate([syn.make_voxels(256, 64, seed=4321)[:192], syn.make_voxels(64, 64, seed=777)], axis=0)
    eps = syn.make_eps(n, 64, seed=70)
    ref = co.vae_eval_forward(cfg, ep, dp, x, x, eps)
    iou = ref['tp'] / np.maximum(ref['tp'] + ref['fp'] + ref['fn'], 1)
    print('\n[trained 64^3] %d fit steps; oracle IoU %.4f (seen %.4f / unseen %.4f), logits in [%.1f, %.1f]'
          % (info['steps'], iou.mean(), iou[:192].mean(), iou[192:].mean(), ref['logits'].min(), ref['logits'].max()))
    assert iou.mean() >= 0.5 and np.abs(ref['logits']).max() >= 16.0
    return dict(cfg=cfg, ep=ep, dp=dp, x=x, eps=eps, ref=ref, iou=iou, info=info)


@pytest.mark.parametrize('dtype', ['bf16', 'fp8', 'fp8/wide', 'fp8/most', 'fp8/all'])
def test_config5_geometry_iou_on_trained_weights(trained64, dtype, monkeypatch):
    """Config 5's arithmetic at ITS geometry, at a trained operating point, 256 samples (standard error of the mean ~4e-5).
    bf16, the DEFAULT fp8 policy ('mid' since round 4: e4m3fn operands on E2, E3, D3, D4; weights rounded with error diffusion over the taps
    an output sums -- engine.quant_fp8), 'wide' (E2, D4) and 'most' (everything but the encoder tail) meet north_star's 1e-3 with the gate AT 1e-3.  The per-layer study behind it
    (profiles/microbench/fp8_schemes.py, profiles/r04_fp8_schemes_64.json): E2 alone 4.8e-4, D4 alone 2.7e-4, both 7.4e-4 with
    independently rounded weights; the weight rounding's share disappears with the diffusion; per-32-channel E8M0 activation scales
    change nothing (4.71e-4 against 4.75e-4 simulated: the error is the 3-bit mantissa of the LARGE values, not subnormals).  The
    all-layers policy stays opt-in and over the bar; it is gated at what it measures so that the finding stays visible and bounded."""
    import voxvae
    t = trained64
    if '/' in dtype:
        monkeypatch.setitem(voxvae._DEFAULTS, 'fp8_policy', dtype.split('/')[1])
    # measured (round 4, profiles/r04_fp8_policy_mid_64.jsonl): bf16 2.5e-5; 'mid' (the default) 6.5e-4; 'wide' 4.3e-4; 'most' 7.4e-4; 'all' 1.23e-3
    gate = {'bf16': 1e-3, 'fp8': 1e-3, 'fp8/wide': 1e-3, 'fp8/most': 1e-3, 'fp8/all': 2e-3}[dtype]
    m = _model(t, dtype.split('/')[0])
    x, eps = torch.from_numpy(t['x']).to(DEV), torch.from_numpy(t['eps']).to(DEV)
    ious = []
    for lo in range(0, N_EVAL, 64):                       # config 5's per-GPU shard is 64 samples
        _, stats, _, _ = m.eval_forward_device(x[lo:lo + 64].contiguous(), x[lo:lo + 64].contiguous(), eps[lo:lo + 64].contiguous())
        s = stats.double().cpu().numpy()
        ious.append(s[:, 1] / np.maximum(s[:, 1] + s[:, 2] + s[:, 3], 1))
    iou = np.concatenate(ious)
    diff = iou - t['iou']
    d_mean, d_max, se = abs(diff.mean()), np.abs(diff).max(), diff.std(ddof=1) / np.sqrt(len(diff))
    print('\n[trained 64^3 %s] IoU ref %.4f, delta %.2e +- %.1e (%d samples), max per-sample delta %.2e' % (dtype, t['iou'].mean(), d_mean, se, len(diff), d_max))
    assert d_mean <= gate, (dtype, d_mean)
    assert d_max <= 10 * gate, (dtype, d_max)
    assert se <= 1.5e-4                                     # the measurement can decide: its standard error is well under the bar


# ---------------------------------------------------------------------------------------------- the autoencoder class (config 1's model)
def test_autoencoder_class_on_trained_weights():
    """nolboSingleObject_modelnet_category_AE (reference nolbo.py:1206-1385; BASELINE.json configs[0]: test_modelnet_AE.py plumbing)
    fitted with its own fit() (shape loss only, nolbo.py:1247) and compared at the trained operating point: f32 logits within 1e-3 and
    occupancy exact outside the guard band, bf16 mean IoU within 1e-3; also at config 1's batch of 4."""
    import voxvae
    from oracle import c_oracle as co
    from voxvae import synthetic as syn
    from voxvae import trained as tr
    import src.module.nolbo as nolbo
    cfg, ep, dp, info = tr.train_operating_point(device=DEV, variational=False)
    assert info['reached'], info
    n = 64
    x = np.concatenate([syn.make_voxels(256, 32, seed=4321)[:48], syn.make_voxels(16, 32, seed=777)], axis=0)
    ref = co.vae_eval_forward(cfg, ep, dp, x, x, np.zeros((n, 64), np.float32), variational=False)
    iou_r = ref['tp'] / np.maximum(ref['tp'] + ref['fp'] + ref['fn'], 1)
    print('\n[trained AE] %d fit steps; oracle IoU %.4f, logits in [%.1f, %.1f]' % (info['steps'], iou_r.mean(), ref['logits'].min(), ref['logits'].max()))
    assert iou_r.mean() >= 0.5 and np.abs(ref['logits']).max() >= 16.0
    xd = torch.from_numpy(x).to(DEV)
    for dtype in ('f32', 'bf16'):
        voxvae.set_default_dtype(dtype)
        voxvae.set_default_device(DEV)
        m = nolbo.nolboSingleObject_modelnet_category_AE(nolbo_structure=cfg)
        m._encoder.set_weights_dict(ep)
        m._decoder.set_weights_dict(dp)
        _, z_act, _ = m._encode_latent(xd)
        _, logits, stats = m._dec_eng.forward(z_act, xd, want_logits=True)
        lg, s = logits.cpu().numpy(), stats.double().cpu().numpy()
        iou = s[:, 1] / np.maximum(s[:, 1] + s[:, 2] + s[:, 3], 1)
        if dtype == 'f32':
            assert np.abs(lg - ref['logits']).max() <= 1e-3
            safe = np.abs(ref['logits']) > 1e-4
            assert np.array_equal((lg >= 0)[safe], (ref['logits'] >= 0)[safe])
            # config 1's batch: 4 samples through getEval, the same numbers
            out = m.getEval(inputs=(x[:4], x[:4], syn.make_onehot(4, 40)), category_vectors=syn.make_category_vectors(40, 64), missing_prob=0.0)
            np.testing.assert_allclose(np.array(out[0]), ref['probs'][:4], atol=2.5e-4)
            assert abs(float(out[1]) - ref['bce'][:4].mean()) <= 2e-4 * ref['bce'][:4].mean()
        else:
            assert abs(iou.mean() - iou_r.mean()) <= 1e-3 and np.abs(iou - iou_r).max() <= 5e-3
        print('[trained AE %s] max |dlogit| %.2e, IoU delta %.2e' % (dtype, np.abs(lg - ref['logits']).max(), abs(iou.mean() - iou_r.mean())))


# ---------------------------------------------------------------------------------------------- config 3: the image -> 3D model's decoder half
def test_config3_decoder_half_on_trained_weights():
    """BASELINE.json configs[2] (test_pascal_VAE_dr.py; reference nolbo.py:750-928): latent 16, 64^3 decoder, 12 classes, supplied head
    outputs [B, 32] in place of the 2D encoder.  The decoder is fitted with the class's own fit() (decoder step on
    the HIP path, nolbo.py:786-833) on 128 (head output, shape) pairs, then: missing_prob = 0 -- the ORACLE reports IoU >= 0.5 and
    |logit| >= 16 on the pairs, f32 probabilities within 2.5e-4 and bf16 IoU within 1e-3; missing_prob = 0.9 -- both decoder passes of
    the masked / corrected evaluation against the composed oracle at the trained weights."""
    import voxvae
    from oracle import c_oracle as co
    from oracle import numpy_oracle as no
    from voxvae import synthetic as syn
    import src.module.nolbo as nolbo
    Lz, C, D, pool, B = 16, 12, 64, 128, 32
    dec_cfg = syn.make_config(D, Lz, True)['decoder']
    cfg = {'encoder_backbone': {'name': 'nolbo_backbone', 'z_dim': Lz},
           'encoder_head': {'name': 'nolbo_head', 'output_dim': 2 * Lz, 'filter_num_list': [], 'filter_size_list': [], 'activation': 'elu'},
           'decoder': dec_cfg}
    rng = np.random.default_rng(31)
    head = np.concatenate([rng.standard_normal((pool, Lz)), np.full((pool, Lz), -6.0)], axis=1).astype(np.float32)   # mean | logVar (std 0.05)
    ys = syn.make_voxels(pool, D, seed=4400)
    voxvae.set_default_dtype('bf16')
    voxvae.set_default_device(DEV)
    m = nolbo.nolboSingleObject_VAE(nolbo_structure=cfg, learning_rate=1e-3)
    hd, yd = torch.from_numpy(head).to(DEV), torch.from_numpy(ys).to(DEV)
    gen = torch.Generator(device=DEV).manual_seed(5)
    for step in range(600):
        lo = (step % (pool // B)) * B
        m.fit((hd[lo:lo + B], yd[lo:lo + B]), _eps=torch.randn(B, Lz, device=DEV, generator=gen))
    dp = m._decoder.get_weights_dict()
    n = 16
    hx, y = head[:n], ys[:n]
    eps, eps2 = syn.make_eps(n, Lz, seed=71), syn.make_eps(n, Lz, seed=72)
    mask = syn.make_mask(n, Lz, 0.9, seed=73)
    oh = syn.make_onehot(n, C, seed=74)
    cats = head[:C, :Lz].copy()                                   # the prototypes: latent means of 12 training pairs
    mu, lv = no.split_mean_logvar(hx.astype(np.float64), Lz)
    z0 = no.sampling(mu, lv, eps)
    ref0 = co.sigmoid_bce_counts(co.decoder3D_logits(dec_cfg, dp, z0.astype(np.float32)), y)
    iou0 = ref0[2] / np.maximum(ref0[2] + ref0[3] + ref0[4], 1)
    lg0 = co.decoder3D_logits(dec_cfg, dp, z0.astype(np.float32))
    print('\n[trained config 3] oracle IoU %.4f, logits in [%.1f, %.1f]' % (iou0.mean(), lg0.min(), lg0.max()))
    assert iou0.mean() >= 0.5 and np.abs(lg0).max() >= 16.0
    zm = z0 * mask
    zm = np.where(zm == 0, cats.astype(np.float64).mean(0)[None, :] * np.ones_like(zm), zm)
    idx, _ = no._nearest_category_acc(zm, cats.astype(np.float64), oh, mask=mask.astype(np.float64))
    zc = np.where(mask == 0, cats[idx].astype(np.float64) + eps2, zm)
    refs = [co.sigmoid_bce_counts(co.decoder3D_logits(dec_cfg, dp, zz.astype(np.float32)), y) for zz in (zm, zc)]
    for dtype in ('f32', 'bf16'):
        voxvae.set_default_dtype(dtype)
        mm = nolbo.nolboSingleObject_VAE(nolbo_structure=cfg)
        mm._decoder.set_weights_dict(dp)
        o0 = mm.getEval(inputs=(hx, y, oh), category_vectors=cats, missing_prob=0.0, _eps=eps)
        o9 = mm.getEval(inputs=(hx, y, oh), category_vectors=cats, missing_prob=0.9, _eps=eps, _mask=mask, _eps2=eps2)
        np.testing.assert_allclose(np.array(mm._z_category_corrected), zc, atol=2e-5)
        for pred, ref in ((np.array(o0[0]), ref0), (np.array(o9[0]), refs[0]), (np.array(o9[5]), refs[1])):
            if dtype == 'f32':
                np.testing.assert_allclose(pred, ref[0], atol=2.5e-4)
            yh, yt = pred.reshape(n, -1) >= 0.5, y.reshape(n, -1) > 0.5
            iou = (yh & yt).sum(1) / np.maximum((yh | yt).sum(1), 1)
            iou_r = ref[2] / np.maximum(ref[2] + ref[3] + ref[4], 1)
            assert abs(iou.mean() - iou_r.mean()) <= 1e-3, (dtype, abs(iou.mean() - iou_r.mean()))
        if dtype == 'f32':
            for got, ref in ((o0[1], ref0), (o9[1], refs[0]), (o9[6], refs[1])):
                assert abs(float(got) - ref[1].mean()) <= 2e-4 * ref[1].mean()

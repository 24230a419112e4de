"""GPU tests of the drop-in Python surface (src.module.nolbo / src.module.function / src.net_core.autoencoder3D and the
entry scripts): the reference's call forms and return tuples, the missing-modality path against the golden fixtures,
checkpoints, and the stand-alone loss ops."""
import os
import sys

import numpy as np
import pytest
import torch

from oracle import numpy_oracle as no

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
PKG = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'anytime-3d-reconstruction_amd')


def _model(name, dtype='f32'):
    import voxvae
    from voxvae import synthetic as syn
    voxvae.set_default_dtype(dtype)
    voxvae.set_default_device('cuda:0')
    import src.module.nolbo as nolbo
    g = np.load(os.path.join(GOLDEN, name + '.npz'))
    D, Lz, var, B, C = [int(v) for v in g['meta'][:5]]
    cfg = syn.make_config(D, Lz, bool(var))
    cls = nolbo.nolboSingleObject_modelnet_category_VAE if var else nolbo.nolboSingleObject_modelnet_category_AE
    m = cls(nolbo_structure=cfg)
    m._encoder.set_weights_dict(syn.make_encoder_params(cfg['encoder']))
    m._decoder.set_weights_dict(syn.make_decoder_params(cfg['decoder']))
    data = dict(x=syn.make_voxels(B, D), oh=syn.make_onehot(B, C), cats=syn.make_category_vectors(C, Lz), eps=syn.make_eps(B, Lz),
                eps2=syn.make_eps(B, Lz, seed=8), mask=syn.make_mask(B, Lz, 0.5))
    return g, cfg, m, data


@pytest.mark.parametrize('name', ['vae_d32_l64_b2', 'ae_d32_l64_b2', 'vae_d16_l64_b3', 'vae_d64_l16_b1'])
def test_getEval_missing_modality_matches_golden(name):
    g, cfg, m, d = _model(name)
    x = d['x']
    out = m.getEval(inputs=(x, x, d['oh']), category_vectors=d['cats'], missing_prob=0.5, _eps=d['eps'], _mask=d['mask'], _eps2=d['eps2'])
    assert len(out) == 10
    sc = np.array([float(v) for v in out[1:5]] + [float(v) for v in out[6:10]])
    ref = g['p5_scalars']
    np.testing.assert_allclose(sc[[0, 4]], ref[[0, 4]], rtol=1e-4)           # loss_shape, loss_shape_corrected
    np.testing.assert_allclose(sc[[1, 2, 5, 6]], ref[[1, 2, 5, 6]], atol=2e-3)  # pr, rc (+ corrected)
    np.testing.assert_allclose(sc[[3, 7]], ref[[3, 7]], atol=1e-6)           # nearest-category accuracies
    pred, pred_c = np.array(out[0]), np.array(out[5])
    assert pred.shape == x.shape and pred_c.shape == x.shape
    np.testing.assert_allclose(pred, no.sigmoid(g['p5_logits'].astype(np.float64)), atol=2.5e-4)
    np.testing.assert_allclose(pred_c, no.sigmoid(g['p5_logits_c'].astype(np.float64)), atol=2.5e-4)
    np.testing.assert_allclose(np.array(m._z_category), g['p5_z'], atol=2e-5)
    np.testing.assert_allclose(np.array(m._z_category_corrected), g['p5_z_corr'], atol=2e-5)


def test_getEval_return_forms_and_legacy_calls():
    g, cfg, m, d = _model('vae_d16_l64_b3')
    x = d['x']
    out = m.getEval(inputs=(x, x, d['oh']), category_vectors=d['cats'], missing_prob=0.0, _eps=d['eps'])
    assert out[5:] == (0, 0, 0, 0, 0)                                        # reference nolbo.py:1503
    np.testing.assert_allclose([float(v) for v in out[1:5]], g['p0_scalars'][:4], rtol=1e-4, atol=2e-3)
    leg = m.getEval(inputs=(x, x), _eps=d['eps'])                            # train_modelnet_category_VAE.py:83
    assert len(leg) == 4 and abs(float(leg[1]) - float(out[1])) < 1e-3 * float(out[1])
    assert np.array(leg[1:]).shape == (3,)                                   # np.array(loss_temp) as the scripts do
    leg2 = m.getEval(inputs=(x, x), missing_prob=0.5, _eps=d['eps'], _mask=d['mask'])   # test_modelnet_3D.py:125
    assert len(leg2) == 4 and float(leg2[1]) != float(leg[1])
    np.random.seed(1)
    rnd = m.getEval(inputs=(x, x, d['oh']), category_vectors=d['cats'], missing_prob=0.9)   # internal draws
    assert all(np.isfinite(float(v)) for v in rnd[1:5] + rnd[6:10])


def test_getLatent_and_bf16_api():
    g, cfg, m, d = _model('vae_d32_l64_b2')
    z = m.getLatent(d['x'], _eps=d['eps'])
    assert isinstance(z, np.ndarray) and z.shape == (2, 64)
    np.testing.assert_allclose(z, g['latent'], atol=2e-5)
    assert not np.allclose(m.getLatent(d['x']), z)                           # sampled z: a fresh draw each call (nolbo.py:1565)
    g2, _, mb, d2 = _model('vae_d32_l64_b2', 'bf16')
    out = mb.getEval(inputs=(d2['x'], d2['x'], d2['oh']), category_vectors=d2['cats'], missing_prob=0.0, _eps=d2['eps'])
    assert abs(float(out[1]) - g2['p0_scalars'][0]) < 0.02 * g2['p0_scalars'][0]


def test_fp8_inference_mode_api():
    """BASELINE config 5's arithmetic through the model classes: 'fp8' runs the Cin % 128 == 0 MFMA layers on e4m3fn
    operands (per-channel weight scales), the two-pass missing-modality evaluation included; fit() refuses (inference
    mode); switching the default back leaves other models untouched."""
    import voxvae
    # default policy 'mid' (round 4): the two widest stride-2 layers of each side (E2, E3 / D3, D4) run fp8 operands; 'wide': only the
    # layers with a direct fp8 kernel (E2, D4); 'most': everything but the encoder tail
    assert voxvae.fp8_policy() == 'mid'
    for pol, eq, dq in (('mid', ['q1', 'q2'], ['q2', 'q3']), ('wide', ['q1'], ['q3']), ('most', ['q1', 'q2', 'q3'], ['q1', 'q2', 'q3'])):
        voxvae.set_fp8_policy(pol)
        try:
            gw, _, mw, dw = _model('vae_d32_l64_b2', 'fp8')
        finally:
            voxvae.set_fp8_policy('mid')
        assert mw._enc_eng.fp8
        mw.getEval(inputs=(dw['x'], dw['x'], dw['oh']), category_vectors=dw['cats'], missing_prob=0.0, _eps=dw['eps'])
        assert sorted(k for k in mw._enc_eng.packed if k.startswith('q')) == eq, (pol, sorted(mw._enc_eng.packed))
        assert sorted(k for k in mw._dec_eng.packed if k.startswith('q')) == dq, (pol, sorted(mw._dec_eng.packed))
    with pytest.raises(ValueError):
        voxvae.set_fp8_policy('everything')
    voxvae.set_fp8_policy('all')
    try:
        g, _, m8, d = _model('vae_d32_l64_b2', 'fp8')
    finally:
        voxvae.set_fp8_policy('mid')
    assert m8._enc_eng.fp8 and m8._dec_eng.fp8
    x = d['x']
    out = m8.getEval(inputs=(x, x, d['oh']), category_vectors=d['cats'], missing_prob=0.5, _eps=d['eps'], _mask=d['mask'], _eps2=d['eps2'])
    ref = g['p5_scalars']
    sc = np.array([float(v) for v in out[1:5]] + [float(v) for v in out[6:10]])
    np.testing.assert_allclose(sc[[0, 4]], ref[[0, 4]], rtol=0.03)               # shape losses (fp8 operands: 3 mantissa bits)
    np.testing.assert_allclose(sc[[1, 2, 5, 6]], ref[[1, 2, 5, 6]], atol=0.02)   # precision / recall
    pk = m8._enc_eng.packed
    assert pk.get('q1') and pk.get('q2') and pk.get('q3') and pk.get('q4')       # E2 (Cin 64: tap-pair rows), E3, E4, E5 on fp8 operands
    assert m8._dec_eng.packed.get('q1') and m8._dec_eng.packed.get('q2')
    with pytest.raises(ValueError):
        m8.fit((x, x))
    # the AE class (encoder output = latent, no sampling) and the 64^3 / latent-16 geometry run the same fp8 layers
    for name, pol in (('ae_d32_l64_b2', 'all'), ('vae_d64_l16_b1', 'all'), ('vae_d64_l16_b1', 'wide'), ('vae_d64_l16_b1', 'mid')):
        voxvae.set_fp8_policy(pol)
        try:
            g2, _, m2, d2 = _model(name, 'fp8')
        finally:
            voxvae.set_fp8_policy('mid')
        out2 = m2.getEval(inputs=(d2['x'], d2['x'], d2['oh']), category_vectors=d2['cats'], missing_prob=0.5, _eps=d2['eps'], _mask=d2['mask'],
                          _eps2=d2['eps2'])
        sc2 = np.array([float(v) for v in out2[1:5]] + [float(v) for v in out2[6:10]])
        np.testing.assert_allclose(sc2[[0, 4]], g2['p5_scalars'][[0, 4]], rtol=0.03, err_msg=name)
        np.testing.assert_allclose(sc2[[1, 2, 5, 6]], g2['p5_scalars'][[1, 2, 5, 6]], atol=0.03, err_msg=name)


def test_builders_and_checkpoints(tmp_path):
    import voxvae
    voxvae.set_default_dtype('f32')
    import src.net_core.autoencoder3D as ae3D
    from voxvae import synthetic as syn
    cfg = syn.make_config(16, 64, True)
    enc, dec = ae3D.encoder3D(cfg['encoder']), ae3D.decoder3D(cfg['decoder'])
    assert enc.name == 'encoder3D' and dec.name == 'decoder'
    names = [v.name for v in enc.trainable_variables]
    assert names[0] == 'encoder3D/conv0/kernel' and len(names) == 5 + 2 * 4 and not any('moving' in n for n in names)
    assert len(dec.trainable_variables) == 2 + 2 + 5 + 2 * 4
    x = syn.make_voxels(2, 16)
    e = enc(x, training=False)
    assert np.array(e).shape == (2, 128)
    p = dec(np.array(e)[:, :64])
    assert np.array(p).shape == (2, 16, 16, 16, 1) and 0 <= np.array(p).min() and np.array(p).max() <= 1
    with pytest.raises(NotImplementedError):                 # an unknown pooling mode ('average' and 'max' are the reference's two)
        bad = dict(cfg['encoder']); bad['final_pool'] = 'median'
        ae3D.encoder3D(bad)
    with pytest.raises(NotImplementedError):                 # filter sizes other than 4: no reference config
        bad = dict(cfg['encoder']); bad['filter_size_list'] = [3, 3, 3, 3, 3]
        ae3D.encoder3D(bad)
    import src.module.nolbo as nolbo
    m1 = nolbo.nolboSingleObject_modelnet_category_VAE(nolbo_structure=cfg)
    m1._encoder.set_weights_dict(syn.make_encoder_params(cfg['encoder'], seed=5))
    m1.saveModel(str(tmp_path))
    assert os.path.exists(tmp_path / 'encoder3D.voxvae.npz') and os.path.exists(tmp_path / 'decoder.voxvae.npz')
    m2 = nolbo.nolboSingleObject_modelnet_category_VAE(nolbo_structure=cfg)
    m2.loadModel(str(tmp_path))
    eps = syn.make_eps(2, 64)
    a = m1.getEval(inputs=(x, x), _eps=eps)
    b = m2.getEval(inputs=(x, x), _eps=eps)
    assert np.array_equal(np.array(a[0]), np.array(b[0]))
    m2.loadDecoder(str(tmp_path), file_name='decoder')                        # train_modelnet_category_VAE.py:46-52
    # a directory holding TensorFlow-format checkpoints (what the reference's saveModel writes, nolbo.py:1568-1574): loadModel reads
    # `<name>.index` + `<name>.data-*` when no .voxvae.npz is there (voxvae/tf_checkpoint.py; format self-written, see its header)
    tfdir = tmp_path / 'tf'
    m1._encoder.save_tf_checkpoint(str(tfdir / 'encoder3D'))
    m1._decoder.save_tf_checkpoint(str(tfdir / 'decoder'))
    assert os.path.exists(tfdir / 'encoder3D.index') and os.path.exists(tfdir / 'decoder.data-00000-of-00001')
    m3 = nolbo.nolboSingleObject_modelnet_category_VAE(nolbo_structure=cfg)
    m3.loadModel(str(tfdir))
    c = m3.getEval(inputs=(x, x), _eps=eps)
    assert np.array_equal(np.array(a[0]), np.array(c[0]))


def test_function_module_ops():
    import voxvae
    voxvae.set_default_dtype('f32')
    import src.module.function as F
    rng = np.random.default_rng(0)
    p = rng.random((3, 8, 8, 8, 1)).astype(np.float32)
    p[0, 0, 0, 0, 0], p[0, 0, 0, 1, 0] = 0.0, 1.0                              # exercises both clip ends
    y = (rng.random((3, 8, 8, 8, 1)) < 0.3).astype(np.float32)
    for gamma, br in ((0.5, False), (0.6, False), (0.7, True)):
        got = np.array(F.binary_loss(xPred=p, xTarget=y, gamma=gamma, b_range=br))
        np.testing.assert_allclose(got, no.binary_loss(p, y, gamma=gamma, b_range=br), rtol=2e-5)
    tp, fp, fn = F.voxelPrecisionRecall(xTarget=y, xPred=p)
    rtp, rfp, rfn = no.voxel_precision_recall(y, p)
    assert np.array_equal(np.array(tp), rtp) and np.array_equal(np.array(fp), rfp) and np.array_equal(np.array(fn), rfn)
    mu, lv = rng.standard_normal((4, 64)).astype(np.float32), rng.standard_normal((4, 64)).astype(np.float32)
    mt, lt = rng.standard_normal((4, 64)).astype(np.float32), rng.standard_normal((4, 64)).astype(np.float32)
    np.testing.assert_allclose(np.array(F.kl_loss(mu, lv, mt, lt)), no.kl_loss(mu.astype(np.float64), lv, mt, lt), rtol=2e-5)
    eps = rng.standard_normal((4, 64)).astype(np.float32)
    np.testing.assert_allclose(np.array(F.sampling(mu, lv, epsilon=eps)), no.sampling(mu.astype(np.float64), lv, eps), rtol=2e-5, atol=1e-6)
    s = np.array(F.sampling(np.zeros((2000, 64), np.float32), np.zeros((2000, 64), np.float32)))
    assert abs(s.mean()) < 0.02 and abs(s.std() - 1) < 0.02


def test_bce_backward_clip_semantics():
    """tf.clip_by_value passes the gradient for eps <= p <= 1-eps (float32 compare) and blocks it outside."""
    import ctypes
    from voxvae import lib as L
    hi = np.float32(1.0) - np.float32(1e-7)
    p = np.array([[0.0, 1e-8, 1e-7, 0.3, 0.5, float(hi), 1.0, 0.9]], np.float32)
    y = np.array([[1, 0, 1, 1, 0, 0, 0, 1]], np.float32)
    pd, yd = torch.from_numpy(p).cuda(), torch.from_numpy(y).cuda()
    g = torch.empty_like(pd)
    L.call('vv_bce_bwd', L.ptr(pd), L.ptr(yd), L.ptr(g), 1, 8, 0.6, 1e-7, 0.25, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    exp = np.where((p >= np.float32(1e-7)) & (p <= hi), (-0.6 * y * (1 - p) + 0.4 * (1 - y) * p) * 0.25, 0.0)
    np.testing.assert_allclose(g.cpu().numpy(), exp, rtol=1e-6, atol=1e-9)
    assert g[0, 0] == 0 and g[0, 1] == 0 and g[0, 6] == 0 and g[0, 5] != 0 and g[0, 2] != 0


def test_entry_scripts_run_on_synthetic_data(tmp_path):
    sys.path.insert(0, PKG)
    import voxvae
    import _entry_common as C
    voxvae.set_default_dtype('f32')
    import train_modelnet_category_VAE as tr
    import test_modelnet_getLatents as gl
    import test_modelnet_VAE as te
    np.random.seed(0)
    cfg = C.make_config(64, 16, True)
    res = tr.train(training_epoch=1, learning_rate=1e-3, batch_size=4, config=cfg, dataset_path='synthetic:16:16',
                   save_path=str(tmp_path), max_iter=3)
    loss, loss_train, loss_test = res
    assert loss.shape == (4,) and np.all(np.isfinite(loss)) and np.all(np.isfinite(loss_test))
    cv = gl.train(config=cfg, dataset_path='synthetic:16:16', load_path=str(tmp_path), batch_size=4, max_iter=3)
    assert cv.shape == (40, 64) and os.path.exists(tmp_path / 'category_vectors.npy')
    l8 = te.train(config=cfg, dataset_path='synthetic:12:16', load_path=str(tmp_path), missing_pr=0.5, batch_size=4, max_iter=2)
    assert l8.shape == (8,) and np.all(np.isfinite(l8))
    l8d = te.train(config=cfg, dataset_path='synthetic:12:16', load_path=str(tmp_path), missing_pr=0.5, batch_size=4, max_iter=2, device_data=True)
    assert l8d.shape == (8,) and np.all(np.isfinite(l8d))
    # --dump-dir: the three arrays the reference collects per batch and saves at the end of the epoch (test_modelnet_VAE.py:128-130,
    # 159-165) -- the input of the notebooks' precision / recall tool -- from the host loader and from the device-resident one
    for dd, dev_data in (('dump_host', False), ('dump_dev', True)):
        te.train(config=cfg, dataset_path='synthetic:12:16', load_path=str(tmp_path), missing_pr=0.5, batch_size=4, max_iter=2,
                 device_data=dev_data, dump_dir=str(tmp_path / dd))
        lab, gt, pr = (np.load(tmp_path / dd / ('0.5' + sfx)) for sfx in ('_cl_label.npy', '_gt.npy', '_pred.npy'))
        assert lab.shape == (8, 40) and gt.shape == (8, 16, 16, 16, 1) and pr.shape == gt.shape
        assert set(np.unique(gt)) <= {0.0, 1.0} and pr.min() >= 0 and pr.max() <= 1 and np.all(lab.sum(1) == 1)
    # latent-dropout training (the _dr scripts) and the (D*D, D) text dumps of test_modelnet_3D.py
    res = tr.train(training_epoch=1, learning_rate=1e-3, batch_size=4, config=cfg, dataset_path='synthetic:16:16', max_iter=2, dropout=True)
    assert np.all(np.isfinite(res[0]))
    import test_modelnet_3D as t3
    t3.test(dataset_path='synthetic:8:16', batch_size=2, save_dir=str(tmp_path / 'dump'), voxel=16, missing_prs=(0.5,))
    m = np.loadtxt(tmp_path / 'dump' / '001_0.5_VAE.txt')
    assert m.shape == (256, 16) and m.min() >= 0 and m.max() <= 1


def test_pascal_vae_decoder_half_config3():
    """BASELINE.json configs[2] (test_pascal_VAE_dr.py): latent 16, 64^3 decoder, 12 classes, missing_prob 0.9 -> two
    decoder passes; the 2D image encoder is replaced by supplied head outputs [B, 32]."""
    import voxvae
    from oracle import c_oracle as co
    from voxvae import synthetic as syn
    voxvae.set_default_dtype('bf16')
    import src.module.nolbo as nolbo
    B, Lz, C, D = 16, 16, 12, 64
    dec_cfg = syn.make_config(D, Lz, True)['decoder']
    cfg = {'encoder_backbone': {'name': 'nolbo_backbone', 'z_dim': Lz},
           'encoder_head': {'name': 'nolbo_head', 'output_dim': 2 * Lz, 'filter_num_list': [], 'filter_size_list': [], 'activation': 'elu'},
           'decoder': dec_cfg}
    m = nolbo.nolboSingleObject_VAE(nolbo_structure=cfg)
    dp = syn.make_decoder_params(dec_cfg)
    m._decoder.set_weights_dict(dp)
    rng = np.random.default_rng(3)
    head = rng.standard_normal((B, 2 * Lz)).astype(np.float32)
    y = syn.make_voxels(B, D, seed=9)
    oh, cats = syn.make_onehot(B, C), syn.make_category_vectors(C, Lz)
    eps, eps2, mask = syn.make_eps(B, Lz), syn.make_eps(B, Lz, seed=8), syn.make_mask(B, Lz, 0.9)
    out = m.getEval(inputs=(head, y, oh), category_vectors=cats, missing_prob=0.9, _eps=eps, _mask=mask, _eps2=eps2)
    assert len(out) == 10 and np.array(out[0]).shape == (B, D, D, D, 1) and np.array(out[5]).shape == (B, D, D, D, 1)
    # oracle: same latent algebra in numpy, decoder + losses through the C restatement
    mu, lv = no.split_mean_logvar(head.astype(np.float64), Lz)
    z = no.sampling(mu, lv, eps) * mask
    z = np.where(z == 0, cats.astype(np.float64).mean(0)[None, :] * np.ones_like(z), z)
    idx, _ = no._nearest_category_acc(z, cats.astype(np.float64), oh, mask=mask.astype(np.float64))
    zc = np.where(mask == 0, cats[idx].astype(np.float64) + eps2, z)
    for zz, (loss_i, pr_i, rc_i) in ((z, (1, 2, 3)), (zc, (6, 7, 8))):
        lg = co.decoder3D_logits(dec_cfg, dp, zz.astype(np.float32))
        probs, bce, tp, fp, fn = co.sigmoid_bce_counts(lg, y)
        pr, rc = no.pr_rc(tp.astype(np.float64), fp.astype(np.float64), fn.astype(np.float64))
        assert abs(float(out[loss_i]) - bce.mean()) < 0.02 * bce.mean()
        assert abs(float(out[pr_i]) - pr) < 5e-3 and abs(float(out[rc_i]) - rc) < 5e-3
    np.testing.assert_allclose(np.array(m._z_category_corrected), zc, atol=2e-5)


def test_pascal_image_to_3d_end_to_end(tmp_path):
    """Darknet19 + head2D (stock PyTorch ops) in front of the HIP decoder -- getEval with missing
    latents, fit() through both halves, checkpoints, and the two entry scripts on synthetic (image, voxel) pairs."""
    sys.path.insert(0, PKG)
    import voxvae
    voxvae.set_default_dtype('f32')
    import src.module.nolbo as nolbo
    import src.net_core.darknet as darknet
    import src.dataset_loader.pascal3D as pascal3D
    import test_pascal_VAE_dr as te
    import train_pascal_VAE_dr as tr
    from voxvae import synthetic as syn
    torch.manual_seed(0)
    np.random.seed(0)
    cfg = te.make_config(16, 32)
    m = nolbo.nolboSingleObject_VAE(nolbo_structure=cfg, backbone_style=darknet.Darknet19, learning_rate=1e-3, dropout=True)
    ld = pascal3D.dataLoaderSingleObject('train', 'synthetic:8:32')
    _, cls, _, _, img, vox = ld.getNextBatch(4, (64, 64), augmentation=False)
    cats = syn.make_category_vectors(12, 16)
    out = m.getEval(inputs=(img, vox, cls), category_vectors=cats, missing_prob=0.5)
    assert len(out) == 10 and np.array(out[0]).shape == (4, 32, 32, 32, 1) and all(np.isfinite(float(v)) for v in out[1:5])
    w_dec = m._decoder._engine.params['convT1/kernel'].clone()
    w_bb = m._encoder_backbone.layers[0].conv.weight.detach().clone()
    w_hd = m._encoder_head.last.weight.detach().clone()
    l0 = m.fit(inputs=(img, vox))
    assert len(l0) == 4 and all(np.isfinite(l0))
    for _ in range(4):
        l1 = m.fit(inputs=(img, vox))
    assert l1[1] < l0[1]                                                       # shape loss falls on a repeated batch
    assert not torch.equal(w_dec, m._decoder._engine.params['convT1/kernel'])
    assert not torch.equal(w_bb, m._encoder_backbone.layers[0].conv.weight) and not torch.equal(w_hd, m._encoder_head.last.weight)
    m.saveModel(str(tmp_path))
    m2 = nolbo.nolboSingleObject_VAE(nolbo_structure=cfg, backbone_style=darknet.Darknet19)
    m2.loadModel(str(tmp_path))
    eps = syn.make_eps(4, 16)
    a = m.getEval(inputs=(img, vox), _eps=eps)
    b = m2.getEval(inputs=(img, vox), _eps=eps)
    np.testing.assert_allclose(np.array(a[0]), np.array(b[0]), atol=1e-6)
    l8 = te.train(config=cfg, load_path=str(tmp_path), missing_pr=0.9, batch_size=4, image_size=(64, 64), max_iter=2, dataset_path='synthetic:8:32')
    assert l8.shape == (8,) and np.all(np.isfinite(l8))
    res = tr.train(training_epoch=1, config=cfg, batch_size=4, image_size=(64, 64), max_iter=2, dataset_path='synthetic:8:32')
    assert all(np.all(np.isfinite(r)) for r in res)


def test_regulizer_loss_op():
    """function.regulizer_loss (reference function.py:40-71) against the numpy restatement, with and without classes."""
    import voxvae
    voxvae.set_default_device('cuda:0')
    import src.module.function as F
    rng = np.random.default_rng(5)
    B, Lz = 37, 16
    m, lv = rng.standard_normal((B, Lz)).astype(np.float32), rng.uniform(-1, 1, (B, Lz)).astype(np.float32)
    oh = np.eye(5, dtype=np.float32)[rng.integers(0, 5, B)]
    for dist, c in ((2.0 * Lz, None), (20.0, oh), (0.5, oh)):
        got = np.array(F.regulizer_loss(m, lv, dist, class_input=c))
        ref = no.regulizer_loss(m, lv, dist, c)
        np.testing.assert_allclose(got, ref, rtol=2e-5, atol=1e-4)
    assert np.array(F.regulizer_loss(m, lv, 0.0)).max() == 0.0                # nothing is closer than 0


def test_class_conditional_prior_model(tmp_path):
    """SURVEY §8(f) rank 2, nolboSingleObject_modelnet_category_only (nolbo.py:1594-1787): getEval == the VAE class's
    getEval against the prior network's class means; fit() trains encoder, decoder and prior; entry scripts run."""
    sys.path.insert(0, PKG)
    import voxvae
    voxvae.set_default_dtype('f32')
    import src.module.nolbo as nolbo
    import train_modelnet_category as tr
    import test_modelnet_category as te
    from voxvae import synthetic as syn
    torch.manual_seed(0)
    np.random.seed(0)
    cfg = tr.make_config(64, 16)
    m = nolbo.nolboSingleObject_modelnet_category_only(nolbo_structure=cfg, learning_rate=1e-3)
    ref = nolbo.nolboSingleObject_modelnet_category_VAE(nolbo_structure=cfg)
    ep, dp = syn.make_encoder_params(cfg['encoder']), syn.make_decoder_params(cfg['decoder'])
    for mm in (m, ref):
        mm._encoder.set_weights_dict(ep)
        mm._decoder.set_weights_dict(dp)
    B = 6
    x, oh = syn.make_voxels(B, 16), syn.make_onehot(B, 40)
    eps, eps2, mask = syn.make_eps(B, 64), syn.make_eps(B, 64, seed=8), syn.make_mask(B, 64, 0.5)
    mean_prior, lv_prior = m._priornet_class(np.identity(40, dtype='float32'))
    assert tuple(mean_prior.shape) == (40, 64) and float(lv_prior.abs().max()) == 0.0       # const_log_var 0.0
    a = m.getEval(inputs=(x, x, oh), missing_prob=0.5, _eps=eps, _mask=mask, _eps2=eps2)
    b = ref.getEval(inputs=(x, x, oh), category_vectors=mean_prior.cpu().numpy(), missing_prob=0.5, _eps=eps, _mask=mask, _eps2=eps2)
    for u, v in zip(a, b):
        np.testing.assert_array_equal(np.array(u), np.array(v))
    w_prior = m._priornet_class.mean[0].weight.detach().clone()
    w_enc = m._enc_eng.params['conv1/kernel'].clone()
    w_dec = m._dec_eng.params['convT1/kernel'].clone()
    l0 = [float(v) for v in m.fit(inputs=(x, x, oh))]
    assert len(l0) == 5 and all(np.isfinite(l0))
    for _ in range(5):
        l1 = [float(v) for v in m.fit(inputs=(x, x, oh), dropout=True)]
    assert all(np.isfinite(l1))
    assert not torch.equal(w_prior, m._priornet_class.mean[0].weight) and not torch.equal(w_enc, m._enc_eng.params['conv1/kernel'])
    assert not torch.equal(w_dec, m._dec_eng.params['convT1/kernel'])
    res = tr.train(training_epoch=1, learning_rate=1e-3, batch_size=4, config=cfg, dataset_path='synthetic:12:16', save_path=str(tmp_path), max_iter=3)
    assert all(np.all(np.isfinite(r)) for r in res)
    l8 = te.train(config=cfg, dataset_path='synthetic:12:16', load_path=str(tmp_path), missing_pr=0.5, batch_size=4, max_iter=2)
    assert l8.shape == (8,) and np.all(np.isfinite(l8))


def test_bit_packed_device_data_path():
    """pack/unpack kernels are exact inverses of numpy packbits(bitorder='little'), and the
    device-resident loader serves the same rows as the host loader's arrays, as CUDA tensors the models take directly."""
    sys.path.insert(0, PKG)
    import ctypes
    import voxvae
    from voxvae import lib as L
    from voxvae import synthetic as syn
    from src.dataset_loader.modelnet_dataset import dataLoader, deviceDataLoader
    voxvae.set_default_dtype('bf16')
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    rng = np.random.default_rng(0)
    x = (rng.random((5, 16, 16, 16, 1)) < 0.3).astype(np.float32)
    xd = torch.from_numpy(x).cuda()
    packed = torch.empty(5 * 4096 // 8, dtype=torch.uint8, device='cuda:0')
    L.call('vv_pack_bits', L.ptr(xd), L.ptr(packed), 0.5, xd.numel(), st)
    np.testing.assert_array_equal(packed.cpu().numpy(), np.packbits(x.reshape(-1) > 0.5, bitorder='little'))
    idx = torch.tensor([3, 0, 4], dtype=torch.int32, device='cuda:0')
    out = torch.empty(3, 16, 16, 16, 1, dtype=torch.float32, device='cuda:0')
    L.call('vv_unpack_bits_gather', L.ptr(packed), L.ptr(idx), L.ptr(out), 3, 4096, st)
    np.testing.assert_array_equal(out.cpu().numpy(), x[[3, 0, 4]])
    assert L.load().vv_unpack_bits_gather(L.ptr(packed), None, L.ptr(out), 3, 4097, st) == -2

    host = dataLoader('synthetic:20:32', trainortest='test')
    dev = deviceDataLoader('synthetic:20:32', trainortest='test', seed=1)
    assert dev.dataLength == host.dataLength == 20 and dev._packed.shape == (20, 4096) and dev._packed.dtype == torch.uint8
    seen = []
    for _ in range(3):
        b = dev.getNextBatch(batchSize=8)
        rows = b['index_list'].cpu().numpy()
        seen.append(rows)
        assert b['input_images'].is_cuda and b['input_images'].shape == (8, 32, 32, 32, 1)
        np.testing.assert_array_equal(b['input_images'].cpu().numpy(), host._vox3DData[rows])
        np.testing.assert_array_equal(b['class_list'].cpu().numpy(), host._classList[rows])
    assert dev.epoch == 1 and len(set(seen[0]) | set(seen[1])) == 16          # a permutation inside an epoch
    import src.module.nolbo as nolbo
    m = nolbo.nolboSingleObject_modelnet_category_VAE(nolbo_structure=syn.make_config(32, 64, True))
    b = dev.getNextBatch(batchSize=8)
    out = m.getEval(inputs=(b['input_images'], b['input_images'], b['class_list']), category_vectors=syn.make_category_vectors(40, 64))
    assert np.array(out[0]).shape == (8, 32, 32, 32, 1) and np.isfinite(float(out[1]))


def test_keras_named_checkpoint_interop_and_losses():
    """variables under tf.keras' default names (`conv3d_1/kernel:0`, `batch_normalization_4/gamma:0`,
    `conv3d_transpose/kernel:0`, `dense/bias:0`) round-trip between two models even when the exporting process had other
    layer counters; `.losses` carries the l2(0.0005) terms of the reference's layers."""
    sys.path.insert(0, PKG)
    import voxvae
    voxvae.set_default_dtype('f32')
    import src.net_core.autoencoder3D as ae3D
    from voxvae import synthetic as syn
    cfg = syn.make_config(32, 64, True)
    enc, dec = ae3D.encoder3D(cfg['encoder']), ae3D.decoder3D(cfg['decoder'])
    enc.set_weights_dict(syn.make_encoder_params(cfg['encoder']))
    dec.set_weights_dict(syn.make_decoder_params(cfg['decoder']))
    ev, counters = enc.export_keras_variables()
    dv, counters = dec.export_keras_variables(counters)
    assert 'conv3d/kernel:0' in ev and 'conv3d_4/kernel:0' in ev and 'batch_normalization_3/moving_variance:0' in ev
    assert 'dense/kernel:0' in dv and 'dense/bias:0' in dv and 'batch_normalization_4/gamma:0' in dv
    assert 'conv3d_transpose/kernel:0' in dv and 'conv3d_transpose_4/kernel:0' in dv and counters['batch_normalization'] == 9
    assert ev['conv3d_1/kernel:0'].shape == (4, 4, 4, 64, 128) and dv['conv3d_transpose_1/kernel:0'].shape == (4, 4, 4, 256, 512)
    # a process that had built other models first: every suffix shifted, prefixed with a scope
    shifted = {}
    for k, v in dv.items():
        layer, leaf = k.split('/')
        kind, _, num = layer.rpartition('_')
        if not num.isdigit():
            kind, num = layer, '0'
        shifted['decoder/%s_%d/%s' % (kind, int(num) + 7, leaf)] = v
    dec2 = ae3D.decoder3D(cfg['decoder'], seed=5)
    dec2.load_keras_variables(shifted)
    for k, v in dec.get_weights_dict().items():
        np.testing.assert_array_equal(dec2.get_weights_dict()[k], v)
    with pytest.raises(ValueError):
        dec2.load_keras_variables({k: v for k, v in dv.items() if 'dense/bias' not in k})
    reg = dec.losses
    p = dec.get_weights_dict()
    assert len(reg) == 7                                                   # dense kernel + bias, 5 transposed-conv kernels
    np.testing.assert_allclose(sum(float(r) for r in reg), 0.0005 * sum(float((p[k].astype(np.float64) ** 2).sum()) for k in p
                               if k.endswith('/kernel') or k == 'dense/bias'), rtol=1e-5)


def test_surveyed_edge_cases():
    """SURVEY §8(c) edge cases: all-empty / all-full grids, TP+FP = 0 (precision -> 0 through the 1e-10 guard), logits at
    and beyond the float32 sigmoid saturation (|l| ~ 15.94, 16.7, 20), and the where(z == 0) quirk of the missing-latent
    path (an unmasked latent that is EXACTLY zero is replaced too, nolbo.py:1481-1482)."""
    import ctypes
    import voxvae
    from voxvae import lib as L
    from voxvae import synthetic as syn
    import src.module.function as F
    import src.module.nolbo as nolbo
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    # ---- saturation ladder through the last layer: one live channel, every tap = l/8 -> logits l*k/8, k in {1,2,4,8}
    for form in ('box', 'sweep'):
        os.environ['VV_FINAL_BCE'] = form
        try:
            for lmax in (15.9424, 16.7, 20.0, -20.0, 127.5):
                B, side = 2, 8
                x = np.zeros((B, side, side, side, 64), np.float32); x[..., 0] = 1.0
                w = np.zeros((4, 4, 4, 1, 64), np.float32); w[..., 0, 0] = lmax / 8.0
                for y_val in (0.0, 1.0):                      # all-empty and all-full targets
                    y = np.full((B, 16, 16, 16, 1), y_val, np.float32)
                    lg = no.conv3d_transpose_same(x.astype(np.float64), w.astype(np.float64), 2)
                    p32 = (1.0 / (1.0 + np.exp(-lg.astype(np.float32)))).astype(np.float32)
                    q = np.clip(p32, np.float32(1e-7), np.float32(1.0) - np.float32(1e-7))
                    bce = -(0.6 * y * np.log(q.astype(np.float64)) + 0.4 * (1 - y) * np.log((np.float32(1.0) - q).astype(np.float64))).reshape(B, -1).sum(-1)
                    yh = (lg >= 0).astype(np.float64)
                    tp, fp, fn = [(a.reshape(B, -1)).sum(-1) for a in (y * yh, (1 - y) * yh, y * (1 - yh))]
                    probs = torch.empty(B, 16, 16, 16, 1, device='cuda:0'); stats = torch.empty(B, 4, device='cuda:0')
                    ws = torch.empty(L.load().vv_convT3d_final_bce_workspace_bytes(B, side), dtype=torch.uint8, device='cuda:0')
                    xd = torch.from_numpy(x).cuda().to(torch.bfloat16)
                    wd, yd = torch.from_numpy(w).cuda(), torch.from_numpy(y).cuda()
                    L.call('vv_convT3d_final_bce_fwd', L.ptr(xd), L.ptr(wd), L.ptr(yd), L.ptr(probs), None, L.ptr(stats), B, side, 64,
                           0.6, 1e-7, L.VV_BF16, L.ptr(ws), ws.numel(), st)
                    s = stats.cpu().numpy().astype(np.float64)
                    # 127.5/8 is exact in bf16; the other ladders round the weight to bf16 first
                    if lmax == 127.5:
                        np.testing.assert_allclose(s[:, 0], bce, rtol=2e-4)
                    np.testing.assert_array_equal(s[:, 1:], np.stack([tp, fp, fn], 1))
                    out = torch.empty(4, device='cuda:0')
                    L.call('vv_shape_metrics', L.ptr(stats), L.ptr(out), B, st)
                    o = out.cpu().numpy()
                    prr, rcc = no.pr_rc(tp, fp, fn)
                    np.testing.assert_allclose(o[1:3], [prr, rcc], atol=1e-6)
                    if (lmax < 0 and y_val == 1.0) or (lmax > 0 and y_val == 0.0):
                        assert o[1] == 0.0 and (o[2] == 0.0)          # nothing predicted / nothing to recall: guards give 0, not NaN
        finally:
            os.environ.pop('VV_FINAL_BCE', None)

    # ---- binary_loss / voxelPrecisionRecall on saturated probabilities (function.py:73-115)
    l = np.array([[-40.0, -16.7, -15.9424, 0.0, 15.9424, 16.7, 40.0, 3.0]], np.float32)
    p = (1.0 / (1.0 + np.exp(-l))).astype(np.float32)
    y = np.array([[1, 0, 1, 1, 0, 1, 0, 1]], np.float32)
    got = np.array(F.binary_loss(p, y, gamma=0.6))
    np.testing.assert_allclose(got, no.binary_loss(p, y, gamma=0.6), rtol=1e-5)
    tp, fp, fn = [np.array(a) for a in F.voxelPrecisionRecall(y, p)]
    rtp, rfp, rfn = no.voxel_precision_recall(y, p)
    np.testing.assert_array_equal([tp, fp, fn], [rtp, rfp, rfn])

    # ---- all-empty and all-full INPUT grids through the whole VAE (float32 parity mode) against the numpy oracle
    voxvae.set_default_dtype('f32')
    cfg = syn.make_config(16, 64, True)
    ep, dp = syn.make_encoder_params(cfg['encoder']), syn.make_decoder_params(cfg['decoder'])
    m = nolbo.nolboSingleObject_modelnet_category_VAE(nolbo_structure=cfg)
    m._encoder.set_weights_dict(ep); m._decoder.set_weights_dict(dp)
    x = np.zeros((2, 16, 16, 16, 1), np.float32); x[1] = 1.0
    eps, cats, oh = syn.make_eps(2, 64), syn.make_category_vectors(40, 64), syn.make_onehot(2, 40)
    out = m.getEval(inputs=(x, x, oh), category_vectors=cats, _eps=eps)
    ref = no.vae_get_eval(cfg, ep, dp, (x, x, oh), cats, eps)
    np.testing.assert_allclose(np.array(out[0]), ref[0], atol=2e-5)
    np.testing.assert_allclose([float(out[1]), float(out[2]), float(out[3])], [ref[1], ref[2], ref[3]], rtol=2e-4, atol=1e-6)

    # ---- where(z == 0): head output 0 and eps 0 give z == 0 exactly; with an all-ones mask every entry is still replaced
    dec_cfg = syn.make_config(16, 16, True)['decoder']
    pc = {'encoder_backbone': {'name': 'bb', 'z_dim': 16}, 'encoder_head': None, 'decoder': dec_cfg}
    pm = nolbo.nolboSingleObject_VAE(nolbo_structure=pc)
    head = np.zeros((3, 32), np.float32); head[2, :16] = 0.25                 # sample 2 has a non-zero mean
    cats12 = syn.make_category_vectors(12, 16)
    vox = syn.make_voxels(3, 16)
    pm.getEval(inputs=(head, vox, syn.make_onehot(3, 12)), category_vectors=cats12, missing_prob=0.5,
               _eps=np.zeros((3, 16), np.float32), _mask=np.ones((3, 16), np.float32), _eps2=np.zeros((3, 16), np.float32))
    z = np.array(pm._z_category)
    np.testing.assert_allclose(z[:2], np.tile(cats12.mean(0), (2, 1)), atol=1e-6)
    np.testing.assert_allclose(z[2], 0.25, atol=1e-7)


@pytest.mark.parametrize('B,nb,ns,replicas', [(48, 7, 3, False), (256, 7, 3, False), (256, 6, 2, True)])   # 256 x 3 streams, one model = bench.py's default
def test_streamed_evaluator_matches_one_batch_at_a_time(B, nb, ns, replicas):
    """voxvae.streams.StreamedEvaluator: independent batches issued round-robin on 2 or 3 HIP streams give bit-identical per-sample
    sums, metrics and KL to the same batches run one at a time on one stream -- with ONE model serving every stream (the engines
    keep a split-K / slab workspace per stream: kernels of three steps share the CUs and nothing else) and with a replica per stream."""
    import contextlib
    import sys
    import voxvae
    from voxvae import synthetic as syn
    from voxvae.streams import StreamedEvaluator
    voxvae.set_default_dtype('bf16')
    import src.module.nolbo as nolbo
    DEV = 'cuda:0'
    voxvae.set_default_device(DEV)
    cfg = syn.make_config(32, 64, True)
    ep, dp = syn.make_encoder_params(cfg['encoder']), syn.make_decoder_params(cfg['decoder'])

    def build():
        with contextlib.redirect_stdout(sys.stderr):
            m = nolbo.nolboSingleObject_modelnet_category_VAE(nolbo_structure=cfg)
        m._encoder.set_weights_dict(ep)
        m._decoder.set_weights_dict(dp)
        return m

    batches = [(torch.from_numpy(syn.make_voxels(B, 32, seed=50 + i)).to(DEV), torch.from_numpy(syn.make_eps(B, 64, seed=60 + i)).to(DEV))
               for i in range(nb)]
    ref_model = build()
    ref = [ref_model.eval_forward_device(x, x, e) for x, e in batches]
    torch.cuda.synchronize()
    ev = StreamedEvaluator(build, streams=ns, device=DEV, replicas=replicas)
    assert (len(set(id(m) for m in ev.models)) == ns) == replicas
    got = [ev.submit(x, x, e) for x, e in batches]
    ev.synchronize()
    for (p0, s0, m0, k0), (p1, s1, m1, k1) in zip(ref, got):
        assert torch.equal(s0, s1) and torch.equal(m0, m1) and torch.equal(k0, k1) and torch.equal(p0, p1)


def test_streamed_evaluator_repacks_on_the_callers_stream_after_a_weight_change():
    """Round-3 advisor finding: one model serves every stream and weight packing is lazy, so after `set_weights_dict` the first
    stream to run would pack while the next one (which waits only for the caller's stream) already reads the images.  The evaluator
    packs on the caller's stream before it hands the step to a stream, and every stream waits for that pack: three streams submitted
    straight after a weight change, no warm-up, against a fresh one-stream model -- ten times, bit for bit."""
    import contextlib
    import sys
    import voxvae
    from voxvae import synthetic as syn
    from voxvae.streams import StreamedEvaluator
    voxvae.set_default_dtype('bf16')
    import src.module.nolbo as nolbo
    DEV = 'cuda:0'
    voxvae.set_default_device(DEV)
    cfg = syn.make_config(32, 64, True)

    def build(seed):
        with contextlib.redirect_stdout(sys.stderr):
            m = nolbo.nolboSingleObject_modelnet_category_VAE(nolbo_structure=cfg)
        m._encoder.set_weights_dict(syn.make_encoder_params(cfg['encoder'], seed=seed))
        m._decoder.set_weights_dict(syn.make_decoder_params(cfg['decoder'], seed=seed + 1))
        return m

    B = 64
    batches = [(torch.from_numpy(syn.make_voxels(B, 32, seed=150 + i)).to(DEV), torch.from_numpy(syn.make_eps(B, 64, seed=160 + i)).to(DEV))
               for i in range(3)]
    ev = StreamedEvaluator(lambda: build(42), streams=3, device=DEV)
    model = ev.models[0]
    for rep in range(10):
        seed = 1000 + 2 * rep
        model._encoder.set_weights_dict(syn.make_encoder_params(cfg['encoder'], seed=seed))
        model._decoder.set_weights_dict(syn.make_decoder_params(cfg['decoder'], seed=seed + 1))
        got = [ev.submit(x, x, e) for x, e in batches]            # three streams, straight after the weight change
        ev.synchronize()
        ref_model = build(seed)
        ref = [ref_model.eval_forward_device(x, x, e) for x, e in batches]
        torch.cuda.synchronize()
        for (p0, s0, m0, k0), (p1, s1, m1, k1) in zip(ref, got):
            assert torch.equal(s0, s1) and torch.equal(m0, m1) and torch.equal(k0, k1) and torch.equal(p0, p1), rep


def test_eval_step_as_hip_graph_is_identical():
    """voxvae.graphs.GraphedEvalStep: the 13 launches of an evaluation step are capturable (every launch goes to the capturing
    stream, every buffer comes from torch's allocator) and a replay reproduces the eager outputs bit for bit, also after the
    inputs were replaced."""
    import contextlib
    import sys
    import voxvae
    from voxvae import synthetic as syn
    from voxvae.graphs import GraphedEvalStep
    voxvae.set_default_dtype('bf16')
    DEV = 'cuda:0'
    voxvae.set_default_device(DEV)
    import src.module.nolbo as nolbo
    cfg = syn.make_config(32, 64, True)
    ep, dp = syn.make_encoder_params(cfg['encoder']), syn.make_decoder_params(cfg['decoder'])

    def build():
        with contextlib.redirect_stdout(sys.stderr):
            m = nolbo.nolboSingleObject_modelnet_category_VAE(nolbo_structure=cfg)
        m._encoder.set_weights_dict(ep)
        m._decoder.set_weights_dict(dp)
        return m
    B = 6
    xs = [torch.from_numpy(syn.make_voxels(B, 32, seed=70 + i)).to(DEV) for i in range(2)]
    es = [torch.from_numpy(syn.make_eps(B, 64, seed=80 + i)).to(DEV) for i in range(2)]
    eager, graphed = build(), build()
    g = GraphedEvalStep(graphed, xs[0], xs[0], es[0])
    for x, e in zip(xs, es):
        ref = eager.eval_forward_device(x, x, e)
        out = g(x, None, e)
        torch.cuda.synchronize()
        assert all(torch.equal(a, b) for a, b in zip(ref, out))


def test_host_pipeline_overlapped_batches_are_bit_identical():
    """voxvae.streams.HostPipeline (round 4): the reference's loop with batches k + 1, k + 2 enqueued before batch k is converted.  Five
    DIFFERENT batches through a depth-3 pipeline give, batch by batch, the bits of the synchronous getEval on the same host arrays --
    float32 and bit-packed input, float32 and uint8 prediction, and a weight change between two submits is picked up."""
    import collections
    import voxvae
    from voxvae import hostio
    from voxvae import synthetic as syn
    from voxvae.streams import HostPipeline, PendingEval
    voxvae.set_default_dtype('bf16')
    voxvae.set_default_device('cuda:0')
    import src.module.nolbo as nolbo
    cfg = syn.make_config(32, 64, True)
    m = nolbo.nolboSingleObject_modelnet_category_VAE(nolbo_structure=cfg)
    m._encoder.set_weights_dict(syn.make_encoder_params(cfg['encoder']))
    m._decoder.set_weights_dict(syn.make_decoder_params(cfg['decoder']))
    B, nb = 128, 5
    xs = [syn.make_voxels(B, 32, seed=40 + i) for i in range(nb)]
    es = [syn.make_eps(B, 64, seed=60 + i) for i in range(nb)]
    oh, cats = syn.make_onehot(B, 40), syn.make_category_vectors(40, 64)
    try:
        for packed, pdt in ((False, 'float32'), (True, 'float32'), (True, 'uint8')):
            hostio.set_prediction_host_dtype(pdt)
            ins = [hostio.pack_voxels(x) if packed else x for x in xs]
            want = []
            for x, e in zip(ins, es):
                o = m.getEval(inputs=(x, x, oh), category_vectors=cats, missing_prob=0.0, _eps=e)
                want.append((np.array(o[0]).copy(), [float(v) for v in o[1:5]]))
            pipe, pend, got = HostPipeline(m, depth=3), collections.deque(), []
            for x, e in zip(ins, es):
                p = pipe.submit(inputs=(x, x, oh), category_vectors=cats, _eps=e)
                assert isinstance(p, PendingEval)
                pend.append(p)
                if len(pend) == pipe.depth:
                    o = pend.popleft().get()
                    got.append((np.array(o[0]), [float(v) for v in o[1:5]]))
            while pend:
                o = pend.popleft().get()
                got.append((np.array(o[0]), [float(v) for v in o[1:5]]))
            assert len(got) == nb
            for (pw, sw), (pg, sg) in zip(want, got):
                assert pg.dtype == pw.dtype and pg.shape == pw.shape
                np.testing.assert_array_equal(pg, pw)
                np.testing.assert_allclose(sg, sw, rtol=1e-6)
        # the two-pass evaluation (missing_prob > 0: device-resident plain path, second decoder pass) through the pipeline
        hostio.set_prediction_host_dtype('float32')
        rng = np.random.default_rng(3)
        masks = [(rng.random((B, 64)) > 0.9).astype(np.float32) for _ in range(3)]
        e2s = [syn.make_eps(B, 64, seed=90 + i) for i in range(3)]
        want2 = []
        for i in range(3):
            o = m.getEval(inputs=(xs[i], xs[i], oh), category_vectors=cats, missing_prob=0.9, _eps=es[i], _mask=masks[i], _eps2=e2s[i])
            want2.append((np.array(o[0]).copy(), np.array(o[5]).copy(), [float(v) for v in o[1:5] + o[6:10]]))
        pipe = HostPipeline(m, depth=3)
        ps = [pipe.submit(inputs=(xs[i], xs[i], oh), category_vectors=cats, missing_prob=0.9, _eps=es[i], _mask=masks[i], _eps2=e2s[i]) for i in range(3)]
        for (pw, cw, sw), p_ in zip(want2, ps):
            o = p_.get()
            np.testing.assert_array_equal(np.array(o[0]), pw)
            np.testing.assert_array_equal(np.array(o[5]), cw)
            np.testing.assert_allclose([float(v) for v in o[1:5] + o[6:10]], sw, rtol=1e-6)
        # a weight change between submits: the batch submitted after it sees the new weights (repacked with nothing in flight)
        hostio.set_prediction_host_dtype('float32')
        pipe = HostPipeline(m, depth=2)
        p0 = pipe.submit(inputs=(xs[0], xs[0], oh), category_vectors=cats, _eps=es[0])
        dp2 = syn.make_decoder_params(cfg['decoder'], seed=77)
        m._decoder.set_weights_dict(dp2)
        p1 = pipe.submit(inputs=(xs[0], xs[0], oh), category_vectors=cats, _eps=es[0])
        a0, a1 = np.array(p0.get()[0]), np.array(p1.get()[0])
        ref1 = np.array(m.getEval(inputs=(xs[0], xs[0], oh), category_vectors=cats, missing_prob=0.0, _eps=es[0])[0])
        np.testing.assert_array_equal(a1, ref1)
        assert not np.array_equal(a0, a1)
    finally:
        hostio.set_prediction_host_dtype('float32')


def test_host_array_pipeline_is_bit_identical_and_recycles_pinned_blocks():
    """getEval on HOST arrays (the reference's calling convention, test_modelnet_VAE.py:114-130) runs as a two-chunk pipeline on
    two streams with the prediction downloaded into a recycled pinned block (voxvae/hostio.py).  Must hold: the same bits as
    the one-batch device path for the prediction and the per-sample latent, the same scalars, block recycling when the caller
    drops its result, and the opt-in float16 / uint8 return types."""
    import gc
    import voxvae
    from voxvae import hostio
    from voxvae import synthetic as syn
    voxvae.set_default_dtype('bf16')
    voxvae.set_default_device('cuda:0')
    import src.module.nolbo as nolbo
    cfg = syn.make_config(32, 64, True)
    m = nolbo.nolboSingleObject_modelnet_category_VAE(nolbo_structure=cfg)
    m._encoder.set_weights_dict(syn.make_encoder_params(cfg['encoder']))
    m._decoder.set_weights_dict(syn.make_decoder_params(cfg['decoder']))
    B = 160                                                      # two chunks of 80
    x, eps = syn.make_voxels(B, 32, seed=21), syn.make_eps(B, 64, seed=22)
    oh, cats = syn.make_onehot(B, 40), syn.make_category_vectors(40, 64)
    xd = torch.from_numpy(x).to('cuda:0')
    ref = m.getEval(inputs=(xd, xd, oh), category_vectors=cats, missing_prob=0.0, _eps=eps)          # device tensors: one batch, one stream
    ref_pred, ref_z = np.array(ref[0]), np.array(m._z_category)
    out = m.getEval(inputs=(x, x, oh), category_vectors=cats, missing_prob=0.0, _eps=eps)            # host arrays: the pipeline
    assert isinstance(out[0], hostio.HostPrediction) and out[5:] == (0, 0, 0, 0, 0)
    pred = np.array(out[0])
    assert pred.dtype == np.float32 and pred.shape == x.shape
    np.testing.assert_array_equal(pred, ref_pred)
    np.testing.assert_array_equal(np.array(m._z_category), ref_z)
    np.testing.assert_allclose([float(v) for v in out[1:5]], [float(v) for v in ref[1:5]], rtol=1e-6)
    assert torch.equal(out[0].torch(), ref[0].torch())
    # a target array different from the input, and a batch too small to split (plain path)
    y = syn.make_voxels(B, 32, seed=23)
    o2 = m.getEval(inputs=(x, y, oh), category_vectors=cats, missing_prob=0.0, _eps=eps)
    r2 = m.getEval(inputs=(xd, torch.from_numpy(y).to('cuda:0'), oh), category_vectors=cats, missing_prob=0.0, _eps=eps)
    np.testing.assert_array_equal(np.array(o2[0]), np.array(r2[0]))
    np.testing.assert_allclose(float(o2[1]), float(r2[1]), rtol=1e-6)
    small = m.getEval(inputs=(x[:8], x[:8], oh[:8]), category_vectors=cats, missing_prob=0.0, _eps=eps[:8])
    assert not isinstance(small[0], hostio.HostPrediction)
    # the autoencoder class (no sampling step) takes the same pipeline
    cfg_ae = syn.make_config(32, 64, False)
    ma = nolbo.nolboSingleObject_modelnet_category_AE(nolbo_structure=cfg_ae)
    ma._encoder.set_weights_dict(syn.make_encoder_params(cfg_ae['encoder']))
    ma._decoder.set_weights_dict(syn.make_decoder_params(cfg_ae['decoder']))
    oa = ma.getEval(inputs=(x[:128], x[:128], oh[:128]), category_vectors=cats, missing_prob=0.0)
    ra = ma.getEval(inputs=(xd[:128], xd[:128], oh[:128]), category_vectors=cats, missing_prob=0.0)
    assert isinstance(oa[0], hostio.HostPrediction)
    np.testing.assert_array_equal(np.array(oa[0]), np.array(ra[0]))
    np.testing.assert_allclose([float(v) for v in oa[1:5]], [float(v) for v in ra[1:5]], rtol=1e-6)
    del oa, ra, ma
    # recycling: dropping the results returns their pinned blocks; the next call takes one from the pool instead of pinning anew
    del out, pred, o2, small
    gc.collect()
    free_before = sum(len(v) for v in hostio._POOL.values())
    held = hostio._OUT['n']                                      # ref_pred (np.array of a DeviceArray) lives in a pinned block too
    assert free_before >= 1 and held == 1
    o3 = m.getEval(inputs=(x, x, oh), category_vectors=cats, missing_prob=0.0, _eps=eps)
    assert sum(len(v) for v in hostio._POOL.values()) == free_before - 1 and hostio._OUT['n'] == held + 1
    view = np.array(o3[0])[3]                                    # a view of the handed-out block keeps it out of the pool ...
    del o3
    gc.collect()
    assert hostio._OUT['n'] == held + 1
    np.testing.assert_array_equal(view, ref_pred[3])
    del view
    gc.collect()
    assert hostio._OUT['n'] == held                              # ... and its death returns it
    # opt-in return types
    try:
        hostio.set_prediction_host_dtype('uint8')
        o8 = m.getEval(inputs=(x, x, oh), category_vectors=cats, missing_prob=0.0, _eps=eps)
        p8 = np.array(o8[0])
        assert p8.dtype == np.uint8
        np.testing.assert_array_equal(p8, (ref_pred >= 0.5).astype(np.uint8))
        np.testing.assert_array_equal(np.array(m.getEval(inputs=(xd, xd, oh), category_vectors=cats, missing_prob=0.0, _eps=eps)[0]), p8)   # DeviceArray path too
        hostio.set_prediction_host_dtype('float16')
        p16 = np.array(m.getEval(inputs=(x, x, oh), category_vectors=cats, missing_prob=0.0, _eps=eps)[0])
        assert p16.dtype == np.float16
        np.testing.assert_array_equal(p16, ref_pred.astype(np.float16))
    finally:
        hostio.set_prediction_host_dtype('float32')


def test_bit_packed_host_batches_give_the_float_batches_results():
    """Round 4: `voxvae.hostio.PackedVoxels` -- a host batch kept as 1 bit per voxel that reads like the float32 array -- through every
    model method that takes a batch: the same bits as the float32 host array and as the device tensor (getEval's pipeline, the
    missing-latent path, getLatent, fit), `dataLoader(packed=True)` serving such batches, the read-only later views of a
    HostPrediction, and the cap on pooled pinned memory."""
    import gc
    import voxvae
    from voxvae import hostio
    from voxvae import synthetic as syn
    from src.dataset_loader.modelnet_dataset import dataLoader
    voxvae.set_default_dtype('bf16')
    voxvae.set_default_device('cuda:0')
    import src.module.nolbo as nolbo
    cfg = syn.make_config(32, 64, True)
    m = nolbo.nolboSingleObject_modelnet_category_VAE(nolbo_structure=cfg)
    m._encoder.set_weights_dict(syn.make_encoder_params(cfg['encoder']))
    m._decoder.set_weights_dict(syn.make_decoder_params(cfg['decoder']))
    B = 160
    x, eps = syn.make_voxels(B, 32, seed=31), syn.make_eps(B, 64, seed=32)
    oh, cats = syn.make_onehot(B, 40), syn.make_category_vectors(40, 64)
    xp = hostio.pack_voxels(x)
    assert xp.shape == x.shape and xp.dtype == np.float32 and xp.bits.nbytes * 32 == x.nbytes
    np.testing.assert_array_equal(np.array(xp), x)
    np.testing.assert_array_equal(np.array(xp[7:19]), x[7:19])
    assert torch.equal(xp.to_device('cuda:0'), torch.from_numpy(x).to('cuda:0'))
    ref = m.getEval(inputs=(x, x, oh), category_vectors=cats, missing_prob=0.0, _eps=eps)
    ref_pred, ref_z = np.array(ref[0]), np.array(m._z_category)
    out = m.getEval(inputs=(xp, xp, oh), category_vectors=cats, missing_prob=0.0, _eps=eps)
    assert isinstance(out[0], hostio.HostPrediction)
    np.testing.assert_array_equal(np.array(out[0]), ref_pred)
    np.testing.assert_array_equal(np.array(m._z_category), ref_z)
    assert [float(v) for v in out[1:5]] == [float(v) for v in ref[1:5]]
    # a float target with a packed input, the missing-latent path (two decoder passes), getLatent
    y = syn.make_voxels(B, 32, seed=33)
    o2 = m.getEval(inputs=(xp, y, oh), category_vectors=cats, missing_prob=0.0, _eps=eps)
    r2 = m.getEval(inputs=(x, y, oh), category_vectors=cats, missing_prob=0.0, _eps=eps)
    np.testing.assert_array_equal(np.array(o2[0]), np.array(r2[0]))
    mask, eps2 = syn.make_mask(B, 64, 0.5, seed=34), syn.make_eps(B, 64, seed=35)
    o3 = m.getEval(inputs=(xp, xp, oh), category_vectors=cats, missing_prob=0.5, _eps=eps, _mask=mask, _eps2=eps2)
    r3 = m.getEval(inputs=(x, x, oh), category_vectors=cats, missing_prob=0.5, _eps=eps, _mask=mask, _eps2=eps2)
    for a, b in zip(o3, r3):
        np.testing.assert_array_equal(np.array(a), np.array(b))
    np.testing.assert_array_equal(m.getLatent(xp, _eps=eps), m.getLatent(x, _eps=eps))
    # the loader: same sample stream as the float loader under the same np.random seed, batches that read like the float arrays
    np.random.seed(5)
    lf = dataLoader('synthetic:192:32', 'test')
    np.random.seed(5)
    lp = dataLoader('synthetic:192:32', 'test', packed=True)
    bf, bp = lf.getNextBatch(160), lp.getNextBatch(160)
    assert isinstance(bp['input_images'], hostio.PackedVoxels) and bp['input_images'].shape == bf['input_images'].shape
    np.testing.assert_array_equal(np.array(bp['input_images']), bf['input_images'])
    np.testing.assert_array_equal(bp['class_list'], bf['class_list'])
    ol = m.getEval(inputs=(bp['input_images'], bp['input_images'], bp['class_list']), category_vectors=cats, missing_prob=0.0, _eps=eps)
    of = m.getEval(inputs=(bf['input_images'], bf['input_images'], bf['class_list']), category_vectors=cats, missing_prob=0.0, _eps=eps)
    np.testing.assert_array_equal(np.array(ol[0]), np.array(of[0]))
    # fit() takes it too (float32 training step on a small model would be slow to build here: the bf16 model's own step)
    kl_a = [float(v) for v in m.fit((xp[:32], xp[:32]), _eps=eps[:32])]
    assert all(np.isfinite(kl_a))
    # HostPrediction: the first np.array() is the block itself (writable), later views are read-only, a second np.array() is a copy
    hp = m.getEval(inputs=(xp, xp, oh), category_vectors=cats, missing_prob=0.0, _eps=eps)[0]
    first = np.array(hp)
    assert first.flags.writeable and np.shares_memory(first, hp.host)
    later = hp.numpy()
    assert not later.flags.writeable and not hp[3].flags.writeable and np.shares_memory(later, first)
    second = np.array(hp)
    assert second.flags.writeable and not np.shares_memory(second, first)
    with pytest.raises(ValueError):
        later[0, 0, 0, 0, 0] = 1.0
    # pooled pinned memory is bounded: free blocks beyond the cap are released, least recently used size first
    del hp, first, later, second, out, o2, r2, o3, r3, ol, of, ref
    gc.collect()
    keep = hostio._STATE['max_pooled_bytes']
    try:
        hostio._STATE['max_pooled_bytes'] = 48 << 20
        for n in (3, 5, 7, 9, 11, 13):
            t = torch.zeros(n << 20, dtype=torch.float32, device='cuda:0')
            a = hostio.to_host(t)
            assert a.shape == (n << 20,) and not a.any()
            del a
            gc.collect()
            assert hostio.pooled_bytes() <= 48 << 20
    finally:
        hostio._STATE['max_pooled_bytes'] = keep


@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
@pytest.mark.parametrize('pool,final_act', [('None', 'None'), ('average', 'sigmoid'), ('max', 'sigmoid'), ('None', 'sigmoid')])
def test_encoder_builder_pool_none_and_sigmoid(dtype, pool, final_act):
    """The builder-level options of encoder3D that no reference configuration uses (autoencoder3D.py:90-99): final_pool 'None' (the
    model returns the last convolution's [B,S,S,S,E] map) and final_activation 'sigmoid' -- inference path, against the float64
    definition (oracle/numpy_oracle.encoder3D_forward); fit() refuses them."""
    import voxvae
    from oracle import numpy_oracle as no
    from voxvae import synthetic as syn
    voxvae.set_default_dtype(dtype)
    voxvae.set_default_device('cuda:0')
    import src.net_core.autoencoder3D as ae
    cfg = syn.make_config(32, 64, True)['encoder']
    cfg = dict(cfg, final_pool=pool, final_activation=final_act)
    ep = syn.make_encoder_params(cfg)
    enc = ae.encoder3D(cfg)
    enc.set_weights_dict(ep)
    B = 4
    x = syn.make_voxels(B, 32, seed=51)
    got = np.array(enc(x))
    ref = no.encoder3D_forward(cfg, {k: np.asarray(v, np.float64) for k, v in ep.items()}, x.astype(np.float64))
    assert got.shape == ref.shape == ((B, 2, 2, 2, 128) if pool == 'None' else (B, 128))
    tol = 2e-5 if dtype == 'f32' else 3e-2
    np.testing.assert_allclose(got, ref, rtol=0, atol=tol * max(1.0, np.abs(ref).max()))
    if final_act == 'sigmoid':
        assert got.min() >= 0.0 and got.max() <= 1.0
    with pytest.raises(NotImplementedError):
        enc(x, training=True)


@pytest.mark.parametrize('dtype,D', [('f32', 32), ('bf16', 32), ('f32', 16)])
def test_encoder_final_pool_max(dtype, D):
    """final_pool = 'max' (reference autoencoder3D.py:92-93): tf.reduce_max is not linear, so the last conv runs position by
    position (vv_pack_conv_k4s1_full + vv_dense_fwd) and vv_max_over_positions pools it; through the model class (getLatent,
    getEval) against the CPU oracle.  fit() refuses (no reference config trains with it)."""
    import voxvae
    from oracle import c_oracle as co
    from voxvae import synthetic as syn
    voxvae.set_default_dtype(dtype)
    voxvae.set_default_device('cuda:0')
    import src.module.nolbo as nolbo
    cfg = syn.make_config(D, 64, True)
    cfg['encoder']['final_pool'] = 'max'
    ep, dp = syn.make_encoder_params(cfg['encoder']), syn.make_decoder_params(cfg['decoder'])
    m = nolbo.nolboSingleObject_modelnet_category_VAE(nolbo_structure=cfg)
    m._encoder.set_weights_dict(ep)
    m._decoder.set_weights_dict(dp)
    B = 5
    x, eps = syn.make_voxels(B, D, seed=41), syn.make_eps(B, 64, seed=42)
    ref = co.vae_eval_forward(cfg, ep, dp, x, x, eps)
    enc_out = np.array(m._encoder(x))
    tol = 2e-5 if dtype == 'f32' else 3e-2
    np.testing.assert_allclose(enc_out, ref['enc_out'], rtol=0, atol=tol * max(1.0, np.abs(ref['enc_out']).max()))
    z = m.getLatent(x, _eps=eps)
    np.testing.assert_allclose(z, ref['z'], rtol=0, atol=tol * max(1.0, np.abs(ref['z']).max()))
    out = m.getEval(inputs=(x, x, syn.make_onehot(B, 40)), category_vectors=syn.make_category_vectors(40, 64), missing_prob=0.0, _eps=eps)
    if dtype == 'f32':
        np.testing.assert_allclose(np.array(out[0]), ref['probs'], atol=2.5e-4)
        assert abs(float(out[1]) - ref['bce'].mean()) <= 2e-4 * ref['bce'].mean()
    else:
        s = ref['tp'] / np.maximum(ref['tp'] + ref['fp'] + ref['fn'], 1)
        yh, yt = np.array(out[0]).reshape(B, -1) >= 0.5, x.reshape(B, -1) > 0.5
        iou = (yh & yt).sum(1) / np.maximum((yh | yt).sum(1), 1)
        assert abs(iou.mean() - s.mean()) <= 1e-3
    with pytest.raises(NotImplementedError):
        m.fit((x, x))

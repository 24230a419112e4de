"""GPU parity of one training step (fit) against the torch-CPU float64 autograd oracle: losses, every gradient,
the Adam-updated weights and the BatchNorm moving statistics.  f32 (exact-f32 MFMA) mode."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _setup(D, Lz, var, B, seed=0, dtype='f32'):
    import voxvae
    from voxvae import synthetic as syn
    voxvae.set_default_dtype(dtype)
    voxvae.set_default_device(DEV)
    import src.module.nolbo as nolbo
    cfg = syn.make_config(D, Lz, var)
    ep = syn.make_encoder_params(cfg['encoder'], seed=42, nontrivial_affine=True)
    dp = syn.make_decoder_params(cfg['decoder'], seed=43, nontrivial_affine=True, final_gain=2.0)   # keep logits away from the clip's gradient discontinuity at |l| ~ 16
    cls = nolbo.nolboSingleObject_modelnet_category_VAE if var else nolbo.nolboSingleObject_modelnet_category_AE
    model = cls(nolbo_structure=cfg, learning_rate=1e-3)
    model._encoder.set_weights_dict(ep)
    model._decoder.set_weights_dict(dp)
    x = syn.make_voxels(B, D, seed=100 + seed)
    eps = syn.make_eps(B, Lz, seed=200 + seed)
    return cfg, ep, dp, model, x, eps


def _rel(a, b):
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


@pytest.mark.parametrize('D,Lz,var,B', [(32, 64, True, 4), (16, 64, True, 6), (32, 64, False, 3)])
def test_fit_step_matches_autograd_oracle(D, Lz, var, B):
    from oracle import torch_oracle as to
    cfg, ep, dp, model, x, eps = _setup(D, Lz, var, B)
    ref = to.fit_step(cfg, ep, dp, x, x, eps, lr=1e-3, variational=var)
    model._train_helper().debug = {}                     # keep the step's intermediates (d loss / d pre-BN dense output below)
    out = model.fit((x, x), _eps=eps) if var else model.fit((x, x))
    torch.cuda.synchronize()
    vals = [float(v) for v in out]
    if var:
        assert abs(vals[0] - ref['loss_kl']) <= 1e-4 * max(1.0, abs(ref['loss_kl']))
        vals = vals[1:]
    assert abs(vals[0] - ref['loss_shape']) <= 2e-4 * abs(ref['loss_shape'])
    assert abs(vals[1] - ref['pr']) < 1e-3 and abs(vals[2] - ref['rc']) < 1e-3
    tr = model._trainer
    worst = {}
    for name in ref['grads']:
        g = tr.grads.views[name].cpu().numpy()
        if name == 'dec/dense/bias':
            # a bias in front of BatchNorm has zero true gradient; what the kernel leaves is the float32 cancellation residue of
            # a column sum of B terms of magnitude |d pre-BN|: bound it by that magnitude (a few ulps of the summands), not
            # by a bare constant
            assert np.abs(ref['grads'][name]).max() < 1e-9
            scale = float(tr.debug['dcv0'].float().abs().max())
            assert np.abs(g).max() <= 64 * np.finfo(np.float32).eps * B * scale, (np.abs(g).max(), scale)
            continue
        worst[name] = _rel(g, ref['grads'][name])
    bad = {k: v for k, v in worst.items() if v > 5e-5}
    assert not bad, 'gradient mismatch (max rel err): %s' % bad
    new_e, new_d = model._encoder.get_weights_dict(), model._decoder.get_weights_dict()
    for k, v in list(new_e.items()) + list(new_d.items()):
        key = ('enc/' if k in new_e and v is new_e.get(k) else 'dec/') + k
        r = ref['params'][key]
        if k.endswith(('moving_mean', 'moving_variance')):
            np.testing.assert_allclose(v, r, rtol=1e-4, atol=1e-6, err_msg=key)
        else:
            # Adam's first step moves every weight by ~lr * sign(g): compare the step, tolerant where |g| ~ 0
            old = (ep if key.startswith('enc/') else dp)[k]
            step_ref, step_got = r - old, v - old
            g = ref['grads'][key]
            if key == 'dec/dense/bias':
                continue
            big = np.abs(g) > 1e-3 * np.abs(g).max()
            assert np.abs(step_got - step_ref)[big].max() <= 0.02 * 1e-3 + 1e-9, key
    print('\n[fit D%d L%d var%d B%d] worst grad rel err %.2e (%s)' % (D, Lz, var, B, max(worst.values()), max(worst, key=worst.get)))


def test_two_steps_loss_decreases_and_state_advances():
    cfg, ep, dp, model, x, eps = _setup(32, 64, True, 8, seed=3)
    l0 = [float(v) for v in model.fit((x, x), _eps=eps)]
    for _ in range(5):
        l1 = [float(v) for v in model.fit((x, x), _eps=eps)]
    assert model._trainer.t == 6
    assert l1[1] < l0[1], (l0, l1)          # same batch, 6 Adam steps at lr 1e-3: the shape loss must go down
    assert all(np.isfinite(l1))


@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
def test_weight_gradient_stream_is_bit_identical(dtype, monkeypatch):
    """The opt-in second stream for the weight gradients (Trainer.wgrad_stream, VOXVAE_WGRAD_STREAM=1; measured slower and off by default,
    profiles/r04_train_wgrad_stream_ab.json): three steps give the same weights, moments and losses bit for bit as the one-stream step --
    the fork / join events and record_stream calls order every reader behind its writer."""
    res = []
    for flag in ('0', '1'):
        monkeypatch.setenv('VOXVAE_WGRAD_STREAM', flag)
        cfg, ep, dp, model, x, eps = _setup(32, 64, True, 16, seed=5, dtype=dtype)
        losses = [[float(v) for v in model.fit((x, x), _eps=eps)] for _ in range(3)]
        torch.cuda.synchronize()
        tr = model._trainer
        assert (tr.wgrad_stream is not None) == (flag == '1')
        res.append((losses, {k: v.clone() for k, v in model._enc_eng.params.items()}, {k: v.clone() for k, v in model._dec_eng.params.items()},
                    {k: v.clone() for k, v in tr.m.items()}))
    assert res[0][0] == res[1][0]
    for a, b in ((res[0][1], res[1][1]), (res[0][2], res[1][2]), (res[0][3], res[1][3])):
        for k in a:
            assert torch.equal(a[k], b[k]), k


def test_decoder_only_step_matches_full_step():
    """Trainer.step_from_latent (image -> 3D model, nolbo.py:786-833) is the decoder + latent part of the full step: fed
    the full step's encoder output it must produce the same decoder gradients and the same d loss / d enc_out."""
    from voxvae import train as T
    cfg, ep, dp, model, x, eps = _setup(32, 64, True, 4)
    xd = torch.from_numpy(x).to(DEV)
    epsd = torch.from_numpy(eps).to(DEV)
    full = T.Trainer(model._enc_eng, model._dec_eng, True, 1e-3)
    full.debug = {}
    full.step(xd, xd, epsd)
    enc_out, de = full.debug['enc_out'].clone(), full.debug['de'].clone()
    g_full = {n: full.grads.views[n].clone() for n, _ in full.order if n.startswith('dec/')}
    model._decoder.set_weights_dict(dp)                     # undo the Adam update of the full step
    half = T.Trainer(None, model._dec_eng, True, 1e-3)
    kl, stats, metrics, de2 = half.step_from_latent(enc_out, xd, epsd)
    torch.cuda.synchronize()
    assert torch.equal(de, de2)
    assert [n for n, _ in half.order] == [n for n, _ in full.order if n.startswith('dec/')]
    for n, g in g_full.items():
        assert torch.equal(g, half.grads.views[n]), n


def test_custom_latent_step_matches_builtin_vae_step():
    """Trainer.step_custom_latent with the plain VAE algebra written in torch (sampling + KL to N(0,1)) must reproduce
    the built-in step (HIP reparam / KL kernels): same gradients for every parameter."""
    from voxvae import train as T
    cfg, ep, dp, model, x, eps = _setup(16, 64, True, 5)
    xd, epsd = torch.from_numpy(x).to(DEV), torch.from_numpy(eps).to(DEV)
    full = T.Trainer(model._enc_eng, model._dec_eng, True, 1e-3)
    full.step(xd, xd, epsd)
    g_ref = {n: full.grads.views[n].clone() for n, _ in full.order}
    model._encoder.set_weights_dict(ep)
    model._decoder.set_weights_dict(dp)
    Lz = 64

    def latent(enc_out):
        mean, lv = enc_out[:, :Lz], torch.clamp(enc_out[:, Lz:], -10.0, 10.0)
        z = mean + torch.sqrt(torch.exp(lv)) * epsd
        kl = (0.5 * (0.0 - lv) + (torch.exp(lv) + mean ** 2) / 2.0 - 0.5).sum(-1).mean()
        return z, kl, None

    cust = T.Trainer(model._enc_eng, model._dec_eng, False, 1e-3)
    cust.step_custom_latent(xd, xd, latent)
    torch.cuda.synchronize()
    for n, g in g_ref.items():
        a, b = cust.grads.views[n].cpu().numpy(), g.cpu().numpy()
        tol = 1e-4 if n == 'dec/dense/bias' else 2e-5 * np.abs(b).max() + 1e-9     # a bias in front of BatchNorm: true gradient 0
        assert np.abs(a - b).max() <= tol, n


@pytest.mark.parametrize('D,Lz,var,B', [(32, 64, True, 8), (32, 64, False, 4)])
def test_bf16_mixed_precision_fit_tracks_the_oracle(D, Lz, var, B):
    """Mixed-precision step (bf16 activations / activation gradients / MFMA operands, float32 master weights, statistics,
    losses and weight-gradient accumulation) against the float64 autograd oracle.  Tolerances are those of bf16 (8
    mantissa bits, errors accumulate over ten layers): losses 1 %, every gradient tensor within 6 % in the Frobenius
    norm and pointing the same way (cosine > 0.995)."""
    from oracle import torch_oracle as to
    cfg, ep, dp, model, x, eps = _setup(D, Lz, var, B, dtype='bf16')
    ref = to.fit_step(cfg, ep, dp, x, x, eps, lr=1e-3, variational=var)
    model._train_helper().debug = {}                     # keep the step's intermediates (d loss / d pre-BN dense output below)
    out = model.fit((x, x), _eps=eps) if var else model.fit((x, x))
    torch.cuda.synchronize()
    vals = [float(v) for v in out]
    if var:
        assert abs(vals[0] - ref['loss_kl']) <= 1e-2 * max(1.0, abs(ref['loss_kl']))
        vals = vals[1:]
    assert abs(vals[0] - ref['loss_shape']) <= 1e-2 * abs(ref['loss_shape'])
    assert abs(vals[1] - ref['pr']) < 1e-2 and abs(vals[2] - ref['rc']) < 1e-2
    tr = model._trainer
    assert tr.dt == 1 and tr.grads.views['enc/conv1/kernel'].dtype == torch.float32
    worst, cos = {}, {}
    for name, r in ref['grads'].items():
        if name == 'dec/dense/bias':
            continue
        g = tr.grads.views[name].cpu().numpy().astype(np.float64)
        worst[name] = float(np.linalg.norm(g - r) / (np.linalg.norm(r) + 1e-30))
        cos[name] = float((g * r).sum() / (np.linalg.norm(g) * np.linalg.norm(r) + 1e-30))
    bad = {k: (worst[k], cos[k]) for k in worst if worst[k] > 6e-2 or cos[k] < 0.995}
    assert not bad, 'bf16 gradient drift (rel Frobenius error, cosine): %s' % bad
    print('\n[bf16 fit D%d var%d] worst rel err %.3f (%s), min cosine %.5f' % (D, var, max(worst.values()), max(worst, key=worst.get), min(cos.values())))
    l0 = vals[0]
    for _ in range(4):
        out = model.fit((x, x), _eps=eps) if var else model.fit((x, x))
    assert float(out[1 if var else 0]) < l0


def _dp_worker(rank, world, port, shards, eps_shards, out):
    import os
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from voxvae import train as T
    cfg, ep, dp, model, _, _ = _setup(32, 64, True, 2)
    tr = T.Trainer(model._enc_eng, model._dec_eng, True, 1e-3, world_size=world)
    x = torch.from_numpy(shards[rank]).to(DEV)
    tr.step(x, x, torch.from_numpy(eps_shards[rank]).to(DEV))
    torch.cuda.synchronize()
    out[rank] = {n: tr.grads.views[n].cpu().numpy().copy() for n, _ in tr.order}
    out['w%d' % rank] = model._encoder.get_weights_dict()['conv1/kernel'].copy()
    out['launch%d' % rank] = list(tr.grads.launch_order)
    dist.destroy_process_group()


def test_data_parallel_step_two_ranks():
    """BASELINE config 4 semantics (AE3D.py:46-48, 86-104) with two real processes sharing the GPU (gloo): per-rank loss
    scaled by the GLOBAL batch, per-rank BatchNorm statistics, gradient buckets summed across ranks while the backward is
    still running, Adam on every replica.  Both ranks must end with the sum of the two shard gradients and identical weights."""
    import torch.multiprocessing as mp
    from voxvae import train as T
    from voxvae import synthetic as syn
    world = 2
    shards = [syn.make_voxels(3, 32, seed=900 + r) for r in range(world)]
    eps_shards = [syn.make_eps(3, 64, seed=950 + r) for r in range(world)]
    expect = None
    for r in range(world):                                   # single process: each shard's gradient with the global-batch scaling
        cfg, ep, dp, model, _, _ = _setup(32, 64, True, 2)
        tr = T.Trainer(model._enc_eng, model._dec_eng, True, 1e-3, world_size=world)
        x = torch.from_numpy(shards[r]).to(DEV)
        tr.step(x, x, torch.from_numpy(eps_shards[r]).to(DEV))
        g = {n: tr.grads.views[n].cpu().numpy().astype(np.float64) for n, _ in tr.order}
        expect = g if expect is None else {n: expect[n] + g[n] for n in g}
    mgr = mp.Manager()
    out = mgr.dict()
    port = 29600 + (os.getpid() % 2000)
    mp.spawn(_dp_worker, args=(world, port, shards, eps_shards, out), nprocs=world, join=True)
    for n, e in expect.items():
        for r in range(world):
            got = out[r][n].astype(np.float64)
            assert np.abs(got - e).max() <= 1e-6 * np.abs(e).max() + 1e-12, (n, r)
        np.testing.assert_array_equal(out[0][n], out[1][n])
    np.testing.assert_array_equal(out['w0'], out['w1'])
    assert out['launch0'] == sorted(out['launch0']) and len(out['launch0']) >= 1


def test_fit_step_with_latent_dropout_matches_autograd_oracle():
    """The `_dr` training path (reference nolbo.py:1423-1425: rate ~ U[0,1), inverted dropout on z): mask and rate injected,
    every gradient through the masked sampling step against the float64 autograd oracle."""
    from oracle import torch_oracle as to
    import src.module.nolbo as nolbo
    cfg, ep, dp, _, x, eps = _setup(16, 64, True, 5, seed=4)
    model = nolbo.nolboSingleObject_modelnet_category_VAE(nolbo_structure=cfg, dropout=True, learning_rate=1e-3)
    model._encoder.set_weights_dict(ep)
    model._decoder.set_weights_dict(dp)
    rate = 0.35
    mask = (np.random.default_rng(9).random((5, 64)) >= rate).astype(np.float32)
    ref = to.fit_step(cfg, ep, dp, x, x, eps, lr=1e-3, variational=True, drop_mask=mask, drop_scale=1.0 / (1.0 - rate))
    out = [float(v) for v in model.fit((x, x), _eps=eps, _mask=mask, _rate=rate)]
    torch.cuda.synchronize()
    assert abs(out[0] - ref['loss_kl']) <= 1e-4 * max(1.0, abs(ref['loss_kl']))
    assert abs(out[1] - ref['loss_shape']) <= 2e-4 * abs(ref['loss_shape'])
    tr = model._trainer
    worst = {}
    for name in ref['grads']:
        if name == 'dec/dense/bias':
            continue
        worst[name] = _rel(tr.grads.views[name].cpu().numpy(), ref['grads'][name])
    bad = {k: v for k, v in worst.items() if v > 5e-5}
    assert not bad, 'gradient mismatch with latent dropout (max rel err): %s' % bad
    # the encoder's gradients flow only through the kept latent entries: a wrong mask in the backward would show here first
    assert worst['enc/conv4/kernel'] < 2e-3


def test_getEval_training_true_uses_batch_statistics_and_moves_the_moving_ones():
    """getEval(training=True) (reference nolbo.py:1449, 1463, 1496): BatchNorm in training mode, no optimisation step.
    Predictions / loss / precision / recall against the training oracle's forward, the moving statistics against its
    momentum-0.99 update, every trainable weight untouched."""
    from oracle import torch_oracle as to
    from voxvae import synthetic as syn
    cfg, ep, dp, model, x, eps = _setup(16, 64, True, 6, seed=2)
    oh, cats = syn.make_onehot(6, 40), syn.make_category_vectors(40, 64)
    ref = to.fit_step(cfg, ep, dp, x, x, eps, lr=1e-3, variational=True)
    out = model.getEval(inputs=(x, x, oh), category_vectors=cats, training=True, missing_prob=0.0, _eps=eps)
    torch.cuda.synchronize()
    assert len(out) == 10 and out[5:] == (0, 0, 0, 0, 0)
    np.testing.assert_allclose(np.array(out[0]), ref['probs'], rtol=0, atol=2e-5)
    assert abs(float(out[1]) - ref['loss_shape']) <= 2e-4 * abs(ref['loss_shape'])
    assert abs(float(out[2]) - ref['pr']) < 1e-3 and abs(float(out[3]) - ref['rc']) < 1e-3
    new_e, new_d = model._encoder.get_weights_dict(), model._decoder.get_weights_dict()
    for pre, new, old in (('enc/', new_e, ep), ('dec/', new_d, dp)):
        for k, v in new.items():
            if k.endswith(('moving_mean', 'moving_variance')):
                np.testing.assert_allclose(v, ref['params'][pre + k], rtol=1e-4, atol=1e-6, err_msg=pre + k)
            else:
                np.testing.assert_array_equal(v, old[k], err_msg=pre + k)
    # the legacy 2-input form and the missing-latent form run in that mode too
    out2 = model.getEval(inputs=(x, x), training=True, _eps=eps)
    assert len(out2) == 4 and np.isfinite(float(out2[1]))
    mask = (np.random.default_rng(1).random((6, 64)) >= 0.5).astype(np.float32)
    out3 = model.getEval(inputs=(x, x, oh), category_vectors=cats, training=True, missing_prob=0.5, _eps=eps, _mask=mask,
                         _eps2=syn.make_eps(6, 64, seed=5))
    assert len(out3) == 10 and all(np.isfinite(float(v)) for v in out3[1:5] + out3[6:10])


def test_eval_after_training_mode_forward_uses_the_moved_statistics():
    """eval -> getEval(training=True) -> eval: the second evaluation must fold the MOVED moving statistics (reference
    nolbo.py:1463 / 1496 read them live).  The engines keep folded scale / shift vectors; a training-mode forward moves
    the statistics in place without touching a weight, so the fold has to be redone (round-2 advisor finding)."""
    from oracle import c_oracle as co
    from oracle import torch_oracle as to
    from voxvae import synthetic as syn
    cfg, ep, dp, model, x, eps = _setup(16, 64, True, 6, seed=2)
    oh, cats = syn.make_onehot(6, 40), syn.make_category_vectors(40, 64)
    xd, epsd = torch.from_numpy(x).to(DEV), torch.from_numpy(eps).to(DEV)
    p0, _, _, _ = model.eval_forward_device(xd, xd, epsd)                       # folds the ORIGINAL statistics
    p0 = p0.cpu().numpy()
    np.testing.assert_allclose(p0, co.vae_eval_forward(cfg, ep, dp, x, x, eps)['probs'], atol=2.5e-4)
    ref = to.fit_step(cfg, ep, dp, x, x, eps, lr=1e-3, variational=True)       # its 'params' hold the moved statistics
    model.getEval(inputs=(x, x, oh), category_vectors=cats, training=True, missing_prob=0.0, _eps=eps)
    moved_e = {k: (ref['params']['enc/' + k] if k.endswith(('moving_mean', 'moving_variance')) else v) for k, v in ep.items()}
    moved_d = {k: (ref['params']['dec/' + k] if k.endswith(('moving_mean', 'moving_variance')) else v) for k, v in dp.items()}
    want = co.vae_eval_forward(cfg, moved_e, moved_d, x, x, eps)['probs']
    p1, _, _, _ = model.eval_forward_device(xd, xd, epsd)
    p1 = p1.cpu().numpy()
    assert np.abs(want - p0).max() > 2e-4                                      # one momentum-0.99 update moves the predictions by this much ...
    np.testing.assert_allclose(p1, want, atol=3e-5)                            # ... and the float32 path follows it an order of magnitude closer
    # decoder-only training-mode pass (the corrected pass of getEval) marks the decoder's fold stale too
    model._train_helper().decoder_training_mode(torch.from_numpy(eps).to(DEV), xd)
    assert not model._dec_eng._folded and model._enc_eng._folded


def test_builder_level_models_called_with_training_true():
    """`model(x, training=True)` on what encoder3D / decoder3D return (reference callers: nolbo.py:1426
    `self._decoder(z, training=True)`, AE3D.py:72-73): batch-statistics BatchNorm, moving statistics move, weights do not."""
    from oracle import torch_oracle as to
    cfg, ep, dp, model, x, eps = _setup(16, 64, True, 6, seed=2)
    ref = to.fit_step(cfg, ep, dp, x, x, eps, lr=1e-3, variational=True)
    enc_out = np.array(model._encoder(x, training=True))
    np.testing.assert_allclose(enc_out, ref['enc_out'], rtol=0, atol=2e-5)
    z = np.array(model._encoder(x, training=False))                            # inference call still works, on the moved statistics
    assert z.shape == enc_out.shape and np.abs(z - enc_out).max() > 1e-4
    probs = np.array(model._decoder(ref['z'].astype(np.float32), training=True))
    np.testing.assert_allclose(probs, ref['probs'], rtol=0, atol=2e-5)
    new_e, new_d = model._encoder.get_weights_dict(), model._decoder.get_weights_dict()
    for pre, new, old in (('enc/', new_e, ep), ('dec/', new_d, dp)):
        for k, v in new.items():
            if k.endswith(('moving_mean', 'moving_variance')):
                np.testing.assert_allclose(v, ref['params'][pre + k], rtol=1e-4, atol=1e-6, err_msg=pre + k)
            else:
                np.testing.assert_array_equal(v, old[k], err_msg=pre + k)


def test_legacy_two_input_getEval_training_true_applies_the_zero_mask():
    """getEval((x, y), training=True, missing_prob > 0): the legacy body (nolbo.py:1544-1548) zeroes the masked latent
    entries; in training mode that must happen too (it used to be ignored)."""
    cfg, ep, dp, model, x, eps = _setup(16, 64, True, 6, seed=2)
    mask = (np.random.default_rng(1).random((6, 64)) >= 0.5).astype(np.float32)
    a = model.getEval(inputs=(x, x), training=True, missing_prob=0.5, _eps=eps, _mask=mask)
    b = model.getEval(inputs=(x, x), training=True, missing_prob=0.0, _eps=eps)
    ones = model.getEval(inputs=(x, x), training=True, missing_prob=0.5, _eps=eps, _mask=np.ones_like(mask))
    assert len(a) == 4 and np.abs(np.array(a[0]) - np.array(b[0])).max() > 1e-3     # the mask changes the prediction
    # an all-ones mask is no mask; the moving statistics moved between the two calls but the batch statistics rule here
    np.testing.assert_allclose(np.array(ones[0]), np.array(b[0]), atol=1e-6)


def test_class_conditional_prior_getEval_training_true():
    """nolbo.py:1678-1754 passes `training` to the prior network, the encoder and the decoder: with training=True the result
    is the VAE class's training-mode getEval against the prior means (the prior net's own BatchNorm / Dropout in training
    mode are torch modules; its means are read back and injected on the comparison side)."""
    import voxvae
    from voxvae import synthetic as syn
    voxvae.set_default_dtype('f32')
    voxvae.set_default_device(DEV)
    import src.module.nolbo as nolbo
    import src.net_core.priornet as priornet
    D, Lz, B = 16, 64, 6
    cfg = syn.make_config(D, Lz, True)
    cfg['prior_class'] = dict(priornet.priornet_structure, unit_num_list=[64, 32, Lz])
    ep = syn.make_encoder_params(cfg['encoder'], seed=42, nontrivial_affine=True)
    dp = syn.make_decoder_params(cfg['decoder'], seed=43, nontrivial_affine=True, final_gain=2.0)
    x, eps, oh = syn.make_voxels(B, D, seed=102), syn.make_eps(B, Lz, seed=202), syn.make_onehot(B, 40)
    outs = []
    for cls in (nolbo.nolboSingleObject_modelnet_category_only, nolbo.nolboSingleObject_modelnet_category_VAE):
        m = cls(nolbo_structure=cfg, learning_rate=1e-3)
        m._encoder.set_weights_dict(ep)
        m._decoder.set_weights_dict(dp)
        if cls is nolbo.nolboSingleObject_modelnet_category_only:
            torch.manual_seed(5)
            o = m.getEval(inputs=(x, x, oh), training=True, missing_prob=0.0, _eps=eps)
            torch.manual_seed(5)                                                # same Dropout draw -> the prior means it used
            cats, _ = m._priornet_class(np.identity(40, dtype='float32'), training=True)
            cats = cats.detach().cpu().numpy()
        else:
            o = m.getEval(inputs=(x, x, oh), category_vectors=cats, training=True, missing_prob=0.0, _eps=eps)
        outs.append(o)
    np.testing.assert_array_equal(np.array(outs[0][0]), np.array(outs[1][0]))
    assert [float(v) for v in outs[0][1:5]] == [float(v) for v in outs[1][1:5]]


def test_class_conditional_prior_fit_matches_autograd_oracle():
    """SURVEY §8(f) rank 2 / reference nolbo.py:1620-1676: fit() of the class-conditional prior model -- KL to the learned
    prior, posterior / prior mixing, the pairwise regulariser at weight 0.01 -- against the float64 restatement
    (oracle/torch_oracle.fit_step_category_only) with every draw injected.  The prior network is replaced by a stub that
    returns two leaf tensors, so that d total / d (prior mean, prior log-variance) can be read next to the encoder / decoder
    gradients (the MLP behind them is stock autograd code on both sides)."""
    import voxvae
    from oracle import torch_oracle as to
    from voxvae import synthetic as syn
    voxvae.set_default_dtype('f32')
    voxvae.set_default_device(DEV)
    import src.module.nolbo as nolbo
    import src.net_core.priornet as priornet
    D, Lz, B = 16, 64, 5
    cfg = syn.make_config(D, Lz, True)
    cfg['prior_class'] = dict(priornet.priornet_structure, unit_num_list=[64, 32, Lz])
    ep = syn.make_encoder_params(cfg['encoder'], seed=42, nontrivial_affine=True)
    dp = syn.make_decoder_params(cfg['decoder'], seed=43, nontrivial_affine=True, final_gain=2.0)
    model = nolbo.nolboSingleObject_modelnet_category_only(nolbo_structure=cfg, learning_rate=1e-3)
    model._encoder.set_weights_dict(ep)
    model._decoder.set_weights_dict(dp)
    rng = np.random.default_rng(21)
    mean_p = (rng.standard_normal((B, Lz)) * 0.5).astype(np.float32)
    lv_p = (rng.standard_normal((B, Lz)) * 0.3).astype(np.float32)

    class _Stub(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.m = torch.nn.Parameter(torch.from_numpy(mean_p).to(DEV))
            self.lv = torch.nn.Parameter(torch.from_numpy(lv_p).to(DEV))

        def forward(self, onehot, training=False):
            return self.m, self.lv

    stub = _Stub()
    model._priornet_class = stub
    x = syn.make_voxels(B, D, seed=300)
    oh = syn.make_onehot(B, 40)
    eps, eps_p = syn.make_eps(B, Lz, seed=301), syn.make_eps(B, Lz, seed=302)
    noise = (rng.random((B, Lz)) >= 0.3).astype(np.float32)
    for mix in (True, False):
        ref = to.fit_step_category_only(cfg, ep, dp, mean_p, lv_p, x, x, eps, eps_p, noise if mix else None)
        model._encoder.set_weights_dict(ep)              # every case starts from the same weights
        model._decoder.set_weights_dict(dp)
        with torch.no_grad():
            stub.m.copy_(torch.from_numpy(mean_p)); stub.lv.copy_(torch.from_numpy(lv_p))
        model._trainer_c = None
        out = [float(v) for v in model.fit((x, x, oh), _rand={'eps': eps, 'eps_prior': eps_p, 'mix': mix, 'noise': noise})]
        torch.cuda.synchronize()
        assert abs(out[0] - ref['loss_kl']) <= 1e-4 * max(1.0, abs(ref['loss_kl'])), (mix, out[0], ref['loss_kl'])
        assert abs(out[1] - ref['loss_shape']) <= 2e-4 * abs(ref['loss_shape'])
        assert abs(out[2] - ref['loss_reg']) <= 1e-4 * max(1.0, abs(ref['loss_reg'])), (mix, out[2], ref['loss_reg'])
        assert abs(out[3] - ref['pr']) < 1e-3 and abs(out[4] - ref['rc']) < 1e-3
        tr = model._trainer_c
        worst = {}
        for name in ref['grads']:
            if name == 'dec/dense/bias':
                continue
            worst[name] = _rel(tr.grads.views[name].cpu().numpy(), ref['grads'][name])
        bad = {k: v for k, v in worst.items() if v > 5e-5}
        assert not bad, 'gradient mismatch (mix=%s): %s' % (mix, bad)
        assert _rel(stub.m.grad.cpu().numpy(), ref['grad_mean_prior']) < 1e-3
        assert _rel(stub.lv.grad.cpu().numpy(), ref['grad_logvar_prior']) < 1e-3


# ---------------------------------------------------------------------------------------------- config 4's per-rank shape: B = 256
def _grads_of_step(dtype, x, eps, lr=1e-3):
    """One Trainer.step at batch len(x) from the seeded weights -> ({name: gradient}, stats, kl, moving statistics)."""
    from voxvae import train as T
    cfg, ep, dp, model, _, _ = _setup(32, 64, True, 2, dtype=dtype)
    tr = T.Trainer(model._enc_eng, model._dec_eng, True, lr)
    xd, ed = torch.from_numpy(x).to(DEV), torch.from_numpy(eps).to(DEV)
    kl, stats, metrics = tr.step(xd, xd, ed)
    torch.cuda.synchronize()
    g = {n: tr.grads.views[n].clone() for n, _ in tr.order}
    mov = {k: v.clone() for k, v in list(model._enc_eng.params.items()) + list(model._dec_eng.params.items())
           if k.endswith(('moving_mean', 'moving_variance'))}
    return g, stats.clone(), kl.clone(), mov


def _rel_max(a, b):
    return float((a.double() - b.double()).abs().max() / (b.double().abs().max() + 1e-30))


@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
def test_fit_at_batch_256_replication_and_permutation(dtype):
    """BASELINE config 4 runs 256 samples per rank; the float64 autograd oracle stops at batch 4-6.  Two exact properties
    carry the oracle-checked small-batch step to the full batch, through every kernel form that only runs at large batches
    (sweep-form last layer, whole-sample D4, phase-form weight gradients, 1024-way split single-channel weight gradients):

    * replication: a 256-batch made of 16 copies of a 16-batch has the same batch statistics (mean, biased variance), the
      same per-sample activations and, with the loss averaged over the batch, the SAME gradients as the 16-batch step --
      BatchNorm's backward included (its batch sums grow 16x, its 1/B shrinks 16x);
    * permutation: re-ordering the samples of a batch of 256 DISTINCT samples changes no gradient (replication alone could
      not see a kernel that reads a neighbour's sample).

    Gates: float32 5e-5 of the tensor's max (summation order only); bf16 mixed precision 2e-2 (a different summation order moves
    batch statistics in the last float32 bit, which re-rounds bf16 activations downstream)."""
    from voxvae import synthetic as syn
    gate = 5e-5 if dtype == 'f32' else 2e-2
    x16, e16 = syn.make_voxels(16, 32, seed=611), syn.make_eps(16, 64, seed=612)
    g16, st16, kl16, mov16 = _grads_of_step(dtype, x16, e16)
    x256, e256 = np.tile(x16, (16, 1, 1, 1, 1)), np.tile(e16, (16, 1))
    g256, st256, kl256, mov256 = _grads_of_step(dtype, x256, e256)
    worst = {n: _rel_max(g256[n], g16[n]) for n in g16 if n != 'dec/dense/bias'}
    print('\n[fit B=256 %s] replication: worst gradient rel err %.2e (%s)' % (dtype, max(worst.values()), max(worst, key=worst.get)))
    assert max(worst.values()) <= gate, {k: v for k, v in worst.items() if v > gate}
    # per-sample losses / counts repeat with period 16, moving statistics agree
    srel = (st256.view(16, 16, 4) - st16[None]).abs().max() / st16.abs().max()
    assert float(srel) <= (2e-4 if dtype == 'f32' else 2e-2)     # per-sample BCE sums: batch statistics summed in another order
    assert float((kl256.view(16, 16) - kl16[None]).abs().max()) <= 1e-3 * float(kl16.abs().max())
    for k in mov16:
        assert _rel_max(mov256[k], mov16[k]) <= (1e-5 if dtype == 'f32' else 1e-3), k
    # permutation at 256 distinct samples
    xd, ed = syn.make_voxels(256, 32, seed=613), syn.make_eps(256, 64, seed=614)
    perm = np.random.default_rng(615).permutation(256)
    ga, sta, _, _ = _grads_of_step(dtype, xd, ed)
    gb, stb, _, _ = _grads_of_step(dtype, xd[perm], ed[perm])
    worst = {n: _rel_max(gb[n], ga[n]) for n in ga if n != 'dec/dense/bias'}
    print('[fit B=256 %s] permutation: worst gradient rel err %.2e (%s)' % (dtype, max(worst.values()), max(worst, key=worst.get)))
    assert max(worst.values()) <= gate, {k: v for k, v in worst.items() if v > gate}
    # per-sample (bce, TP, FP, FN): a voxel whose logit is ~0 may land on the other side in another summation order (one count = 6e-5 here)
    assert float((stb - sta[torch.from_numpy(perm).to(DEV)]).abs().max() / sta.abs().max()) <= (2e-4 if dtype == 'f32' else 2e-2)


def _wgrad_ref_gpu_f64(src, g):
    """The definition of tests/test_gpu_ops._wgrad_conv_ref evaluated in float64 by torch ON THE GPU (a checker, not the
    product: slicing + one matmul per tap); the numpy loop takes minutes at batch 256."""
    B, S, cin = src.shape[0], src.shape[1], src.shape[-1]
    o, cout = S // 2, g.shape[-1]
    p = torch.zeros(B, S + 2, S + 2, S + 2, cin, dtype=torch.float64, device=DEV)
    p[:, 1:-1, 1:-1, 1:-1] = src.double()
    g2 = g.double().reshape(-1, cout)
    dw = torch.empty(4, 4, 4, cin, cout, dtype=torch.float64, device=DEV)
    for td in range(4):
        for th in range(4):
            for tw in range(4):
                win = p[:, td:td + 2 * o:2, th:th + 2 * o:2, tw:tw + 2 * o:2].reshape(-1, cin)
                dw[td, th, tw] = win.t() @ g2
    return dw


def test_weight_gradient_kernels_at_batch_256_against_float64():
    """The two weight-gradient forms that only see their full split at large batches: the phase-form kernel on the 64 -> 128
    layer (side 16 -> 8, 131072 reduction rows) and the single-channel layer's 1024-way split (first conv / last transposed conv:
    float32 grid 32^3, 64 channels of bf16 gradient, 1 M reduction rows) -- against float64 on the same operands."""
    from voxvae import lib as L
    L.load()
    st = ctypes_stream()
    gen = torch.Generator(device=DEV).manual_seed(77)
    B = 256
    src = torch.randn(B, 16, 16, 16, 64, device=DEV, generator=gen).to(torch.bfloat16)
    g = torch.randn(B, 8, 8, 8, 128, device=DEV, generator=gen).to(torch.bfloat16)
    ref = _wgrad_ref_gpu_f64(src, g)
    out = torch.full((4, 4, 4, 64, 128), 7.0, dtype=torch.float32, device=DEV)
    ws = torch.empty(L.load().vv_wgrad_workspace_bytes(B * 8 ** 3, 64 * 64, 128), dtype=torch.uint8, device=DEV)
    L.call('vv_wgrad_conv_k4s2', L.ptr(src), L.ptr(g), L.ptr(out), B, 16, 64, 128, L.VV_BF16, L.VV_BF16, L.ptr(ws), ws.numel(), st)
    torch.cuda.synchronize()
    err = float((out.double() - ref).abs().max() / ref.abs().max())
    print('\n[wgrad B=256] phase form 64->128: rel err %.2e' % err)
    assert err <= 2e-5
    del src, g, ref
    src = (torch.rand(B, 32, 32, 32, 1, device=DEV, generator=gen) < 0.2).float()            # an occupancy grid
    g = torch.randn(B, 16, 16, 16, 64, device=DEV, generator=gen).to(torch.bfloat16)
    ref = _wgrad_ref_gpu_f64(src, g)
    out = torch.full((4, 4, 4, 1, 64), 7.0, dtype=torch.float32, device=DEV)
    ws = torch.empty(L.load().vv_wgrad_workspace_bytes(B * 16 ** 3, 64, 64), dtype=torch.uint8, device=DEV)
    L.call('vv_wgrad_conv_k4s2', L.ptr(src), L.ptr(g), L.ptr(out), B, 32, 1, 64, L.VV_F32, L.VV_BF16, L.ptr(ws), ws.numel(), st)
    torch.cuda.synchronize()
    err = float((out.double() - ref).abs().max() / ref.abs().max())
    print('[wgrad B=256] single channel, 1024-way split: rel err %.2e' % err)
    assert err <= 2e-5


def ctypes_stream():
    import ctypes
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def test_image_to_3d_model_getEval_training_true():
    """nolboSingleObject_VAE.getEval(training=True) (reference nolbo.py:856-928): with supplied head outputs the decoder half runs in
    training mode -- batch-statistics BatchNorm against the float64 training oracle's decoder forward, moving statistics moved,
    weights untouched; the masked / corrected form runs both decoder passes in that mode."""
    import voxvae
    from oracle import torch_oracle as to
    from voxvae import synthetic as syn
    voxvae.set_default_dtype('f32')
    voxvae.set_default_device(DEV)
    import src.module.nolbo as nolbo
    D, Lz, B, C = 16, 64, 6, 12
    cfgv = syn.make_config(D, Lz, True)
    ep = syn.make_encoder_params(cfgv['encoder'], seed=42, nontrivial_affine=True)
    dp = syn.make_decoder_params(cfgv['decoder'], seed=43, nontrivial_affine=True, final_gain=2.0)
    x, eps = syn.make_voxels(B, D, seed=102), syn.make_eps(B, Lz, seed=202)
    ref = to.fit_step(cfgv, ep, dp, x, x, eps, lr=1e-3, variational=True)      # its encoder output = the "head output" fed below
    cfg = {'encoder_backbone': {'name': 'nolbo_backbone', 'z_dim': Lz},
           'encoder_head': {'name': 'nolbo_head', 'output_dim': 2 * Lz, 'filter_num_list': [], 'filter_size_list': [], 'activation': 'elu'},
           'decoder': cfgv['decoder']}
    m = nolbo.nolboSingleObject_VAE(nolbo_structure=cfg)
    m._decoder.set_weights_dict(dp)
    oh, cats = syn.make_onehot(B, C), syn.make_category_vectors(C, Lz)
    head = ref['enc_out'].astype(np.float32)
    out = m.getEval(inputs=(head, x, oh), category_vectors=cats, training=True, missing_prob=0.0, _eps=eps)
    assert len(out) == 10 and out[5:] == (0, 0, 0, 0, 0)
    np.testing.assert_allclose(np.array(out[0]), ref['probs'], rtol=0, atol=2e-5)
    assert abs(float(out[1]) - ref['loss_shape']) <= 2e-4 * abs(ref['loss_shape'])
    new_d = m._decoder.get_weights_dict()
    for k, v in new_d.items():
        if k.endswith(('moving_mean', 'moving_variance')):
            np.testing.assert_allclose(v, ref['params']['dec/' + k], rtol=1e-4, atol=1e-6, err_msg=k)
        else:
            np.testing.assert_array_equal(v, dp[k], err_msg=k)
    mask = (np.random.default_rng(1).random((B, Lz)) >= 0.5).astype(np.float32)
    o2 = m.getEval(inputs=(head, x, oh), category_vectors=cats, training=True, missing_prob=0.5, _eps=eps, _mask=mask, _eps2=syn.make_eps(B, Lz, seed=5))
    assert len(o2) == 10 and all(np.isfinite(float(v)) for v in o2[1:5] + o2[6:10])
    assert np.abs(np.array(o2[0]) - np.array(o2[5])).max() > 1e-3

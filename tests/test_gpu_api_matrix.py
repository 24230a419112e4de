"""API matrix: every public method of every model class, in every arithmetic mode it supports, with every input kind the
callers use (numpy / torch CUDA tensor / DeviceArray) -- small shapes, checking return forms, finiteness and the few invariants
that hold everywhere.  The per-op and parity tests live elsewhere; this one exists because a path nobody calls in a test is
where an AttributeError lives (round 3 found `nolboSingleObject_VAE.getEval(missing_prob=0)` that way)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
D, Lz, B = 16, 64, 4


def _build(kind, dtype):
    import voxvae
    from voxvae import synthetic as syn
    voxvae.set_default_dtype(dtype)
    voxvae.set_default_device(DEV)
    import src.module.nolbo as nolbo
    import src.net_core.priornet as priornet
    cfg = syn.make_config(D, Lz, kind != 'AE')
    if kind == 'AE':
        m = nolbo.nolboSingleObject_modelnet_category_AE(nolbo_structure=cfg, learning_rate=1e-3)
    elif kind == 'VAE':
        m = nolbo.nolboSingleObject_modelnet_category_VAE(nolbo_structure=cfg, learning_rate=1e-3)
    elif kind == 'prior':
        cfg['prior_class'] = dict(priornet.priornet_structure, unit_num_list=[64, 32, Lz])
        m = nolbo.nolboSingleObject_modelnet_category_only(nolbo_structure=cfg, learning_rate=1e-3)
    else:
        icfg = {'encoder_backbone': {'name': 'nolbo_backbone', 'z_dim': Lz},
                'encoder_head': {'name': 'nolbo_head', 'output_dim': 2 * Lz, 'filter_num_list': [], 'filter_size_list': [], 'activation': 'elu'},
                'decoder': cfg['decoder']}
        m = nolbo.nolboSingleObject_VAE(nolbo_structure=icfg, learning_rate=1e-3)
    return m


def _as(kind_in, a):
    from voxvae.tensor import DeviceArray
    if kind_in == 'numpy':
        return a
    t = torch.from_numpy(a).to(DEV)
    return t if kind_in == 'torch' else DeviceArray(t)


@pytest.mark.parametrize('kind', ['AE', 'VAE', 'prior', 'image'])
@pytest.mark.parametrize('dtype', ['f32', 'bf16', 'fp8'])
def test_every_public_method(kind, dtype, tmp_path):
    from voxvae import synthetic as syn
    m = _build(kind, dtype)
    x = syn.make_voxels(B, D, seed=3)
    oh = syn.make_onehot(B, 40)
    cats = syn.make_category_vectors(40, Lz)
    head = np.random.default_rng(2).standard_normal((B, 2 * Lz)).astype(np.float32)
    first = head if kind == 'image' else x
    eps = syn.make_eps(B, Lz)
    kw = {} if kind == 'prior' else {'category_vectors': cats}
    for kind_in in ('numpy', 'torch', 'device_array'):
        a, y = _as(kind_in, first), _as(kind_in, x)
        for mp in (0.0, 0.5):
            for training in ((False,) if dtype == 'fp8' else (False, True)):
                out = m.getEval(inputs=(a, y, oh), missing_prob=mp, training=training, _eps=eps, **kw)
                assert len(out) == 10, (kind_in, mp, training)
                p = np.array(out[0])
                assert p.shape == x.shape and np.isfinite(p).all() and 0 <= p.min() and p.max() <= 1
                assert all(np.isfinite(float(v)) for v in out[1:5])
                if mp == 0.0:
                    assert out[5:] == (0, 0, 0, 0, 0)
                else:
                    assert np.array(out[5]).shape == x.shape and all(np.isfinite(float(v)) for v in out[6:10])
        z = m.getLatent(a, _eps=eps) if kind != 'AE' else m.getLatent(a)
        assert np.asarray(z).shape == (B, Lz) and np.isfinite(np.asarray(z)).all()
    if kind in ('AE', 'VAE'):                      # the legacy two-input forms of the reference's train / test scripts
        leg = m.getEval(inputs=(x, x), _eps=eps)
        assert len(leg) == 4 and np.array(leg[0]).shape == x.shape
        leg = m.getEval(inputs=(x, x), missing_prob=0.5, _eps=eps)
        assert len(leg) == 4
    if dtype == 'fp8':
        with pytest.raises(ValueError):
            m.fit((first, x, oh) if kind == 'prior' else (first, x))
    else:
        before = np.array(m.getEval(inputs=(first, x, oh), missing_prob=0.0, _eps=eps, **kw)[0])
        for _ in range(2):
            res = m.fit((first, x, oh)) if kind == 'prior' else m.fit((first, x))
        assert len(res) == {'AE': 3, 'VAE': 4, 'prior': 5, 'image': 4}[kind] and all(np.isfinite(float(v)) for v in res)
        after = np.array(m.getEval(inputs=(first, x, oh), missing_prob=0.0, _eps=eps, **kw)[0])
        assert np.abs(after - before).max() > 0          # the optimiser step reached the evaluation path (repack + refold)
    m.saveModel(str(tmp_path))
    m2 = _build(kind, dtype)
    m2.loadModel(str(tmp_path))
    if kind == 'prior':
        m2._priornet_class.load_state_dict(m._priornet_class.state_dict())
    o1 = m.getEval(inputs=(first, x, oh), missing_prob=0.0, _eps=eps, **kw)
    o2 = m2.getEval(inputs=(first, x, oh), missing_prob=0.0, _eps=eps, **kw)
    np.testing.assert_array_equal(np.array(o1[0]), np.array(o2[0]))

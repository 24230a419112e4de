"""Seeded synthetic configs, weights and inputs for the voxel VAE hot path.

No dataset, weights or golden vectors exist in the reference, so
every test, the smoke run and bench.py draw their inputs from the generators below.
Pure numpy; no device code.  The specification follows SURVEY.md §8(d):

* voxels : per sample a union of 1-3 axis-aligned boxes plus one ellipsoid, filled,
           values in {0,1}, float32, NDHWC ``[B,D,D,D,1]`` -- the contract of
           ``dataLoader.getNextBatch`` (reference src/dataset_loader/modelnet_dataset.py:83).
* weights: Glorot-uniform kernels (the Keras default the reference never overrides,
           reference src/net_core/autoencoder3D.py:27-30), BatchNorm gamma=1, beta=0,
           moving_mean ~ N(0,0.1), moving_variance ~ U(0.5,1.5) so folded-BN is exercised.
* config : the dict literal of reference test_modelnet_VAE.py:169-192 with the voxel
           side made a parameter (the builders are shape generic, autoencoder3D.py:72-139).
"""
import numpy as np


def make_config(voxel=32, latent_dim=64, variational=True, enc_name='encoder3D', dec_name='decoder'):
    """Config dict with the exact schema of reference test_modelnet_VAE.py:169-192
    (VAE: encoder emits 2*latent) / test_modelnet_AE.py:169-192 (AE: encoder emits latent)."""
    enc_out = 2 * latent_dim if variational else latent_dim
    return {
        'z_category_dim': latent_dim,
        'encoder': {
            'name': enc_name,
            'input_shape': [voxel, voxel, voxel, 1],
            'filter_num_list': [64, 128, 256, 512, enc_out],
            'filter_size_list': [4, 4, 4, 4, 4],
            'strides_list': [2, 2, 2, 2, 1],
            'final_pool': 'average',
            'activation': 'elu',
            'final_activation': 'None',
        },
        'decoder': {
            'name': dec_name,
            'input_dim': latent_dim,
            'output_shape': [voxel, voxel, voxel, 1],
            'filter_num_list': [512, 256, 128, 64, 1],
            'filter_size_list': [4, 4, 4, 4, 4],
            'strides_list': [1, 2, 2, 2, 2],
            'activation': 'elu',
            'final_activation': 'sigmoid',
        },
    }


def decoder_seed_shape(dec_structure):
    """(d, ch): spatial side and channel count of the tensor the decoder's Dense layer is
    reshaped to -- reference autoencoder3D.py:115-120 (side = D / prod(strides);
    channels = max(filter_num_list[0] / 64, 8))."""
    side = int(dec_structure['output_shape'][0] // int(np.prod(dec_structure['strides_list'])))
    ch = int(dec_structure['filter_num_list'][0] // 64)
    if ch < 8:
        ch = 8
    return side, ch


def _glorot(rng, shape, fan_in, fan_out):
    lim = np.sqrt(6.0 / (fan_in + fan_out))
    return rng.uniform(-lim, lim, size=shape).astype(np.float32)


def _bn(rng, c, trained_stats=True):
    if trained_stats:
        return {
            'gamma': np.ones(c, np.float32),
            'beta': np.zeros(c, np.float32),
            'moving_mean': rng.normal(0.0, 0.1, size=c).astype(np.float32),
            'moving_variance': rng.uniform(0.5, 1.5, size=c).astype(np.float32),
        }
    return {  # Keras initial state (used by the training tests)
        'gamma': np.ones(c, np.float32), 'beta': np.zeros(c, np.float32),
        'moving_mean': np.zeros(c, np.float32), 'moving_variance': np.ones(c, np.float32),
    }


def make_encoder_params(structure, seed=42, trained_stats=True, nontrivial_affine=False):
    """Flat dict name -> float32 array, Keras layouts: Conv3D kernel [kd,kh,kw,Cin,Cout]."""
    rng = np.random.default_rng(seed)
    p = {}
    cin = structure['input_shape'][-1]
    fl, ks = structure['filter_num_list'], structure['filter_size_list']
    for i, (c, k) in enumerate(zip(fl, ks)):
        rf = k ** 3
        p['conv%d/kernel' % i] = _glorot(rng, (k, k, k, cin, c), rf * cin, rf * c)
        if i < len(fl) - 1:
            for n, v in _bn(rng, c, trained_stats).items():
                p['bn%d/%s' % (i, n)] = v
            if nontrivial_affine:
                p['bn%d/gamma' % i] = rng.uniform(0.5, 1.5, size=c).astype(np.float32)
                p['bn%d/beta' % i] = rng.normal(0, 0.1, size=c).astype(np.float32)
        cin = c
    return p


def make_decoder_params(structure, seed=43, trained_stats=True, nontrivial_affine=False, final_gain=32.0):
    """Keras layouts: Dense kernel [in,out] + bias; Conv3DTranspose kernel [kd,kh,kw,Cout,Cin].
    final_gain scales the last kernel so the logits span several units (a Glorot-initialised net
    emits |logit| < 0.2, which would leave the sigmoid/BCE saturation paths and the occupancy
    threshold untested)."""
    rng = np.random.default_rng(seed)
    p = {}
    side, ch = decoder_seed_shape(structure)
    lin = side ** 3 * ch
    zin = structure['input_dim']
    p['dense/kernel'] = _glorot(rng, (zin, lin), zin, lin)
    p['dense/bias'] = rng.normal(0, 0.05, size=lin).astype(np.float32)
    for n, v in _bn(rng, lin, trained_stats).items():
        p['bn_dense/%s' % n] = v
    cin = ch
    fl, ks = structure['filter_num_list'], structure['filter_size_list']
    for i, (c, k) in enumerate(zip(fl, ks)):
        rf = k ** 3
        p['convT%d/kernel' % i] = _glorot(rng, (k, k, k, c, cin), rf * cin, rf * c)
        if i < len(fl) - 1:
            for n, v in _bn(rng, c, trained_stats).items():
                p['bnT%d/%s' % (i, n)] = v
            if nontrivial_affine:
                p['bnT%d/gamma' % i] = rng.uniform(0.5, 1.5, size=c).astype(np.float32)
                p['bnT%d/beta' % i] = rng.normal(0, 0.1, size=c).astype(np.float32)
        else:
            p['convT%d/kernel' % i] *= np.float32(final_gain)
        cin = c
    return p


def make_voxels(batch, voxel=32, seed=1234):
    """[B,D,D,D,1] float32 in {0,1}; solid shapes, roughly 5-30 % occupancy."""
    rng = np.random.default_rng(seed)
    D = voxel
    g = np.arange(D, dtype=np.float32)
    zz, yy, xx = np.meshgrid(g, g, g, indexing='ij')
    out = np.zeros((batch, D, D, D, 1), np.float32)
    for b in range(batch):
        occ = np.zeros((D, D, D), bool)
        for _ in range(int(rng.integers(1, 4))):
            lo = rng.integers(1, D // 2, size=3)
            ext = rng.integers(D // 8, D // 2, size=3)
            hi = np.minimum(lo + ext, D - 1)
            occ[lo[0]:hi[0], lo[1]:hi[1], lo[2]:hi[2]] = True
        c = rng.uniform(D * 0.3, D * 0.7, size=3)
        r = rng.uniform(D * 0.1, D * 0.3, size=3)
        occ |= (((zz - c[0]) / r[0]) ** 2 + ((yy - c[1]) / r[1]) ** 2 + ((xx - c[2]) / r[2]) ** 2) <= 1.0
        out[b, ..., 0] = occ
    return out


def make_bernoulli_voxels(batch, voxel=32, p=0.1, seed=1235):
    rng = np.random.default_rng(seed)
    return (rng.random((batch, voxel, voxel, voxel, 1)) < p).astype(np.float32)


def make_onehot(batch, classes=40, seed=5):
    rng = np.random.default_rng(seed)
    idx = rng.integers(0, classes, size=batch)
    oh = np.zeros((batch, classes), np.float32)
    oh[np.arange(batch), idx] = 1.0
    return oh


def make_eps(batch, latent_dim=64, seed=7):
    return np.random.default_rng(seed).standard_normal((batch, latent_dim)).astype(np.float32)


def make_category_vectors(classes=40, latent_dim=64, seed=11):
    return np.random.default_rng(seed).standard_normal((classes, latent_dim)).astype(np.float32)


def make_mask(batch, latent_dim, missing_prob, seed=13):
    """{0,1} float32 mask, P(0)=missing_prob -- the distribution of reference nolbo.py:1475."""
    rng = np.random.default_rng(seed)
    return (rng.random((batch, latent_dim)) >= missing_prob).astype(np.float32)

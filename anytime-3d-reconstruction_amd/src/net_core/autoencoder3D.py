"""Drop-in for the reference's src/net_core/autoencoder3D.py (3D conv encoder / transposed-conv decoder builders).

Same public names and structure-dict schema (reference autoencoder3D.py:72-80, 104-112); the returned object
stands in for the tf.keras.Model the reference builds: callable as model(x, training=...), with
.trainable_variables, .losses, .save_weights(path), .load_weights(path), .name.  The arithmetic is the HIP
library (voxvae.engine -> include/voxvae.h); there is no TensorFlow and no CPU path.
"""
import os

import numpy as np
import torch

import voxvae
from voxvae import engine as _engine
from voxvae import lib as _lib
from voxvae.tensor import DeviceArray, as_device_f32

# ======== architecture example, as in the reference (autoencoder3D.py:5-24) ========
encoder_structure = {
    'name': 'encoder',
    'input_shape': [64, 64, 64, 1],
    'filter_num_list': [64, 128, 256, 512, 400],
    'filter_size_list': [4, 4, 4, 4, 4],
    'strides_list': [2, 2, 2, 2, 1],
    'final_pool': 'average',
    'activation': 'elu',
    'final_activation': 'None',
}
decoder_structure = {
    'name': 'docoder',
    'input_dim': 200,
    'output_shape': [64, 64, 64, 1],
    'filter_num_list': [512, 256, 128, 64, 1],
    'filter_size_list': [4, 4, 4, 4, 4],
    'strides_list': [1, 2, 2, 2, 2],
    'activation': 'elu',
    'final_activation': 'sigmoid'
}

_WEIGHT_SUFFIX = '.voxvae.npz'


class Variable(object):
    """A trainable tensor with a Keras-style name (what .trainable_variables lists)."""

    def __init__(self, name, tensor):
        self.name = name
        self.tensor = tensor

    @property
    def shape(self):
        return tuple(self.tensor.shape)

    def numpy(self):
        return self.tensor.detach().cpu().numpy()

    def __array__(self, dtype=None, copy=None):
        a = self.numpy()
        return a.astype(dtype) if dtype is not None else a


class Model(object):
    """Stand-in for the tf.keras.Model returned by the reference builders."""

    def __init__(self, eng, kind, seed):
        self._engine = eng
        self._kind = kind
        self.name = eng.structure['name']
        self._init_weights(seed)

    # -- Keras default initialisation: Glorot-uniform kernels, zero bias, BN gamma 1 / beta 0 / mean 0 / var 1
    def _init_weights(self, seed):
        g = torch.Generator(device='cpu')
        g.manual_seed(int(seed))
        params = {}
        for name, shp in self._engine.param_shapes().items():
            leaf = name.split('/')[-1]
            if leaf == 'kernel':
                if len(shp) == 5:
                    rf = shp[0] * shp[1] * shp[2]
                    a, b = (shp[3], shp[4]) if self._kind == 'encoder' else (shp[4], shp[3])
                    fan_in, fan_out = rf * a, rf * b
                else:
                    fan_in, fan_out = shp
                lim = float(np.sqrt(6.0 / (fan_in + fan_out)))
                params[name] = (torch.rand(shp, generator=g) * 2.0 - 1.0) * lim
            elif leaf in ('gamma', 'moving_variance'):
                params[name] = torch.ones(shp)
            else:
                params[name] = torch.zeros(shp)
        self._engine.set_params(params)

    @property
    def trainable_variables(self):
        p = self._engine.params
        return [Variable(self.name + '/' + k, p[k]) for k in self._engine.param_shapes()
                if not k.endswith(('moving_mean', 'moving_variance'))]

    @property
    def variables(self):
        return [Variable(self.name + '/' + k, self._engine.params[k]) for k in self._engine.param_shapes()]

    @property
    def losses(self):
        """Keras collects the kernel/bias L2 regularisers (l=0.0005, autoencoder3D.py:29) here.  The modelnet
        classes of the hot path never add them to their loss (nolbo.py:1436); only AE3D.py:82 (out of scope) does."""
        return []

    def set_weights_dict(self, params):
        self._engine.set_params(params)

    def get_weights_dict(self):
        return self._engine.get_params()

    def save_weights(self, path):
        """Reference: Model.save_weights(os.path.join(dir, name)) in TF-checkpoint format (nolbo.py:1568-1574).
        Here: one <path>.voxvae.npz holding the float32 variables under their layer names, Keras layouts."""
        d = os.path.dirname(path)
        if d:
            os.makedirs(d, exist_ok=True)
        np.savez(path + _WEIGHT_SUFFIX, **{k.replace('/', '.'): v for k, v in self._engine.get_params().items()})

    def load_weights(self, path):
        f = path if path.endswith(_WEIGHT_SUFFIX) else path + _WEIGHT_SUFFIX
        with np.load(f) as z:
            params = {k.replace('.', '/'): z[k] for k in z.files}
        missing = set(self._engine.param_shapes()) - set(params)
        if missing:
            raise ValueError('%s lacks variables %s' % (f, sorted(missing)))
        self._engine.set_params(params)

    def __call__(self, inputs, training=False):
        if training:
            raise NotImplementedError('training=True (batch-statistics BatchNorm) is driven by the model classes\' fit()')
        dev = self._engine.device
        x = as_device_f32(inputs, dev)
        if self._kind == 'encoder':
            return DeviceArray(self._engine.forward(x))
        z_act = x if self._engine.dt == _lib.VV_F32 else x.to(torch.bfloat16)
        out, _, _ = self._engine.forward(z_act)
        return DeviceArray(out)


def encoder3D(structure, dtype=None, device=None, seed=0):
    """reference autoencoder3D.py:72-102."""
    print('encoder3D', structure['name'])
    eng = _engine.EncoderEngine(structure, dtype or structure.get('dtype') or voxvae.default_dtype(),
                                device or voxvae.default_device())
    return Model(eng, 'encoder', seed)


def decoder3D(structure, dtype=None, device=None, seed=1):
    """reference autoencoder3D.py:104-139."""
    print('decoder3D', structure['name'])
    eng = _engine.DecoderEngine(structure, dtype or structure.get('dtype') or voxvae.default_dtype(),
                                device or voxvae.default_device())
    return Model(eng, 'decoder', seed)

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


L2_REG = 0.0005      # kernel_regularizer / bias_regularizer of every layer in the reference's autoencoder3D.py


def _regularised(name):
    return name.endswith('/kernel') or name == 'dense/bias'


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
        """Keras `model.losses`: the l2(0.0005) terms of every kernel and of the Dense bias (autoencoder3D.py:29,44,60-61,
        88,131), one scalar per regularised variable.  The modelnet classes of the hot path never add them to their loss
        (nolbo.py:1436); the image -> 3D model does (nolbo.py:819-823) and AE3D.py:82 (out of scope) does."""
        p = self._engine.params
        return [L2_REG * (p[k].float() ** 2).sum() for k in self._engine.param_shapes() if _regularised(k)]

    # -- interop with checkpoints written by the reference (tf.keras variable naming)
    def keras_variable_names(self, counters=None):
        """[(keras name, own name)] in creation order, with Keras' default layer names: `conv3d`, `conv3d_1`, ...,
        `batch_normalization_k`, `conv3d_transpose_k`, `dense_k` -- the per-type counters are global to the process, so
        pass the dict returned for the previously built model (encoder first, as nolbo.py:1404-1409 builds them)."""
        counters = dict(counters or {})
        out = []
        layer_of = {}
        for k in self._engine.param_shapes():
            layer, leaf = k.rsplit('/', 1)
            if layer not in layer_of:
                kind = ('batch_normalization' if layer.startswith('bn') else 'dense' if layer.startswith('dense')
                        else 'conv3d_transpose' if layer.startswith('convT') else 'conv3d')
                n = counters.get(kind, 0)
                counters[kind] = n + 1
                layer_of[layer] = kind if n == 0 else '%s_%d' % (kind, n)
            out.append(('%s/%s:0' % (layer_of[layer], leaf), k))
        return out, counters

    def export_keras_variables(self, counters=None):
        """{keras variable name: float32 array} in Keras layouts -- what `{v.name: v.numpy() for v in model.variables}`
        gives on the reference side."""
        names, counters = self.keras_variable_names(counters)
        p = self._engine.get_params()
        return {kn: p[own] for kn, own in names}, counters

    def load_keras_variables(self, named):
        """Inverse of export_keras_variables for dicts / npz files produced on a TensorFlow box.  Layers are matched per
        type in creation order (numeric suffix), so the absolute counter values of the exporting process do not matter."""
        if isinstance(named, str):
            with np.load(named) as z:
                named = {k: z[k] for k in z.files}
        def split(name):
            layer, leaf = name.split(':')[0].rsplit('/', 1)
            layer = layer.split('/')[-1]
            kind, _, num = layer.rpartition('_')
            if not num.isdigit():
                kind, num = layer, '0'
            return kind, int(num), leaf
        by_kind = {}
        for name, arr in named.items():
            kind, num, leaf = split(name)
            by_kind.setdefault(kind, {}).setdefault(num, {})[leaf] = np.asarray(arr, dtype=np.float32)
        own, _ = self.keras_variable_names()
        seq = {k: [v[n] for n in sorted(v)] for k, v in by_kind.items()}
        params, taken = {}, {}
        for kn, name in own:
            kind, num, leaf = split(kn)
            layers = seq.get(kind, [])
            if num >= len(layers) or leaf not in layers[num]:
                raise ValueError('checkpoint lacks %s (layer %d of type %s)' % (leaf, num, kind))
            params[name] = layers[num][leaf]
        self._engine.set_params(params)

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

    def save_tf_checkpoint(self, path):
        """`<path>.index` + `<path>.data-00000-of-00001` with Keras' object-graph keys (voxvae/tf_checkpoint.py)."""
        from voxvae import tf_checkpoint
        p = self._engine.get_params()
        tf_checkpoint.save_keras_checkpoint(path, {k: p[k] for k in self._engine.param_shapes()})

    def load_weights(self, path):
        """<path>.voxvae.npz (this repo's save_weights) or, when only `<path>.index` exists, a TensorFlow checkpoint written
        by the reference's `save_weights(path)` (nolbo.py:1568-1574), read by voxvae/tf_checkpoint.py."""
        f = path if path.endswith(_WEIGHT_SUFFIX) else path + _WEIGHT_SUFFIX
        if not os.path.exists(f) and os.path.exists(path + '.index'):
            from voxvae import tf_checkpoint
            self._engine.set_params(tf_checkpoint.load_keras_checkpoint(path, self._engine.param_shapes()))
            return
        with np.load(f) as z:
            params = {k.replace('.', '/'): z[k] for k in z.files}
        missing = set(self._engine.param_shapes()) - set(params)
        if missing:
            raise ValueError('%s lacks variables %s' % (f, sorted(missing)))
        self._engine.set_params(params)

    def __call__(self, inputs, training=False):
        """model(x, training) of the reference's callers (nolbo.py:1426 `self._decoder(z, training=True)`, AE3D.py:72-73).
        training=True: BatchNorm normalises with the statistics of this batch and moves its moving statistics (momentum
        0.99), exactly what the Keras layers do under training=True; nothing else changes (no optimiser step)."""
        dev = self._engine.device
        x = as_device_f32(inputs, dev)
        if training:
            from voxvae import train as _train
            if getattr(self, '_fwd_train', None) is None:
                self._fwd_train = (_train.Trainer.forward_only(enc=self._engine) if self._kind == 'encoder'
                                   else _train.Trainer.forward_only(dec=self._engine))
            if self._kind == 'encoder':
                return DeviceArray(self._fwd_train.encoder_training_mode(x))
            eng = self._engine
            y0 = torch.zeros(x.shape[0], eng.D, eng.D, eng.D, 1, dtype=torch.float32, device=dev)
            if not eng.final_sigmoid:
                raise NotImplementedError("decoder(z, training=True) with final_activation != 'sigmoid'")
            probs, _, _ = self._fwd_train.decoder_training_mode(x, y0)
            return DeviceArray(probs)
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

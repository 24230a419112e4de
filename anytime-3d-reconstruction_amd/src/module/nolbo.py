"""Drop-in for the ModelNet model classes of the reference's src/module/nolbo.py:

    nolboSingleObject_modelnet_category_AE    (reference nolbo.py:1206-1385)
    nolboSingleObject_modelnet_category_VAE   (reference nolbo.py:1387-1592)

Same constructor arguments, methods, argument meaning and return tuples; the arithmetic is the MI355X HIP
library.  Host inputs are numpy float32 NDHWC arrays exactly as the reference's data loader produces
(src/dataset_loader/modelnet_dataset.py:83); device-resident torch tensors / DeviceArrays are accepted too and
skip the host->device copy.

The reference draws three random tensors inside these methods (the sampling epsilon, nolbo.py:1470; the
np.random mask, :1475; the prior epsilon, :1508) and the dropout rate/mask (:1423-1425).  They are drawn here
the same way by default (mask through np.random.choice with the same arguments, so seeding np.random
reproduces the reference's mask stream); the keyword-only `_eps`, `_mask`, `_eps2` arguments (an extension)
inject them, which is what makes bit-level parity tests possible.
"""
import ctypes
import os

import numpy as np
import torch

import src.net_core.autoencoder3D as ae3D
import src.net_core.darknet as darknet
import src.net_core.priornet as priornet
from voxvae import engine as _E
from voxvae import lib as _L
from voxvae.tensor import DeviceArray, as_device_f32


def _st():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


class _ModelnetBase(object):
    _variational = True

    @property
    def _latent_dim(self):
        return self._enc_backbone_str['z_category_dim']

    def _buildModel(self):
        print('build Models...')
        self._encoder = ae3D.encoder3D(structure=self._enc_str)
        # ==============set decoder3D
        self._decoder = ae3D.decoder3D(structure=self._dec_str)
        self._enc_eng, self._dec_eng = self._encoder._engine, self._decoder._engine
        self._device = self._enc_eng.device
        self._act_dt = self._dec_eng.dt
        print('done')

    # ---------------------------------------------------------------- device-side building blocks
    def _dev(self, a):
        return as_device_f32(a, self._device)

    def _dev_pair(self, a, b):
        """(input, target) on the device; an autoencoder's caller usually passes the SAME host array twice (the reference's
        scripts feed `output_images = input_images`): it is uploaded once (33.6 MB at 32^3, batch 256)."""
        x = self._dev(a)
        return x, (x if b is a else self._dev(b))

    def _to_act(self, z):
        return z if self._act_dt == _L.VV_F32 else z.to(torch.bfloat16)

    def _encode_latent(self, x, eps=None, want_kl=False):
        """encoder -> (slice | clip | sampling) for the VAE, identity for the AE.  Returns (z, z_act, kl)."""
        enc_out = self._enc_eng.forward(x)
        if not self._variational:
            return enc_out, self._to_act(enc_out), None
        Lz = self._latent_dim
        if enc_out.shape[1] != 2 * Lz:
            raise ValueError('VAE encoder must emit 2*z_category_dim channels, got %d' % enc_out.shape[1])
        eps = torch.randn(x.shape[0], Lz, dtype=torch.float32, device=self._device) if eps is None else self._dev(eps)
        z, z_act, kl, _, _ = _E.reparam_kl(enc_out, eps, Lz, self._act_dt)
        return z, z_act, kl

    def _decode_metrics(self, z_act, target, h1=None):
        out, _, stats, m = self._dec_eng.forward(z_act, target, want_metrics=True, h1=h1)
        return out, stats, m

    def _encode_decode_seed(self, x, eps=None):
        """encoder -> latent -> first decoder layer for the paths that decode the latent unchanged (getEval with
        missing_prob = 0, eval_forward_device): the fused latent tail when it applies, else the split calls.
        Returns (z, z_act, kl, h1 or None)."""
        if _E.latent_tail_supported(self._enc_eng, self._dec_eng, self._variational):
            pos = _E.pos_latent_tail_supported(self._enc_eng, self._dec_eng, self._variational, x.shape[0])
            h = self._enc_eng.forward(x, stop_before_tail=True, stop_before_pos=pos)
            pos = pos and h.shape[1] == 4          # the layer in front of the tail was left to the fused call
            if self._variational:
                Lz = self._latent_dim
                eps = torch.randn(x.shape[0], Lz, dtype=torch.float32, device=self._device) if eps is None else self._dev(eps)
            else:
                eps = None
            z, z_act, kl, _, h1 = _E.latent_tail(self._enc_eng, self._dec_eng, h, eps, self._variational, pos_layer=pos)
            return z, z_act, kl, h1
        z, z_act, kl = self._encode_latent(x, eps)
        return z, z_act, kl, None

    def _category_acc(self, z, cats, onehot, mask=None):
        B, Lz, C = z.shape[0], z.shape[1], cats.shape[0]
        idx = torch.empty(B, dtype=torch.int32, device=self._device)
        _L.call('vv_nearest_category', _L.ptr(z), _L.ptr(mask), _L.ptr(cats), C, _L.ptr(idx), B, Lz, _st())
        acc = None
        if onehot is not None:
            acc = torch.empty(1, dtype=torch.float32, device=self._device)
            _L.call('vv_category_accuracy', _L.ptr(idx), _L.ptr(onehot), C, _L.ptr(acc), B, _st())
        return idx, acc

    def _fit(self, inputs, eps, drop_mask, drop_rate):
        """One optimisation step (GradientTape + Adam.apply_gradients of the reference) through voxvae.train.Trainer."""
        from voxvae import train as _T
        if getattr(self, '_trainer', None) is None:
            import torch.distributed as dist
            world = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
            self._trainer = _T.Trainer(self._enc_eng, self._dec_eng, self._variational, self._learning_rate, world_size=world)
        input_images, output_images = inputs
        x, y = self._dev_pair(input_images, output_images)
        mask, scale = None, 1.0
        if self._dropout:      # reference nolbo.py:1423-1425: rate ~ U[0,1) per step, inverted dropout on z
            rate = float(np.random.rand()) if drop_rate is None else float(drop_rate)
            Lz = self._latent_dim
            if drop_mask is None:
                drop_mask = (np.random.rand(x.shape[0], Lz) >= rate).astype('float32')
            mask, scale = self._dev(drop_mask), 1.0 / (1.0 - rate)
        kl, stats, m = self._trainer.step(x, y, None if eps is None else self._dev(eps), mask, scale)
        return kl, m

    # ---------------------------------------------------------------- public API (reference signatures)
    def getEval(self, inputs, category_vectors=None, training=False, missing_prob=0.0, *, _eps=None, _mask=None, _eps2=None):
        """reference nolbo.py:1449-1528 (VAE) / :1260-1332 (AE): returns the 10-tuple
        (pred, loss_shape, pr, rc, acc_cat, pred_corrected, loss_corrected, pr_corrected, rc_corrected, acc_cat_corrected),
        the last five being 0 when missing_prob == 0.
        Legacy form still used by the reference's train scripts (train_modelnet_category_VAE.py:83-84, body kept as
        the commented block nolbo.py:1530-1555): inputs=(x, y) without category_vectors -> (pred, loss_shape, pr, rc)."""
        if training:
            return self._getEval_training_mode(inputs, category_vectors, missing_prob, _eps, _mask, _eps2)
        if len(inputs) == 2 or category_vectors is None:
            return self._getEval_legacy(inputs, missing_prob, _eps, _mask)
        input_images, output_images, category_list = inputs
        from voxvae.hostio import PackedVoxels as _PV
        if missing_prob == 0.0 and isinstance(input_images, (np.ndarray, _PV)) and isinstance(output_images, (np.ndarray, _PV)):
            out = self._getEval_host_chunked(input_images, output_images, category_list, category_vectors, _eps,
                                             lazy=getattr(self, '_lazy_host', False))
            if out is not None:
                return out
        x, y = self._dev_pair(input_images, output_images)
        onehot = self._dev(category_list)
        cats = self._dev(category_vectors)
        B, Lz, C = x.shape[0], self._latent_dim, cats.shape[0]
        h1 = None
        if missing_prob > 0:
            z, z_act, _ = self._encode_latent(x, _eps)
        else:
            z, z_act, _, h1 = self._encode_decode_seed(x, _eps)
        mask = None
        if missing_prob > 0:
            if _mask is None:   # reference nolbo.py:1475-1476, same RNG call
                _mask = np.reshape(np.random.choice(2, B * Lz, p=[missing_prob, 1. - missing_prob]), [B, Lz]).astype('float32')
            mask = self._dev(_mask)
            zf = torch.empty_like(z)
            zf_act = zf if self._act_dt == _L.VV_F32 else torch.empty(B, Lz, dtype=torch.bfloat16, device=self._device)
            _L.call('vv_latent_mask_fill', _L.ptr(z), _L.ptr(mask), _L.ptr(cats), C, _L.ptr(zf),
                    None if zf_act is zf else _L.ptr(zf_act), self._act_dt, B, Lz, _st())
            z, z_act = zf, zf_act
        _, acc = self._category_acc(z, cats, onehot)
        pred, _, m = self._decode_metrics(z_act, y, h1)
        self._z_category = DeviceArray(z)
        res = (DeviceArray(pred), DeviceArray(m[0]), DeviceArray(m[1]), DeviceArray(m[2]), DeviceArray(acc[0]))
        if missing_prob == 0.0:
            return res + (0, 0, 0, 0, 0)
        idx, _ = self._category_acc(z, cats, None, mask)                          # :1505-1506
        eps2 = torch.randn(B, Lz, dtype=torch.float32, device=self._device) if _eps2 is None else self._dev(_eps2)
        zc = torch.empty_like(z)
        zc_act = zc if self._act_dt == _L.VV_F32 else torch.empty(B, Lz, dtype=torch.bfloat16, device=self._device)
        _L.call('vv_latent_correct', _L.ptr(z), _L.ptr(mask), _L.ptr(cats), _L.ptr(idx), _L.ptr(eps2), _L.ptr(zc),
                None if zc_act is zc else _L.ptr(zc_act), self._act_dt, B, Lz, _st())  # :1507-1510
        _, acc_c = self._category_acc(zc, cats, onehot)                           # :1512-1518
        pred_c, _, mc = self._decode_metrics(zc_act, y)                           # :1520-1527
        self._z_category_corrected = DeviceArray(zc)
        return res + (DeviceArray(pred_c), DeviceArray(mc[0]), DeviceArray(mc[1]), DeviceArray(mc[2]), DeviceArray(acc_c[0]))

    def _getEval_host_chunked(self, input_images, output_images, category_list, category_vectors, _eps, lazy=False):
        """getEval(missing_prob=0) on HOST arrays (the reference's calling convention, test_modelnet_VAE.py:114-130) as a
        pipeline over sample ranges: the upload of chunk k+1 runs under the kernels of chunk k, the download of chunk k's
        prediction (into a recycled pinned block) under the kernels of chunk k+1 -- two streams, this model's one pair of
        engines (per-stream workspaces).  Every kernel of the path treats samples independently with a batch-size-independent
        summation order, so the result is the whole-batch result bit for bit (tests/test_gpu_api.py).  Returns None when the
        batch is too small to split or no pinned block is available (the caller then takes the plain path).
        lazy (voxvae.streams.HostPipeline): nothing here waits for the device -- the prediction's download is enqueued behind the
        kernels and the returned HostPrediction waits for it on first access; ONE pass over the whole batch by default (the overlap
        comes from the NEXT batch, issued by the pipeline on another stream, and whole-batch kernels fill the chip)."""
        from voxvae import hostio as _H
        if getattr(self, '_enc_eng', None) is None or input_images.ndim != 5:
            return None                                  # the image -> 3D model: its inputs are images / head outputs, not voxel grids
        B = int(input_images.shape[0])
        # Two chunks pay when the download is the long pole (float32 probabilities: 33.6 MB at the PCIe rate = 0.60 ms beside 0.49 ms of
        # kernels); with the uint8 occupancy return (0.16 ms) one whole-batch pass is faster than two half-batch ones, whose
        # one-workgroup-per-sample kernels fill half the chip each (profiles/r04_host_chunks.json: 0.94 against 1.05 ms per call)
        nchunk = int(os.environ.get('VV_HOST_CHUNKS', '1' if lazy else ('2' if _H.prediction_host_dtype() == 'float32' else '1')))

        def usable(a):      # a float32 C-contiguous array, or a bit-packed host batch (voxvae/hostio.py: 1 bit per voxel over PCIe)
            return isinstance(a, _H.PackedVoxels) or (a.dtype == np.float32 and a.flags['C_CONTIGUOUS'])

        def upload(a, lo, hi):
            return a.to_device(dev, lo, hi) if isinstance(a, _H.PackedVoxels) else torch.from_numpy(a[lo:hi]).to(dev)

        if (nchunk < 2 and not lazy) or nchunk < 1 or B < 64 * nchunk or not usable(input_images):
            return None
        same = output_images is input_images
        if not same and (not usable(output_images) or tuple(output_images.shape) != tuple(input_images.shape)):
            return None
        pd = _H.prediction_host_dtype()
        host, hview = _H.pinned_array(tuple(input_images.shape), pd)
        if host is None:
            return None
        dev = self._device
        main = torch.cuda.current_stream(dev)
        if lazy and nchunk == 1:
            io_streams = [main]                     # the pipeline gives every batch in flight its own stream: no fork needed
        else:
            if getattr(self, '_io_streams', None) is None or len(self._io_streams) != nchunk:
                self._io_streams = [torch.cuda.Stream(device=dev) for _ in range(nchunk)]
            io_streams = self._io_streams
        Lz = self._latent_dim
        eps = None
        if self._variational:
            eps = torch.randn(B, Lz, dtype=torch.float32, device=dev) if _eps is None else self._dev(_eps)
        onehot, cats = self._dev(category_list), self._dev(category_vectors)
        self._enc_eng.ensure_packed()               # weight packing (first call / after a weight change) stays on the caller's stream
        self._dec_eng.ensure_packed()
        bounds = [B * k // nchunk // 4 * 4 for k in range(nchunk)] + [B]
        zs, stats, preds = [], [], []
        for k, s in enumerate(io_streams):
            lo, hi = bounds[k], bounds[k + 1]
            if s is not main:
                s.wait_stream(main)
            with torch.cuda.stream(s):
                # pageable source, synchronous copy at the PCIe rate.  (Staging the chunk through a pinned block to make the upload
                # asynchronous was measured and is NOT used: an async host -> device copy issued beside running kernels took 10-50x
                # longer on this platform -- profiles/microbench/mb_pin.py: 78 ms against 14 ms for two matmuls with and without it.)
                x = upload(input_images, lo, hi)
                y = x if same else upload(output_images, lo, hi)
                z, z_act, _, h1 = self._encode_decode_seed(x, None if eps is None else eps[lo:hi])
                pred, _, st_ = self._dec_eng.forward(z_act, y, h1=h1)
                hview[lo:hi].copy_(_H.device_prediction_as(pred), non_blocking=True)
            for t in (z, st_, pred):
                t.record_stream(main)
            zs.append(z); stats.append(st_); preds.append(pred)
        for s in io_streams:
            if s is not main:
                main.wait_stream(s)
        z = torch.cat(zs, dim=0)
        m = _E.shape_metrics(torch.cat(stats, dim=0))
        _, acc = self._category_acc(z, cats, onehot)
        self._z_category = DeviceArray(z)
        ready = None
        if lazy:                                    # the caller's stream has joined every chunk stream above: one event covers the downloads
            ready = [torch.cuda.Event()]
            ready[0].record(main)
        else:
            for s in io_streams:                    # the host array is handed out: its downloads must have landed
                s.synchronize()
        return (_H.HostPrediction(host, preds, ready), DeviceArray(m[0]), DeviceArray(m[1]), DeviceArray(m[2]), DeviceArray(acc[0]), 0, 0, 0, 0, 0)

    def _train_helper(self):
        from voxvae import train as _T
        if getattr(self, '_trainer', None) is None:
            import torch.distributed as dist
            world = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
            self._trainer = _T.Trainer(self._enc_eng, self._dec_eng, self._variational, self._learning_rate, world_size=world)
        return self._trainer

    def _getEval_training_mode(self, inputs, category_vectors, missing_prob, _eps, _mask, _eps2):
        """getEval(training=True) (reference nolbo.py:1449, 1463, 1496: `self._encoder(x, training=training)`): the same
        algebra with BatchNorm in training mode -- batch statistics, and the moving statistics move (momentum 0.99) -- and no
        optimisation step.  Both decoder passes run in that mode, as in the reference."""
        tr = self._train_helper()
        if len(inputs) == 2 or category_vectors is None:
            x, y = self._dev_pair(inputs[0], inputs[1])
            zero = None
            if missing_prob > 0:     # legacy body nolbo.py:1544-1548: masked entries become 0 (as in _getEval_legacy)
                Bz, Lz = x.shape[0], self._latent_dim
                if _mask is None:
                    _mask = np.reshape(np.random.choice(2, Bz * Lz, p=[missing_prob, 1. - missing_prob]), [Bz, Lz]).astype('float32')
                mk = self._dev(_mask)
                zero = lambda z: self._zero_masked(z, mk)[0]
            _, _, probs, _, m = tr.forward_training_mode(x, y, None if _eps is None else self._dev(_eps), z_fn=zero)
            return DeviceArray(probs), DeviceArray(m[0]), DeviceArray(m[1]), DeviceArray(m[2])
        input_images, output_images, category_list = inputs
        x, y = self._dev_pair(input_images, output_images)
        onehot = self._dev(category_list)
        cats = self._dev(category_vectors)
        B, Lz, C = x.shape[0], self._latent_dim, cats.shape[0]
        mask = None
        if missing_prob > 0:
            if _mask is None:
                _mask = np.reshape(np.random.choice(2, B * Lz, p=[missing_prob, 1. - missing_prob]), [B, Lz]).astype('float32')
            mask = self._dev(_mask)

        def fill(z):
            if mask is None:
                return z
            zf = torch.empty_like(z)
            _L.call('vv_latent_mask_fill', _L.ptr(z), _L.ptr(mask), _L.ptr(cats), C, _L.ptr(zf), None, _L.VV_F32, B, Lz, _st())
            return zf

        z, _, probs, _, m = tr.forward_training_mode(x, y, None if _eps is None else self._dev(_eps), z_fn=fill)
        _, acc = self._category_acc(z, cats, onehot)
        self._z_category = DeviceArray(z)
        res = (DeviceArray(probs), DeviceArray(m[0]), DeviceArray(m[1]), DeviceArray(m[2]), DeviceArray(acc[0]))
        if missing_prob == 0.0:
            return res + (0, 0, 0, 0, 0)
        idx, _ = self._category_acc(z, cats, None, mask)
        eps2 = torch.randn(B, Lz, dtype=torch.float32, device=self._device) if _eps2 is None else self._dev(_eps2)
        zc = torch.empty_like(z)
        _L.call('vv_latent_correct', _L.ptr(z), _L.ptr(mask), _L.ptr(cats), _L.ptr(idx), _L.ptr(eps2), _L.ptr(zc), None, _L.VV_F32,
                B, Lz, _st())
        _, acc_c = self._category_acc(zc, cats, onehot)
        probs_c, _, mc = tr.decoder_training_mode(zc, y)
        self._z_category_corrected = DeviceArray(zc)
        return res + (DeviceArray(probs_c), DeviceArray(mc[0]), DeviceArray(mc[1]), DeviceArray(mc[2]), DeviceArray(acc_c[0]))

    def _zero_masked(self, z, mask):
        """where(mask == 0, 0, z) (legacy body nolbo.py:1544-1548): the latent_correct kernel with a zero prototype table and
        zero epsilon.  Returns (float32, activation-dtype) copies."""
        B, Lz = z.shape
        zeros = torch.zeros(1, Lz, dtype=torch.float32, device=self._device)
        idx0 = torch.zeros(B, dtype=torch.int32, device=self._device)
        e0 = torch.zeros(B, Lz, dtype=torch.float32, device=self._device)
        zc = torch.empty_like(z)
        zc_act = zc if self._act_dt == _L.VV_F32 else torch.empty(B, Lz, dtype=torch.bfloat16, device=self._device)
        _L.call('vv_latent_correct', _L.ptr(z), _L.ptr(mask), _L.ptr(zeros), _L.ptr(idx0), _L.ptr(e0), _L.ptr(zc),
                None if zc_act is zc else _L.ptr(zc_act), self._act_dt, B, Lz, _st())
        return zc, zc_act

    def _getEval_legacy(self, inputs, missing_prob, _eps, _mask):
        input_images, output_images = inputs[0], inputs[1]
        x, y = self._dev_pair(input_images, output_images)
        z, z_act, _ = self._encode_latent(x, _eps)
        if missing_prob > 0:   # commented body nolbo.py:1544-1548: masked entries become 0
            B, Lz = z.shape
            if _mask is None:
                _mask = np.reshape(np.random.choice(2, B * Lz, p=[missing_prob, 1. - missing_prob]), [B, Lz]).astype('float32')
            _, z_act = self._zero_masked(z, self._dev(_mask))
        pred, _, m = self._decode_metrics(z_act, y)
        return DeviceArray(pred), DeviceArray(m[0]), DeviceArray(m[1]), DeviceArray(m[2])

    def getLatent(self, inputs, *, _eps=None):
        """reference nolbo.py:1557-1566 (VAE: a SAMPLED z) / :1355-1358 (AE: the encoder output) -> numpy [B,L]."""
        z, _, _ = self._encode_latent(self._dev(inputs), _eps)
        return np.array(DeviceArray(z))

    def eval_forward_device(self, x, y, eps=None):
        """Device-resident core of getEval(missing_prob=0) (reference nolbo.py:1463-1501): what bench.py times.
        x, y: float32 CUDA tensors [B,D,D,D,1]; returns (pred, stats [B,4], metrics [4], kl [B] or None), all on device."""
        z, z_act, kl, h1 = self._encode_decode_seed(x, eps)
        pred, stats, m = self._decode_metrics(z_act, y, h1)
        return pred, stats, m, kl

    # ---------------------------------------------------------------- checkpoints (reference nolbo.py:1568-1592)
    def saveEncoder(self, save_path):
        file_name = self._enc_str['name']
        self._encoder.save_weights(os.path.join(save_path, file_name))

    def saveDecoder(self, save_path):
        file_name = self._dec_str['name']
        self._decoder.save_weights(os.path.join(save_path, file_name))

    def saveModel(self, save_path):
        self.saveEncoder(save_path=save_path)
        self.saveDecoder(save_path=save_path)

    def loadEncoder(self, load_path, file_name=None):
        if file_name == None:
            file_name = self._enc_str['name']
        self._encoder.load_weights(os.path.join(load_path, file_name))

    def loadDecoder(self, load_path, file_name=None):
        if file_name == None:
            file_name = self._dec_str['name']
        self._decoder.load_weights(os.path.join(load_path, file_name))

    def loadModel(self, load_path):
        self.loadEncoder(load_path=load_path)
        self.loadDecoder(load_path=load_path)


class nolboSingleObject_modelnet_category_AE(_ModelnetBase):
    """reference nolbo.py:1206-1385."""
    _variational = False

    def __init__(self, nolbo_structure,
                 learning_rate=1e-4,
                 dropout=False):
        self._enc_backbone_str = nolbo_structure
        self._enc_str = nolbo_structure['encoder']
        self._dec_str = nolbo_structure['decoder']
        self._dropout = dropout
        self._learning_rate = learning_rate
        self._buildModel()

    def fit(self, inputs, *, _mask=None, _rate=None):
        """reference nolbo.py:1230-1258 -> (loss_shape, pr, rc)."""
        _, m = self._fit(inputs, None, _mask, _rate)
        return DeviceArray(m[0]), DeviceArray(m[1]), DeviceArray(m[2])


class nolboSingleObject_modelnet_category_VAE(_ModelnetBase):
    """reference nolbo.py:1387-1592."""
    _variational = True

    def __init__(self, nolbo_structure,
                 dropout=False,
                 learning_rate=1e-4):
        self._enc_backbone_str = nolbo_structure
        self._enc_str = nolbo_structure['encoder']
        self._dec_str = nolbo_structure['decoder']
        self._dropout = dropout
        self._learning_rate = learning_rate
        self._buildModel()

    def fit(self, inputs, *, _eps=None, _mask=None, _rate=None):
        """reference nolbo.py:1411-1447 -> (loss_kl, loss_shape, pr, rc)."""
        kl, m = self._fit(inputs, _eps, _mask, _rate)
        return DeviceArray(kl.mean()), DeviceArray(m[0]), DeviceArray(m[1]), DeviceArray(m[2])


class nolboSingleObject_VAE(_ModelnetBase):
    """The reference's image -> 3D model, nolbo.py:750-982 (BASELINE.json configs[2], test_pascal_VAE_dr.py): decoder3D +
    losses + latent masking / prior correction + modality dropout run on the HIP path exactly as in the modelnet classes
    (the reference's getEval bodies are line-for-line the same algorithm, nolbo.py:856-928 vs 1449-1528).

    The 2D image encoder is outside the voxel hot path and runs on stock PyTorch ops:
    `backbone_style=darknet.Darknet19` builds the backbone and, as the reference does (nolbo.py:775-783), a `head2D` on
    top of it from `nolbo_structure['encoder_head']`; any other callable images -> features works as `encoder_backbone`.
    With neither, `input_images` are taken to BE head outputs [B, 2*z_dim] (synthetic head features, SURVEY §8d
    config 3).  fit() (nolbo.py:786-833) trains all three: the decoder step runs on the HIP path and hands back
    d loss / d head-output, which continues through the 2D encoder by torch autograd (f32 models only)."""
    _variational = True

    def __init__(self, nolbo_structure,
                 backbone_style=None, encoder_backbone=None,
                 dropout=False,
                 learning_rate=1e-4):
        self._enc_backbone_str = nolbo_structure['encoder_backbone']
        self._enc_head_str = nolbo_structure.get('encoder_head')
        self._dec_str = nolbo_structure['decoder']
        self._backbone_style = backbone_style
        self._encoder_backbone = encoder_backbone
        self._dropout = dropout
        self._learning_rate = learning_rate
        self._buildModel()

    @property
    def _latent_dim(self):
        return self._enc_backbone_str['z_dim']

    def _buildModel(self):
        print('build Models...')
        # ==============set decoder3D
        self._decoder = ae3D.decoder3D(structure=self._dec_str)
        self._dec_eng = self._decoder._engine
        self._device = self._dec_eng.device
        self._act_dt = self._dec_eng.dt
        if self._encoder_backbone is None and self._backbone_style is not None:
            self._encoder_backbone = self._backbone_style(name=self._enc_backbone_str['name'], device=self._device)
        # ==============set encoder head (nolbo.py:775-783)
        self._encoder_head = None
        if self._encoder_backbone is not None and self._enc_head_str is not None and hasattr(self._encoder_backbone, 'output_shape'):
            self._encoder_head = darknet.head2D(name=self._enc_head_str['name'],
                                                input_shape=self._encoder_backbone.output_shape[1:],
                                                output_dim=self._enc_head_str['output_dim'],
                                                filter_num_list=self._enc_head_str['filter_num_list'],
                                                filter_size_list=self._enc_head_str['filter_size_list'],
                                                last_pooling='max', activation=self._enc_head_str['activation'],
                                                device=self._device)
        self._trainer2d = None
        print('done')

    def _encoder_2d(self, x, training=False):
        if self._encoder_backbone is None:
            return x
        f = self._encoder_backbone(x, training=training) if self._encoder_head is not None else self._encoder_backbone(x)
        return self._encoder_head(f, training=training) if self._encoder_head is not None else f

    def _encode_latent(self, x, eps=None, want_kl=False):
        enc_out = self._dev(self._encoder_2d(x, training=False))
        Lz = self._latent_dim
        if enc_out.dim() != 2 or enc_out.shape[1] != 2 * Lz:
            raise ValueError('the 2D encoder must emit [B, %d] (mean | logVar), got %s' % (2 * Lz, tuple(enc_out.shape)))
        eps = torch.randn(enc_out.shape[0], Lz, dtype=torch.float32, device=self._device) if eps is None else self._dev(eps)
        z, z_act, kl, _, _ = _E.reparam_kl(enc_out, eps, Lz, self._act_dt)
        return z, z_act, kl

    def _encode_decode_seed(self, x, eps=None):
        """No voxel encoder here (the latent comes from the 2D encoder / supplied head outputs): the split calls."""
        z, z_act, kl = self._encode_latent(x, eps)
        return z, z_act, kl, None

    def _getEval_training_mode(self, inputs, category_vectors, missing_prob, _eps, _mask, _eps2):
        """getEval(training=True) of the image -> 3D model (reference nolbo.py:856-928 passes `training` to the backbone, the head
        and the decoder): the 2D encoder runs in training mode (torch modules), the decoder's BatchNorm normalises with batch
        statistics and moves its moving statistics; no optimisation step."""
        from voxvae import train as _T
        if category_vectors is None or len(inputs) != 3:
            raise ValueError('nolboSingleObject_VAE.getEval takes (images, voxels, one-hot) and category_vectors')
        if getattr(self, '_fwd_dec', None) is None:
            self._fwd_dec = _T.Trainer.forward_only(dec=self._dec_eng)
        input_images, output_images, category_list = inputs
        y, onehot, cats = self._dev(output_images), self._dev(category_list), self._dev(category_vectors)
        with torch.no_grad():
            enc_out = self._dev(self._encoder_2d(input_images, training=True))
        B, Lz, C = enc_out.shape[0], self._latent_dim, cats.shape[0]
        eps = torch.randn(B, Lz, dtype=torch.float32, device=self._device) if _eps is None else self._dev(_eps)
        z, _, _, _, _ = _E.reparam_kl(enc_out.contiguous(), eps, Lz, _L.VV_F32)
        mask = None
        if missing_prob > 0:
            if _mask is None:
                _mask = np.reshape(np.random.choice(2, B * Lz, p=[missing_prob, 1. - missing_prob]), [B, Lz]).astype('float32')
            mask = self._dev(_mask)
            zf = torch.empty_like(z)
            _L.call('vv_latent_mask_fill', _L.ptr(z), _L.ptr(mask), _L.ptr(cats), C, _L.ptr(zf), None, _L.VV_F32, B, Lz, _st())
            z = zf
        _, acc = self._category_acc(z, cats, onehot)
        probs, _, m = self._fwd_dec.decoder_training_mode(z, y)
        self._z_category = DeviceArray(z)
        res = (DeviceArray(probs), DeviceArray(m[0]), DeviceArray(m[1]), DeviceArray(m[2]), DeviceArray(acc[0]))
        if missing_prob == 0.0:
            return res + (0, 0, 0, 0, 0)
        idx, _ = self._category_acc(z, cats, None, mask)
        eps2 = torch.randn(B, Lz, dtype=torch.float32, device=self._device) if _eps2 is None else self._dev(_eps2)
        zc = torch.empty_like(z)
        _L.call('vv_latent_correct', _L.ptr(z), _L.ptr(mask), _L.ptr(cats), _L.ptr(idx), _L.ptr(eps2), _L.ptr(zc), None, _L.VV_F32, B, Lz, _st())
        _, acc_c = self._category_acc(zc, cats, onehot)
        probs_c, _, mc = self._fwd_dec.decoder_training_mode(zc, y)
        self._z_category_corrected = DeviceArray(zc)
        return res + (DeviceArray(probs_c), DeviceArray(mc[0]), DeviceArray(mc[1]), DeviceArray(mc[2]), DeviceArray(acc_c[0]))

    def _dev(self, a):
        if callable(getattr(a, 'numpy', None)) and not isinstance(a, (torch.Tensor, DeviceArray)):
            a = a.numpy()
        return as_device_f32(a, self._device)

    def fit(self, inputs, _eps=None, _dropout=None):
        """nolbo.py:786-833: (input_images, output_images) -> (loss_kl, loss_shape, pr, rc).  total = KL + shape + the l2
        terms of the 2D head; Adam(learning_rate) on backbone + head (torch) and decoder (HIP)."""
        from voxvae import train as _T
        input_images, output_images = inputs
        y = self._dev(output_images)
        if self._trainer2d is None:
            self._trainer2d = _T.Trainer(None, self._dec_eng, variational=True, learning_rate=self._learning_rate)
            mods = [m for m in (self._encoder_backbone, self._encoder_head) if isinstance(m, torch.nn.Module)]
            params = [p for m in mods for p in m.parameters()]
            self._opt2d = torch.optim.Adam(params, lr=self._learning_rate, eps=1e-7) if params else None   # Keras epsilon
        trainable_2d = self._opt2d is not None
        enc_out = self._encoder_2d(input_images, training=True) if trainable_2d else self._dev(self._encoder_2d(input_images))
        B, Lz = enc_out.shape[0], self._latent_dim
        eps = torch.randn(B, Lz, dtype=torch.float32, device=self._device) if _eps is None else self._dev(_eps)
        drop_mask, drop_scale = None, 1.0
        if self._dropout:
            rate, keep = _dropout if _dropout is not None else (float(np.random.rand()), None)
            keep = (torch.rand(B, Lz, device=self._device) >= rate).float() if keep is None else self._dev(keep)
            drop_mask, drop_scale = keep, 1.0 / (1.0 - rate)
        kl, stats, metrics, de = self._trainer2d.step_from_latent(enc_out.detach().contiguous(), y, eps, drop_mask, drop_scale,
                                                                   l2=ae3D.L2_REG)      # + decoder.losses, nolbo.py:819-823
        if trainable_2d:
            self._opt2d.zero_grad(set_to_none=True)
            reg = [l for m in (self._encoder_head, self._encoder_backbone) if hasattr(m, 'losses') for l in m.losses]
            enc_out.backward(de, retain_graph=bool(reg))
            if reg:
                torch.stack(reg).sum().backward()
            self._opt2d.step()
        m = metrics.cpu().numpy()
        return float(kl.mean().item()), float(m[0]), float(m[1]), float(m[2])

    def saveEncoderBackbone(self, save_path):
        if hasattr(self._encoder_backbone, 'save_weights'):
            self._encoder_backbone.save_weights(os.path.join(save_path, self._enc_backbone_str['name']))

    def saveEncoderHead(self, save_path):
        if self._encoder_head is not None:
            self._encoder_head.save_weights(os.path.join(save_path, self._enc_head_str['name']))

    def saveEncoder(self, save_path):
        self.saveEncoderBackbone(save_path)
        self.saveEncoderHead(save_path)

    def loadEncoderBackbone(self, load_path, file_name=None):
        if hasattr(self._encoder_backbone, 'load_weights'):
            self._encoder_backbone.load_weights(os.path.join(load_path, file_name or self._enc_backbone_str['name']))

    def loadEncoderHead(self, load_path, file_name=None):
        if self._encoder_head is not None:
            self._encoder_head.load_weights(os.path.join(load_path, file_name or self._enc_head_str['name']))

    def loadEncoder(self, load_path, file_name=None):
        self.loadEncoderBackbone(load_path)
        self.loadEncoderHead(load_path)

    def saveModel(self, save_path):
        self.saveEncoder(save_path=save_path)
        self.saveDecoder(save_path=save_path)

    def loadModel(self, load_path):
        self.loadEncoder(load_path=load_path)
        self.loadDecoder(load_path=load_path)


class nolboSingleObject_modelnet_category_only(_ModelnetBase):
    """reference nolbo.py:1594-1787: the VAE with a learned class-conditional prior.  Encoder, decoder, losses and the
    missing-latent correction run on the HIP path as in nolboSingleObject_modelnet_category_VAE -- getEval is that class's
    getEval with the prototype table replaced by the prior network's means over `category_indices` (nolbo.py:1685-1688).
    The prior network (a 40 -> ... -> L MLP, src/net_core/priornet.py) and the [B, L] latent algebra of fit() -- KL to the
    learned prior, prior / posterior mixing, the pairwise regulariser -- are outside the voxel path (SURVEY §8(f)
    rank 2) and run as torch autograd code between the HIP encoder and decoder (voxvae.train.Trainer.step_custom_latent)."""
    _variational = True

    def __init__(self, nolbo_structure,
                 learning_rate=1e-4):
        self._enc_backbone_str = nolbo_structure
        self._enc_str = nolbo_structure['encoder']
        self._dec_str = nolbo_structure['decoder']
        self._prior_class_str = nolbo_structure['prior_class']
        self._dropout = False
        self._learning_rate = learning_rate
        self._buildModel()
        # ==============set prior network
        self._priornet_class = priornet.priornet(structure=self._prior_class_str, device=self._device)
        self._trainer_c = None

    def fit(self, inputs, dropout=False, *, _rand=None):
        """nolbo.py:1620-1676: (x, y, onehot) -> (loss_kl, loss_shape, loss_reg, pr, rc); total = KL(q || prior) + shape +
        0.01 reg.  `_rand` (tests) = dict(eps, eps_prior, mix (bool), noise [B,L], drop_rate, drop_keep [B,L])."""
        from voxvae import train as _T
        input_images, output_images, category_list = inputs
        x, y = self._dev_pair(input_images, output_images)
        onehot = self._dev(category_list)
        B, Lz = x.shape[0], self._latent_dim
        if self._trainer_c is None:
            self._trainer_c = _T.Trainer(self._enc_eng, self._dec_eng, variational=False, learning_rate=self._learning_rate)
            self._opt_prior = torch.optim.Adam(self._priornet_class.parameters(), lr=self._learning_rate, eps=1e-7)
        r = _rand or {}
        dev = self._device
        eps = self._dev(r['eps']) if 'eps' in r else torch.randn(B, Lz, device=dev)
        eps_p = self._dev(r['eps_prior']) if 'eps_prior' in r else torch.randn(B, Lz, device=dev)
        mix = bool(r['mix']) if 'mix' in r else not (np.random.rand() > 0.5)       # :1642: z itself with probability 1/2
        noise = None
        if mix:
            missing_pr = 0.3
            noise = self._dev(r['noise']) if 'noise' in r else self._dev(
                np.random.choice(a=[True, False], size=(B, Lz), p=[1. - missing_pr, missing_pr]).astype('float32'))
        drop = None
        if dropout:
            rate = float(r.get('drop_rate', np.random.rand()))
            keep = self._dev(r['drop_keep']) if 'drop_keep' in r else (torch.rand(B, Lz, device=dev) >= rate).float()
            drop = (keep, 1.0 / (1.0 - rate))
        dist = 2.0 * Lz

        def latent(enc_out):
            mean_p, lv_p = self._priornet_class(onehot, training=True)
            mean, lv = enc_out[:, :Lz], torch.clamp(enc_out[:, Lz:2 * Lz], -10.0, 10.0)
            z = mean + torch.sqrt(torch.exp(lv)) * eps                               # function.py:35-38
            z_prior = mean_p + torch.sqrt(torch.exp(lv_p)) * eps_p
            z_in = z if noise is None else torch.where(noise == 1., z, z_prior)
            if drop is not None:
                z_in = z_in * drop[0] * drop[1]
            kl = (0.5 * (lv_p - lv) + (torch.exp(lv) + (mean - mean_p) ** 2) / (2.0 * torch.exp(lv_p)) - 0.5).sum(-1).mean()   # function.py:84-98
            d = (torch.abs(mean_p[:, None, :] - mean_p[None, :, :]) / torch.exp(0.5 * lv_p)[:, None, :]).sum(-1) - dist
            reg = torch.where(d > 0, torch.zeros_like(d), d * d).sum(-1).mean()      # function.py:40-71, no class input (:1663)
            return z_in, kl + 0.01 * reg, (kl.detach(), reg.detach())

        self._opt_prior.zero_grad(set_to_none=True)
        stats, m, (kl, reg) = self._trainer_c.step_custom_latent(x, y, latent)
        self._opt_prior.step()
        return DeviceArray(kl), DeviceArray(m[0]), DeviceArray(reg), DeviceArray(m[1]), DeviceArray(m[2])

    def getEval(self, inputs, category_indices=np.identity(40), training=False, missing_prob=0.0, *, _eps=None, _mask=None, _eps2=None):
        """nolbo.py:1678-1754 -> the 10-tuple of the VAE class, classified / corrected against the prior means."""
        # reference :1686, :1691, :1724: `training` goes to the prior network, the encoder and the decoder alike
        mean_prior, _ = self._priornet_class(np.asarray(category_indices, dtype='float32'), training=bool(training))
        return _ModelnetBase.getEval(self, inputs, category_vectors=mean_prior.detach().contiguous(), training=training,
                                     missing_prob=missing_prob, _eps=_eps, _mask=_mask, _eps2=_eps2)

    def savePriorCategory(self, save_path):
        self._priornet_class.save_weights(os.path.join(save_path, self._prior_class_str['name']))

    def loadPriorCategory(self, load_path, file_name=None):
        self._priornet_class.load_weights(os.path.join(load_path, file_name or self._prior_class_str['name']))

    def saveModel(self, save_path):
        self.saveEncoder(save_path=save_path)
        self.saveDecoder(save_path=save_path)
        self.savePriorCategory(save_path=save_path)

    def loadModel(self, load_path):
        self.loadEncoder(load_path=load_path)
        self.loadDecoder(load_path=load_path)
        self.loadPriorCategory(load_path=load_path)

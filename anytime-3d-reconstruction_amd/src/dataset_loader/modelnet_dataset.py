"""Drop-in for the reference's src/dataset_loader/modelnet_dataset.py (dataLoader, lines 5-91): same constructor,
attributes (`epoch`, `batchStart`, `dataLength`) and `getNextBatch(batchSize)` contract -- a dict of float32 numpy
arrays `input_images [B,D,D,D,1]` in {0,1}, `class_list [B,40]` one-hot, `inst_list [B,...]`.

The reference reads `<data_path>/32to64_4rot_64sqr/{train,test}/{i}Full.npy|Class.npy|Inst.npy`; that dataset is
not distributable and does not exist here, so `data_path=None` (or 'synthetic[:N[:D]]') serves a seeded synthetic
set with the same shapes/dtypes (voxvae.synthetic) instead.  Real shards are loaded exactly as the reference does."""
import os
import sys

import numpy as np

from voxvae import synthetic as _syn


class dataLoader(object):
    """Public contract (what the reference's scripts touch): `epoch`, `batchStart`, `dataLength`, `getNextBatch(batchSize)`.
    Internally the split is three arrays plus a visiting order; an epoch ends when the next batch would run past the end,
    and the order is then reshuffled in place with np.random.shuffle (so seeding np.random fixes the sample stream, as in
    the reference)."""

    def __init__(self, data_path, trainortest='train', partial_num=30, synthetic_size=None, voxel=None, classes=40, packed=False):
        """packed=True (extension): the split is kept as 1 bit per voxel and `input_images` is a voxvae.hostio.PackedVoxels -- it
        reads like the float32 array (`np.array()`, `.shape`, indexing) and the model classes move the bits, not the floats, across
        PCIe (4 KiB instead of 128 KiB per 32^3 sample); the grids are binarised at 0.5 as the reference's loader does (:25)."""
        self._packed_mode = bool(packed)
        self.epoch, self.batchStart, self.dataLength = 0, 0, 0
        self._root, self._split, self._shards = data_path, trainortest, partial_num
        self._classes = classes
        self._synthetic = data_path is None or str(data_path).startswith('synthetic')
        self._syn_n, self._syn_d = synthetic_size, voxel
        if self._synthetic and isinstance(data_path, str):
            fields = data_path.split(':')[1:]              # 'synthetic[:N[:D]]'
            if len(fields) > 0 and fields[0]:
                self._syn_n = int(fields[0])
            if len(fields) > 1 and fields[1]:
                self._syn_d = int(fields[1])
        self._grids = self._labels = self._ids = None
        self._order = None
        self._loadData()
        self._dataIdxShuffle()

    def _read_shards(self):
        sub = 'train' if self._split == 'train' else 'test'
        count = self._shards if self._split == 'train' else 5
        folder = os.path.join(self._root, '32to64_4rot_64sqr', sub)
        parts = {'Full': [], 'Class': [], 'Inst': []}
        for i in range(count):
            for kind in parts:
                parts[kind].append(np.load(os.path.join(folder, '%d%s.npy' % (i, kind))))
            sys.stdout.write("%s data:%02d/%02d   \r" % (sub, i + 1, count))
        print('')
        return tuple(np.concatenate(parts[kind], axis=0) for kind in ('Full', 'Class', 'Inst'))

    def _loadData(self):
        print('load data...')
        if self._synthetic:
            n = self._syn_n or (512 if self._split == 'train' else 128)
            d = self._syn_d or 32
            seed = 1234 if self._split == 'train' else 4321
            self._grids = _syn.make_voxels(n, d, seed=seed)
            self._labels = _syn.make_onehot(n, self._classes, seed=seed + 1)
            self._ids = np.arange(n, dtype=np.float32).reshape(n, 1)
        else:
            self._grids, self._labels, self._ids = self._read_shards()
        self.dataLength = len(self._grids)
        self._order = np.arange(self.dataLength)
        if getattr(self, '_packed_mode', False):
            g = np.asarray(self._grids)
            self._sample_shape_host = tuple(g.shape[1:])
            self._bits = np.packbits(g.reshape(len(g), -1) > 0.5, axis=1, bitorder='little')
        print('done!')

    # kept under the reference's name: train scripts call it to restart an epoch
    def _dataIdxShuffle(self):
        np.random.shuffle(self._order)
        self.batchStart = 0

    def getNextBatch(self, batchSize=32):
        if self.batchStart + batchSize > self.dataLength:      # the tail that does not fill a batch is dropped
            self.epoch += 1
            self._dataIdxShuffle()
        rows = self._order[self.batchStart:self.batchStart + batchSize]
        self.batchStart += batchSize
        f32 = np.float32
        if getattr(self, '_packed_mode', False):
            from voxvae.hostio import PackedVoxels
            images = PackedVoxels(self._bits[rows], (len(rows),) + self._sample_shape_host)
        else:
            images = self._grids[rows].astype(f32)
        return {'input_images': images, 'class_list': self._labels[rows].astype(f32), 'inst_list': self._ids[rows].astype(f32)}

    # the names the previous revision (and the deviceDataLoader below) used for the three arrays
    @property
    def _vox3DData(self):
        return self._grids

    @_vox3DData.setter
    def _vox3DData(self, v):
        self._grids = v

    @property
    def _classList(self):
        return self._labels

    @property
    def _instList(self):
        return self._ids


class deviceDataLoader(dataLoader):
    """On-device variant of dataLoader: the whole split lives in HBM as packed bits (32^3 = 4 KiB per
    sample instead of 128 KiB of float32 on the host), the epoch permutation is drawn on the device, and getNextBatch
    gathers + unpacks the batch with one kernel -- no per-iteration host slicing or host-to-device copy
    (reference loop: modelnet_dataset.py:74-91 + the implicit copy at test_modelnet_VAE.py:114-130).

    Same attributes and batch-dict keys as dataLoader; the values are float32 CUDA tensors (what the model classes take
    without a copy), plus 'index_list' (int32, the rows served)."""

    def __init__(self, data_path, trainortest='train', partial_num=30, synthetic_size=None, voxel=None, classes=40,
                 device='cuda:0', seed=None):
        import ctypes
        import torch
        from voxvae import lib as _L
        self._torch, self._L, self._ctypes = torch, _L, ctypes
        self._device = torch.device(device)
        self._gen = torch.Generator(device=self._device)
        if seed is not None:
            self._gen.manual_seed(int(seed))
        super().__init__(data_path, trainortest, partial_num, synthetic_size, voxel, classes)

    def _loadData(self):
        super()._loadData()
        torch = self._torch
        vox = np.asarray(self._vox3DData)
        n = vox.shape[0]
        self._sample_shape = tuple(vox.shape[1:])
        self._voxels = int(np.prod(self._sample_shape))
        if self._voxels % 8:
            raise ValueError('voxel count per sample must be a multiple of 8')
        bits = np.packbits(vox.reshape(n, -1) > 0.5, axis=1, bitorder='little')          # binarised as the reference does (:25)
        self._packed = torch.from_numpy(np.ascontiguousarray(bits)).to(self._device)
        self._classes_dev = torch.from_numpy(np.asarray(self._classList, dtype=np.float32)).to(self._device)
        self._inst_dev = torch.from_numpy(np.asarray(self._instList, dtype=np.float32)).to(self._device)
        self._vox3DData = None                                                            # the float grids are not kept

    def _dataIdxShuffle(self):
        self._perm = self._torch.randperm(self.dataLength, device=self._device, generator=self._gen, dtype=self._torch.int32)
        self.batchStart = 0

    def getNextBatch(self, batchSize=32):
        torch, L = self._torch, self._L
        if self.batchStart + batchSize > self.dataLength:
            self.epoch += 1
            self._dataIdxShuffle()
        idx = self._perm[self.batchStart:self.batchStart + batchSize].contiguous()
        self.batchStart += batchSize
        B = idx.numel()
        out = torch.empty((B,) + self._sample_shape, dtype=torch.float32, device=self._device)
        st = self._ctypes.c_void_p(torch.cuda.current_stream(self._device).cuda_stream)
        L.call('vv_unpack_bits_gather', L.ptr(self._packed), L.ptr(idx), L.ptr(out), B, self._voxels, st)
        li = idx.long()
        return {'input_images': out, 'class_list': self._classes_dev[li], 'inst_list': self._inst_dev[li], 'index_list': idx}

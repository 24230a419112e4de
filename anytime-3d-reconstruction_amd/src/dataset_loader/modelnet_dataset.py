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
    def __init__(self, data_path, trainortest='train', partial_num=30, synthetic_size=None, voxel=None, classes=40):
        self.epoch = 0
        self._data_path = data_path
        self._partial_num = partial_num
        self.batchStart = 0

        self._vox3DData = []
        self._classList = []
        self._instList = []
        self.dataLength = 0
        self._dataIdx = None
        self._trainortest = trainortest
        self._synthetic = data_path is None or str(data_path).startswith('synthetic')
        self._syn_n, self._syn_d, self._classes = synthetic_size, voxel, classes
        if self._synthetic and isinstance(data_path, str) and ':' in data_path:
            parts = data_path.split(':')
            self._syn_n = int(parts[1]) if len(parts) > 1 and parts[1] else synthetic_size
            self._syn_d = int(parts[2]) if len(parts) > 2 and parts[2] else voxel

        self._loadData()
        self._dataIdxShuffle()

    def _loadData(self):
        print('load data...')
        if self._synthetic:
            n = self._syn_n or (512 if self._trainortest == 'train' else 128)
            d = self._syn_d or 32
            seed = 1234 if self._trainortest == 'train' else 4321
            self._vox3DData = _syn.make_voxels(n, d, seed=seed)
            self._classList = _syn.make_onehot(n, self._classes, seed=seed + 1)
            self._instList = np.arange(n, dtype=np.float32).reshape(n, 1)
        else:
            sub = 'train' if self._trainortest == 'train' else 'test'
            count = self._partial_num if self._trainortest == 'train' else 5
            vox, cls, inst = [], [], []
            for i in range(count):
                base = os.path.join(self._data_path, '32to64_4rot_64sqr', sub, str(i))
                vox.append(np.load(base + 'Full.npy'))
                cls.append(np.load(base + 'Class.npy'))
                inst.append(np.load(base + 'Inst.npy'))
                sys.stdout.write("%s data:%02d/%02d   \r" % (sub, i + 1, count))
            print('')
            self._vox3DData = np.concatenate(vox, axis=0)
            self._classList = np.concatenate(cls, axis=0)
            self._instList = np.concatenate(inst, axis=0)
        self.dataLength = len(self._vox3DData)
        self._dataIdx = [i for i in range(self.dataLength)]
        print('done!')

    def _dataIdxShuffle(self):
        np.random.shuffle(self._dataIdx)
        self.batchStart = 0

    def getNextBatch(self, batchSize=32):
        if self.batchStart + batchSize > self.dataLength:
            self.epoch += 1
            self._dataIdxShuffle()
        dataStart = self.batchStart
        dataEnd = self.batchStart + batchSize
        self.batchStart += batchSize
        dataList = self._dataIdx[dataStart:dataEnd]
        batch_dict = {
            'input_images': (self._vox3DData[dataList]).astype('float32'),
            'class_list': (self._classList[dataList]).astype('float32'),
            'inst_list': (self._instList[dataList]).astype('float32'),
        }
        return batch_dict


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

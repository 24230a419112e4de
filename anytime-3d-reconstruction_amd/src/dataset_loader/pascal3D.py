"""Pascal3D+ single-object loader, mirror of the reference's src/dataset_loader/pascal3D.py:56-283 (interface only).

`dataLoaderSingleObject(trainOrVal, Pascal3DDataPath)` with `getNextBatch(batchSizeof3DShape, imageSize, augmentation)`
returning `(instList, classList, sin, cos, inputImages, outputImages)` and the counters the scripts print (`epoch`,
`dataStart`, `dataLength`).  The dataset itself (images + CAD voxelisations, read with cv2 in the reference) does not
exist here, so `Pascal3DDataPath=None` or `'synthetic:N:D'` serves seeded stand-ins of the same shapes and dtypes:
12 classes (the Pascal3D+ categories), D^3 occupancy grids (default 64, the reference's CAD grid), and an image whose
three channels are the max-projections of the object along the three axes, resampled to `imageSize` -- so a 2D encoder
has something learnable to look at."""
import numpy as np

from voxvae import synthetic as syn

CLASSES = 12


class dataLoaderSingleObject(object):
    def __init__(self, trainOrVal='train', Pascal3DDataPath=None, voxel=64):
        n = 256 if trainOrVal == 'train' else 96
        if isinstance(Pascal3DDataPath, str) and Pascal3DDataPath.startswith('synthetic:'):
            parts = Pascal3DDataPath.split(':')
            n = int(parts[1])
            voxel = int(parts[2]) if len(parts) > 2 else voxel
        elif Pascal3DDataPath is not None:
            raise FileNotFoundError('Pascal3D+ is not available in this build; pass None or synthetic:N:D')
        seed = 4321 if trainOrVal == 'train' else 8765
        self._rng = np.random.default_rng(seed)
        self._voxels = syn.make_voxels(n, voxel, seed=seed)                  # [n,D,D,D,1] in {0,1}
        self._classes = self._rng.integers(0, CLASSES, n)
        self._euler = self._rng.uniform(-np.pi, np.pi, (n, 3)).astype('float32')
        self._order = np.arange(n)
        self.epoch = 0
        self.dataStart = 0
        self.dataLength = n
        self._shuffle = trainOrVal == 'train'
        if self._shuffle:
            self._rng.shuffle(self._order)

    @staticmethod
    def _project(v, size):
        col, row = size                                                      # the reference passes (image_col, image_row)
        g = v[..., 0]
        chans = []
        for ax in range(3):
            p = g.max(axis=ax)                                               # [D,D]
            ri = (np.arange(row) * p.shape[0] // row).clip(0, p.shape[0] - 1)
            ci = (np.arange(col) * p.shape[1] // col).clip(0, p.shape[1] - 1)
            chans.append(p[ri][:, ci])
        return np.stack(chans, axis=-1).astype('float32')                    # [row, col, 3]

    def getNextBatch(self, batchSizeof3DShape=32, imageSize=None, augmentation=True):
        size = imageSize or (256, 256)
        idx = self._order[self.dataStart:self.dataStart + batchSizeof3DShape]
        self.dataStart += len(idx)
        if self.dataStart >= self.dataLength:
            self.epoch += 1
            self.dataStart = 0
            if self._shuffle:
                self._rng.shuffle(self._order)
        out = self._voxels[idx]
        imgs = np.stack([self._project(v, size) for v in out])
        if augmentation:
            imgs = np.clip(imgs + self._rng.normal(0, 0.05, imgs.shape).astype('float32'), 0, 1)
        cls = np.eye(CLASSES, dtype='float32')[self._classes[idx]]
        inst = np.zeros((len(idx), 1), dtype='float32')
        e = self._euler[idx]
        return inst, cls, np.sin(e), np.cos(e), imgs.astype('float32'), out.astype('float32')

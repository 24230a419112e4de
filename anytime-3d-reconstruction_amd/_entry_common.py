"""Shared by the train_*/test_* entry points: the config dict literals of the reference scripts
(test_modelnet_VAE.py:169-192, train_modelnet_category_VAE.py:109-132) with the voxel side as a parameter, and the
argparse overrides SURVEY §5 asks for (--voxel, --batch, --dtype, --synthetic ...)."""
import argparse


def make_config(latent_dim=64, voxel=64, variational=True):
    return {
        'z_category_dim': latent_dim,
        'encoder': {
            'name': 'encoder3D',
            'input_shape': [voxel, voxel, voxel, 1],  # or [None,None,None,1]
            'filter_num_list': [64, 128, 256, 512, (2 if variational else 1) * latent_dim],
            'filter_size_list': [4, 4, 4, 4, 4],
            'strides_list': [2, 2, 2, 2, 1],
            'final_pool': 'average',
            'activation': 'elu',
            'final_activation': 'None',
        },
        'decoder': {
            'name': 'decoder',
            'input_dim': latent_dim,
            'output_shape': [voxel, voxel, voxel, 1],
            'filter_num_list': [512, 256, 128, 64, 1],
            'filter_size_list': [4, 4, 4, 4, 4],
            'strides_list': [1, 2, 2, 2, 2],
            'activation': 'elu',
            'final_activation': 'sigmoid'
        },
    }


def parse(description, train=False):
    ap = argparse.ArgumentParser(description=description)
    ap.add_argument('--dataset-path', default=None, help="ModelNet shard directory; default: seeded synthetic voxels")
    ap.add_argument('--voxel', type=int, default=32, help='voxel side (the reference ships 64; BASELINE.json asks for 32)')
    ap.add_argument('--latent', type=int, default=64)
    ap.add_argument('--batch', type=int, default=64 if train else 72)
    ap.add_argument('--dtype', default='f32', choices=['f32', 'bf16'] + ([] if train else ['fp8']),
                    help="'f32': the reference's arithmetic; 'bf16'; 'fp8' (test_* scripts only: e4m3fn MFMA layers, inference)")
    ap.add_argument('--load-path', default=None)
    ap.add_argument('--save-path', default=None)
    ap.add_argument('--max-iter', type=int, default=None, help='stop after this many iterations (smoke runs)')
    ap.add_argument('--missing-pr', type=float, default=0.9)
    ap.add_argument('--epochs', type=int, default=1000)
    ap.add_argument('--lr', type=float, default=1e-4)
    ap.add_argument('--device-data', action='store_true', help='keep the split in HBM as packed bits (deviceDataLoader)')
    ap.add_argument('--packed-data', action='store_true', help='host loader that keeps the split as bits and serves PackedVoxels batches (1 bit per voxel over PCIe)')
    ap.add_argument('--pipeline', type=int, default=1, help='test_modelnet_VAE.py: batches in flight (voxvae.streams.HostPipeline); 1 = the reference\'s synchronous loop')
    ap.add_argument('--dump-dir', default=None, help='test_modelnet_VAE.py: save <missing_pr>_cl_label/_gt/_pred.npy here (reference :159-165)')
    return ap.parse_args()


# ---------------------------------------------------------------------------------------------------------------------
# Shared loop plumbing of the train_* / test_* entry points.  The reference scripts each carry their own copy of the
# same bookkeeping (running means reset at epoch boundaries, one status line rewritten in place, a NaN stop); here it
# lives once.  The printed fields and their order are those of the reference scripts.
import sys
import time

import numpy as np


class RunningMeans:
    """Per-epoch running means of several result tuples (the `loss = (loss * it + new) / (it + 1)` idiom)."""

    def __init__(self, **widths):
        self.n = 0
        self.sums = {k: np.zeros(w) for k, w in widths.items()}

    def reset(self):
        self.n = 0
        for v in self.sums.values():
            v[:] = 0

    def add(self, **values):
        for k, v in values.items():
            self.sums[k] += np.array([float(x) for x in v])
        self.n += 1

    def __getitem__(self, key):
        return self.sums[key] / max(self.n, 1)

    def finite(self):
        return all(np.all(np.isfinite(v)) for v in self.sums.values())


class Progress:
    """One carriage-return status line: iteration, mean step time, epoch, loader position, then labelled groups."""

    def __init__(self, width=4):
        self.pos_fmt = "cur_o/tot_o:{:0%dd}/{:0%dd} " % (width, width)
        self.elapsed, self.n = 0.0, 0
        self._t0 = None

    def reset(self):
        self.elapsed, self.n = 0.0, 0

    def tic(self):
        self._t0 = time.time()

    def toc(self):
        self.elapsed += time.time() - self._t0
        self.n += 1

    @staticmethod
    def group(pairs, sep=", "):
        return sep.join("%s:%.4f" % (k, v) for k, v in pairs)

    def show(self, epoch, position, total, *groups):
        head = "it:{:04d} rt:{:.2f} Ep_o:{:03d} ".format(self.n, self.elapsed / max(self.n, 1), int(epoch) + 1)
        sys.stdout.write(head + self.pos_fmt.format(position, total) + " ".join(groups) + "  \r")
        sys.stdout.flush()


def epochs_of(loader, limit, position_attr, on_new_epoch=None):
    """Yields (epoch, position, total) before every batch until `limit` epochs have been served; calls `on_new_epoch()`
    when the loader's epoch counter has advanced (the scripts save the model and restart their running means there)."""
    epoch, first = loader.epoch, True
    while loader.epoch < limit:
        if loader.epoch != epoch and not first:
            epoch = loader.epoch
            if on_new_epoch is not None:
                on_new_epoch()
        first = False
        yield loader.epoch, getattr(loader, position_attr), loader.dataLength


def stop_on_nan(means):
    if means.finite():
        return False
    print('')
    print('NaN')
    return True

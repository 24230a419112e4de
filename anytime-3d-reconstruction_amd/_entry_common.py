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
    ap.add_argument('--dtype', default='f32', choices=['f32', 'bf16'])
    ap.add_argument('--load-path', default=None)
    ap.add_argument('--save-path', default=None)
    ap.add_argument('--max-iter', type=int, default=None, help='stop after this many iterations (smoke runs)')
    ap.add_argument('--missing-pr', type=float, default=0.9)
    ap.add_argument('--epochs', type=int, default=1000)
    ap.add_argument('--lr', type=float, default=1e-4)
    ap.add_argument('--device-data', action='store_true', help='keep the split in HBM as packed bits (deviceDataLoader)')
    return ap.parse_args()

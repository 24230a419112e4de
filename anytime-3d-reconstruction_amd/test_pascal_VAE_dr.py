"""Entry point mirroring the reference's test_pascal_VAE_dr.py (loop :80-144, config :186-212): one epoch of
nolboSingleObject_VAE.getEval over the validation split -- Darknet19 + head2D image encoder (stock PyTorch ops),
latent 16, 64^3 decoder, missing-latent correction against the class prototypes.
`python test_pascal_VAE_dr.py --batch 16 --image 128 --missing-pr 0.9 --max-iter 2`."""
import os
import sys

import numpy as np

import _entry_common as C
import voxvae
import src.dataset_loader.pascal3D as pascal3D
import src.net_core.darknet as Darknet


def make_config(latent_dim=16, voxel=64):
    return {
        'encoder_backbone': {'name': 'nolbo_backbone', 'z_dim': latent_dim, 'activation': 'elu'},
        'encoder_head': {'name': 'nolbo_head', 'output_dim': 2 * latent_dim, 'filter_num_list': [], 'filter_size_list': [],
                         'activation': 'elu'},
        'decoder': C.make_config(latent_dim, voxel, True)['decoder'],
    }


class _PascalBatches:
    """Adapts the Pascal loader (positional tuple, `dataStart`) to the dict / `batchStart` interface of the shared loop."""

    def __init__(self, loader, image_size):
        self._l, self._size = loader, image_size

    epoch = property(lambda self: self._l.epoch)
    batchStart = property(lambda self: self._l.dataStart)
    dataLength = property(lambda self: self._l.dataLength)

    def getNextBatch(self, batchSize):
        _, classes, _, _, images, voxels = self._l.getNextBatch(batchSizeof3DShape=batchSize, imageSize=self._size, augmentation=False)
        return {'input_images': images, 'output_images': voxels, 'class_list': classes}


def train(
        learning_rate=1e-4,
        config=None,
        load_path=None,
        load_encoder_backbone_path=None, load_encoder_backbone_name=None,
        load_decoder_path=None, load_decoder_name=None,
        missing_pr=0.3,
        learn='train', batch_size=72, image_size=(256, 256), max_iter=None, dataset_path=None,
):
    import src.module.nolbo as nolbo
    model = nolbo.nolboSingleObject_VAE(nolbo_structure=config, backbone_style=Darknet.Darknet19, learning_rate=learning_rate)
    voxel = config['decoder']['output_shape'][0]
    loader = _PascalBatches(pascal3D.dataLoaderSingleObject(trainOrVal='val', Pascal3DDataPath=dataset_path, voxel=voxel), image_size)
    category_vectors = None
    if load_path is not None:
        print('load weights...')
        model.loadModel(load_path=load_path)
        print('done!')
        cv = os.path.join(load_path, 'category_vectors.npy')
        if os.path.exists(cv):
            category_vectors = np.load(cv).astype('float32')
    if load_encoder_backbone_path is not None:
        model.loadEncoderBackbone(load_path=load_encoder_backbone_path, file_name=load_encoder_backbone_name)
    if load_decoder_path is not None:
        model.loadDecoder(load_path=load_decoder_path, file_name=load_decoder_name)
    if category_vectors is None:
        from voxvae import synthetic as syn
        category_vectors = syn.make_category_vectors(pascal3D.CLASSES, config['encoder_backbone']['z_dim'])

    means = C.RunningMeans(eval=8)
    bar = C.Progress(width=5)
    print('start training...')
    for epoch, position, total in C.epochs_of(loader, 1, 'batchStart'):
        bar.tic()
        batch = loader.getNextBatch(batch_size)
        out = model.getEval(inputs=(batch['input_images'], batch['output_images'], batch['class_list']),
                            category_vectors=category_vectors, missing_prob=missing_pr)
        means.add(eval=out[1:5] + out[6:10])
        bar.toc()
        m = means['eval']
        bar.show(epoch, position, total, bar.group(zip(('loss', 'pr', 'rc', 'c'), m[:4])) + ",",
                 bar.group(zip(('closs', 'cpr', 'crc', 'cc'), m[4:])))
        if C.stop_on_nan(means):
            return None
        if max_iter is not None and means.n >= max_iter:
            break
    print('')
    return means['eval']


latent_dim = 16
config = make_config(latent_dim, 64)

if __name__ == '__main__':
    import argparse
    ap = argparse.ArgumentParser(description=__doc__)
    ap.add_argument('--voxel', type=int, default=64)
    ap.add_argument('--latent', type=int, default=16)
    ap.add_argument('--batch', type=int, default=72)
    ap.add_argument('--image', type=int, default=256)
    ap.add_argument('--dtype', default='f32', choices=['f32', 'bf16'])
    ap.add_argument('--load-path', default=None)
    ap.add_argument('--missing-pr', type=float, default=0.3)
    ap.add_argument('--max-iter', type=int, default=None)
    ap.add_argument('--dataset-path', default=None)
    a = ap.parse_args()
    voxvae.set_default_dtype(a.dtype)
    sys.exit(0 if train(config=make_config(a.latent, a.voxel), load_path=a.load_path, missing_pr=a.missing_pr, batch_size=a.batch,
                        image_size=(a.image, a.image), max_iter=a.max_iter, dataset_path=a.dataset_path) is not None else 1)

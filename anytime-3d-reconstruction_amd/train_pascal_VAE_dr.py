"""Entry point mirroring the reference's train_pascal_VAE_dr.py (loop :119-170): nolboSingleObject_VAE with
dropout=True (random-rate latent dropout, nolbo.py:801-803) trained on (image, voxel) pairs; every iteration also
evaluates the current train and validation batches with the legacy getEval(inputs)[1:] form.
`python train_pascal_VAE_dr.py --batch 8 --image 128 --voxel 32 --max-iter 3`."""
import sys

import _entry_common as C
import voxvae
import src.dataset_loader.pascal3D as pascal3D
import src.net_core.darknet as Darknet
from test_pascal_VAE_dr import make_config


def train(
        training_epoch=1000,
        learning_rate=1e-4,
        config=None,
        save_path=None, load_path=None,
        load_encoder_backbone_path=None, load_encoder_backbone_name=None,
        load_decoder_path=None, load_decoder_name=None,
        batch_size=72, image_size=(256, 256), max_iter=None, dataset_path=None,
):
    import src.module.nolbo as nolbo
    model = nolbo.nolboSingleObject_VAE(nolbo_structure=config, backbone_style=Darknet.Darknet19, learning_rate=learning_rate,
                                        dropout=True)
    voxel = config['decoder']['output_shape'][0]
    loaders = {split: pascal3D.dataLoaderSingleObject(trainOrVal=split, Pascal3DDataPath=dataset_path, voxel=voxel)
               for split in ('train', 'val')}
    if load_path is not None:
        print('load weights...')
        model.loadModel(load_path=load_path)
        print('done!')
    if load_encoder_backbone_path is not None:
        model.loadEncoderBackbone(load_path=load_encoder_backbone_path, file_name=load_encoder_backbone_name)
    if load_decoder_path is not None:
        model.loadDecoder(load_path=load_decoder_path, file_name=load_decoder_name)

    means = C.RunningMeans(fit=4, train=3, test=3)
    bar = C.Progress()

    def new_epoch():
        print('')
        means.reset()
        bar.reset()
        if save_path is not None:
            print('save model...')
            model.saveModel(save_path=save_path)

    print('start training...')
    done = 0
    for epoch, position, total in C.epochs_of(loaders['train'], training_epoch, 'dataStart', new_epoch):
        bar.tic()
        batch = loaders['train'].getNextBatch(batchSizeof3DShape=batch_size, imageSize=image_size)
        batch_val = loaders['val'].getNextBatch(batchSizeof3DShape=batch_size, imageSize=image_size, augmentation=False)
        pair, pair_val = (batch[4], batch[5]), (batch_val[4], batch_val[5])          # (input_images, output_images)
        fit = model.fit(inputs=pair)
        means.add(fit=fit, train=model.getEval(inputs=pair)[1:], test=model.getEval(inputs=pair_val)[1:])
        bar.toc()
        tr, te = means['train'], means['test']
        bar.show(epoch, position, total, bar.group([('kl', means['fit'][0]), ('shape', tr[0]), ('pr', tr[1]), ('rc', tr[2])]),
                 bar.group([('shape', te[0]), ('pr', te[1]), ('rc', te[2])]))
        if C.stop_on_nan(means):
            return None
        done += 1
        if max_iter is not None and done >= max_iter:
            break
    print('')
    return means['fit'], means['train'], means['test']


if __name__ == '__main__':
    import argparse
    ap = argparse.ArgumentParser(description=__doc__)
    ap.add_argument('--voxel', type=int, default=64)
    ap.add_argument('--latent', type=int, default=16)
    ap.add_argument('--batch', type=int, default=72)
    ap.add_argument('--image', type=int, default=256)
    ap.add_argument('--save-path', default=None)
    ap.add_argument('--load-path', default=None)
    ap.add_argument('--max-iter', type=int, default=None)
    ap.add_argument('--epochs', type=int, default=1000)
    ap.add_argument('--lr', type=float, default=1e-4)
    ap.add_argument('--dataset-path', default=None)
    a = ap.parse_args()
    voxvae.set_default_dtype('f32')          # fit() runs the exact-f32 path this round
    sys.exit(0 if train(training_epoch=a.epochs, learning_rate=a.lr, config=make_config(a.latent, a.voxel), save_path=a.save_path,
                        load_path=a.load_path, batch_size=a.batch, image_size=(a.image, a.image), max_iter=a.max_iter,
                        dataset_path=a.dataset_path) is not None else 1)

"""Entry point with the role of the reference's train_modelnet_category_VAE.py (control flow :22-107, config :109-132,
printed fields :92-100) on the MI355X path; `model_class='AE'` gives train_modelnet_category_AE.py, `dropout=True` the
_dr variants.  `python train_modelnet_category_VAE.py --voxel 32 --batch 64 --dtype bf16`."""
import sys

import _entry_common as C
import voxvae
from src.dataset_loader.modelnet_dataset import dataLoader


def train(
        training_epoch=1000,
        learning_rate=1e-4, batch_size=32,
        config=None,
        dataset_path=None,
        save_path=None, load_path=None,
        load_decoder_path=None, load_decoder_name=None,
        max_iter=None, model_class='VAE', dropout=False,
):
    import src.module.nolbo as nolbo
    variational = model_class == 'VAE'
    cls = nolbo.nolboSingleObject_modelnet_category_VAE if variational else nolbo.nolboSingleObject_modelnet_category_AE
    model = cls(nolbo_structure=config, learning_rate=learning_rate, dropout=dropout)
    voxel = config['encoder']['input_shape'][0]
    loaders = {split: dataLoader(data_path=dataset_path, trainortest=split, voxel=voxel) for split in ('train', 'test')}
    if load_path is not None:
        print('load weights...')
        model.loadModel(load_path=load_path)
        print('done!')
    if load_decoder_path is not None:
        print('load decoder weights...')
        model.loadDecoder(load_path=load_decoder_path, file_name=load_decoder_name)
        print('done!')

    means = C.RunningMeans(fit=4 if variational else 3, train=3, test=3)
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
    for epoch, position, total in C.epochs_of(loaders['train'], training_epoch, 'batchStart', new_epoch):
        bar.tic()
        x = loaders['train'].getNextBatch(batchSize=batch_size)['input_images']
        x_test = loaders['test'].getNextBatch(batchSize=batch_size)['input_images']
        fit = model.fit(inputs=(x, x))
        means.add(fit=fit, train=model.getEval(inputs=(x, x))[1:], test=model.getEval(inputs=(x_test, x_test))[1:])
        bar.toc()
        tr, te = means['train'], means['test']
        first = ([('kl', means['fit'][0])] if variational else []) + [('shape', tr[0]), ('pr', tr[1]), ('rc', tr[2])]
        bar.show(epoch, position, total, bar.group(first), bar.group([('shape', te[0]), ('pr', te[1]), ('rc', te[2])]))
        if C.stop_on_nan(means):
            return None
        done += 1
        if max_iter is not None and done >= max_iter:
            break
    print('')
    if save_path is not None:
        model.saveModel(save_path=save_path)
    return means['fit'], means['train'], means['test']


latent_dim = 64
config = C.make_config(latent_dim, 64, True)   # the reference literal (64^3); main() rebuilds it for --voxel

if __name__ == '__main__':
    a = C.parse(__doc__, train=True)
    voxvae.set_default_dtype(a.dtype)        # 'f32': exact-f32 parity mode; 'bf16': mixed precision (float32 master weights)
    sys.exit(0 if train(
        training_epoch=a.epochs, learning_rate=a.lr, batch_size=a.batch,
        config=C.make_config(a.latent, a.voxel, True),
        dataset_path=a.dataset_path,
        save_path=a.save_path, load_path=a.load_path, max_iter=a.max_iter,
    ) is not None else 1)

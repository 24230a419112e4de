"""Entry point with the role of the reference's train_modelnet_category.py (loop :47-107, config :109-139): the VAE with
a learned class-conditional prior (nolboSingleObject_modelnet_category_only); `dropout=True` is train_modelnet_category_dr.py.
`python train_modelnet_category.py --voxel 32 --batch 64 --max-iter 10`."""
import sys

import _entry_common as C
import voxvae
from src.dataset_loader.modelnet_dataset import dataLoader


def make_config(latent_dim=64, voxel=64):
    cfg = C.make_config(latent_dim, voxel, True)
    cfg['prior_class'] = {
        'name': 'priornet_class',
        'input_dim': 40,  # class num (one-hot vector)
        'unit_num_list': [32, latent_dim],
        'core_activation': 'elu',
        'const_log_var': 0.0,
    }
    return cfg


def train(
        training_epoch=1000,
        learning_rate=1e-4, batch_size=32,
        config=None, dataset_path=None,
        save_path=None, load_path=None,
        max_iter=None, dropout=False,
):
    import src.module.nolbo as nolbo
    model = nolbo.nolboSingleObject_modelnet_category_only(nolbo_structure=config, learning_rate=learning_rate)
    voxel = config['encoder']['input_shape'][0]
    loaders = {split: dataLoader(data_path=dataset_path, trainortest=split, voxel=voxel) for split in ('train', 'test')}
    if load_path is not None:
        print('load weights...')
        model.loadModel(load_path=load_path)
        print('done!')

    means = C.RunningMeans(fit=5, train=4, test=4)
    bar = C.Progress()

    def new_epoch():
        print('')
        means.reset()
        bar.reset()
        if save_path is not None:
            print('save model...')
            model.saveModel(save_path=save_path)

    def triple(batch):
        return batch['input_images'], batch['input_images'], batch['class_list']

    print('start training...')
    done = 0
    for epoch, position, total in C.epochs_of(loaders['train'], training_epoch, 'batchStart', new_epoch):
        bar.tic()
        inputs = triple(loaders['train'].getNextBatch(batchSize=batch_size))
        inputs_test = triple(loaders['test'].getNextBatch(batchSize=batch_size))
        fit = model.fit(inputs=inputs, dropout=dropout)
        means.add(fit=fit, train=model.getEval(inputs=inputs)[1:5], test=model.getEval(inputs=inputs_test)[1:5])
        bar.toc()
        f, tr, te = means['fit'], means['train'], means['test']
        bar.show(epoch, position, total,
                 bar.group([('kl', f[0]), ('shape', f[1]), ('reg', f[2]), ('pr', f[3]), ('rc', f[4]), ('c', tr[3])]),
                 bar.group([('shape', te[0]), ('pr', te[1]), ('rc', te[2]), ('c', te[3])]))
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
config = make_config(latent_dim, 64)

if __name__ == '__main__':
    a = C.parse(__doc__, train=True)
    voxvae.set_default_dtype(a.dtype)        # 'f32': exact-f32 parity mode; 'bf16': mixed precision (float32 master weights)
    sys.exit(0 if train(training_epoch=a.epochs, learning_rate=a.lr, batch_size=a.batch, config=make_config(a.latent, a.voxel),
                        dataset_path=a.dataset_path, save_path=a.save_path, load_path=a.load_path, max_iter=a.max_iter) is not None else 1)

"""Entry point with the role of the reference's test_modelnet_VAE.py (loop :104-156, printed fields :141-150, config
:169-192): one epoch of getEval over the test split with missing-latent correction.
`python test_modelnet_VAE.py --voxel 32 --batch 256 --dtype bf16 --missing-pr 0.9 [--device-data] [--pipeline 3] [--dump-dir DIR]`.
--dump-dir writes the three arrays the reference collects per batch (:128-130) and saves at the end of the epoch (:159-165, the
input of the notebooks' precision / recall tool): `<missing_pr>_cl_label.npy`, `<missing_pr>_gt.npy`, `<missing_pr>_pred.npy`."""
import os
import sys

import numpy as np

import _entry_common as C
import voxvae
from src.dataset_loader.modelnet_dataset import dataLoader, deviceDataLoader

FIELDS = ('loss', 'pr', 'rc', 'c'), ('closs', 'cpr', 'crc', 'cc')


def _host(a):
    """np.array() of a batch member: host arrays as they are, device-resident batches (--device-data) copied back."""
    return a.detach().cpu().numpy() if hasattr(a, 'detach') else np.array(a)


def evaluate(model, loader, missing_pr, batch_size, max_iter, class_key='class_list', dump_dir=None, pipeline=1, **extra):
    """The shared test loop: getEval over one epoch, running means of the 8 reported numbers.
    dump_dir: also keep every batch's labels, targets and predictions and save them as the reference's
    `<missing_pr>_cl_label.npy / _gt.npy / _pred.npy` (test_modelnet_VAE.py:128-130, 159-165).
    pipeline > 1: that many batches in flight (voxvae.streams.HostPipeline: batch k is converted and accounted while the kernels and
    copies of batches k + 1 .. are running); the numbers and their order are those of the synchronous loop."""
    import collections
    means = C.RunningMeans(eval=8)
    labels, gts, preds = [], [], []
    bar = C.Progress(width=5)
    pipe, pending = None, collections.deque()
    if pipeline > 1:
        from voxvae.streams import HostPipeline
        pipe = HostPipeline(model, depth=pipeline)
    state = {'stop': False}

    def account(batch, x, out, epoch, position, total):
        means.add(eval=out[1:5] + out[6:10])
        if dump_dir is not None:
            labels.append(_host(batch[class_key]))
            gts.append(_host(x))
            preds.append(_host(out[0]))
        bar.toc()
        m = means['eval']
        bar.show(epoch, position, total, bar.group(zip(FIELDS[0], m[:4])) + ",", bar.group(zip(FIELDS[1], m[4:])))
        if C.stop_on_nan(means):
            state['stop'] = None
        elif max_iter is not None and means.n >= max_iter:
            state['stop'] = True

    print('start training...')
    submitted = 0
    for epoch, position, total in C.epochs_of(loader, 1, 'batchStart'):
        bar.tic()
        batch = loader.getNextBatch(batchSize=batch_size)
        x = batch['input_images']
        if pipe is None:
            account(batch, x, model.getEval(inputs=(x, x, batch[class_key]), missing_prob=missing_pr, **extra), epoch, position, total)
        else:
            pending.append((batch, x, pipe.submit(inputs=(x, x, batch[class_key]), missing_prob=missing_pr, **extra), epoch, position, total))
            submitted += 1
            if len(pending) == pipe.depth:
                b_, x_, p_, e_, po_, t_ = pending.popleft()
                account(b_, x_, p_.get(), e_, po_, t_)
        if state['stop'] is None:
            return None
        if state['stop'] or (max_iter is not None and submitted >= max_iter and pipe is not None):
            break
    while pending and state['stop'] is not None:
        b_, x_, p_, e_, po_, t_ = pending.popleft()
        account(b_, x_, p_.get(), e_, po_, t_)
    if state['stop'] is None:
        return None
    print('')
    if dump_dir is not None and labels:
        os.makedirs(dump_dir, exist_ok=True)
        for suffix, parts in (('_cl_label.npy', labels), ('_gt.npy', gts), ('_pred.npy', preds)):
            np.save(os.path.join(dump_dir, str(missing_pr) + suffix), np.concatenate(parts, axis=0))
    return means['eval']


def train(
        training_epoch=1000,
        learning_rate=1e-4,
        config=None, dataset_path=None,
        save_path=None, load_path=None,
        missing_pr=0.3,
        learn='train', batch_size=72, max_iter=None, model_class='VAE', device_data=False, dump_dir=None, packed_data=False, pipeline=1,
):
    import src.module.nolbo as nolbo
    cls = nolbo.nolboSingleObject_modelnet_category_VAE if model_class == 'VAE' else nolbo.nolboSingleObject_modelnet_category_AE
    model = cls(nolbo_structure=config, learning_rate=learning_rate)
    voxel = config['encoder']['input_shape'][0]
    # device_data: the split stays in HBM as packed bits and batches are gathered + unpacked on the GPU (no per-iteration copy)
    if packed_data and not device_data:     # host loader, bits instead of floats (voxvae/hostio.py: PackedVoxels)
        loader = dataLoader(data_path=dataset_path, trainortest='test', voxel=voxel, packed=True)
    else:
        loader = (deviceDataLoader if device_data else dataLoader)(data_path=dataset_path, trainortest='test', voxel=voxel)
    category_vectors = None
    if load_path is not None:
        print('load weights...')
        model.loadModel(load_path=load_path)
        print('done!')
        cv = os.path.join(load_path, 'category_vectors.npy')
        if os.path.exists(cv):
            category_vectors = np.load(cv).astype('float32')     # the reference re-reads this file every iteration (:123)
    if category_vectors is None:                                  # no prototypes on disk: seeded stand-ins
        from voxvae import synthetic as syn
        category_vectors = syn.make_category_vectors(40, config['z_category_dim'])
    return evaluate(model, loader, missing_pr, batch_size, max_iter, dump_dir=dump_dir, pipeline=pipeline, category_vectors=category_vectors)


latent_dim = 64
config = C.make_config(latent_dim, 64, True)

if __name__ == '__main__':
    a = C.parse(__doc__)
    voxvae.set_default_dtype(a.dtype)
    sys.exit(0 if train(
        learning_rate=a.lr, config=C.make_config(a.latent, a.voxel, True), dataset_path=a.dataset_path,
        load_path=a.load_path, missing_pr=a.missing_pr, batch_size=a.batch, max_iter=a.max_iter, device_data=a.device_data,
        dump_dir=a.dump_dir, packed_data=a.packed_data, pipeline=a.pipeline,
    ) is not None else 1)

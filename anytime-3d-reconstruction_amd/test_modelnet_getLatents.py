"""Entry point mirroring the reference's test_modelnet_getLatents.py (:93-115): per-class mean of getLatent over the
training split -> <load_path>/category_vectors.npy [40, latent] (consumed by test_modelnet_VAE.py)."""
import os
import sys

import numpy as np

import _entry_common as C
import voxvae
from src.dataset_loader.modelnet_dataset import dataLoader


def train(config=None, dataset_path=None, load_path=None, batch_size=72, max_iter=None, model_class='VAE', classes=40):
    import src.module.nolbo as nolbo
    cls = nolbo.nolboSingleObject_modelnet_category_VAE if model_class == 'VAE' else nolbo.nolboSingleObject_modelnet_category_AE
    model = cls(nolbo_structure=config)
    voxel = config['encoder']['input_shape'][0]
    loader = dataLoader(data_path=dataset_path, trainortest='train', voxel=voxel)
    if load_path != None and os.path.exists(os.path.join(load_path, config['encoder']['name'] + '.voxvae.npz')):
        model.loadModel(load_path=load_path)
    category_vectors = np.zeros((classes, config['z_category_dim']))
    category_num = np.zeros(classes)
    iteration = 0
    while loader.epoch < 1:
        batch = loader.getNextBatch(batchSize=batch_size)
        if loader.epoch >= 1:
            break
        latents = model.getLatent(batch['input_images'])
        idx = np.argmax(batch['class_list'], axis=-1)
        for l, c in zip(latents, idx):
            category_vectors[c] += l
            category_num[c] += 1.0
        iteration += 1
        sys.stdout.write("it:{:04d} cur/tot:{:05d}/{:05d}  \r".format(iteration, loader.batchStart, loader.dataLength))
        if max_iter is not None and iteration >= max_iter:
            break
    print('')
    category_vectors = category_vectors / np.maximum(category_num, 1.0)[:, None]
    if load_path != None:
        os.makedirs(load_path, exist_ok=True)
        np.save(os.path.join(load_path, 'category_vectors.npy'), category_vectors)
    return category_vectors


if __name__ == '__main__':
    a = C.parse(__doc__)
    voxvae.set_default_dtype(a.dtype)
    train(config=C.make_config(a.latent, a.voxel, True), dataset_path=a.dataset_path, load_path=a.load_path, batch_size=a.batch,
          max_iter=a.max_iter)
    sys.exit(0)

"""Entry point with the role of the reference's test_modelnet_category.py (loop :94-150): one epoch of
nolboSingleObject_modelnet_category_only.getEval over the test split; prototypes = the prior network's class means.
`python test_modelnet_category.py --voxel 32 --batch 64 --missing-pr 0.5 --load-path <dir>`."""
import sys

import _entry_common as C
import voxvae
from src.dataset_loader.modelnet_dataset import dataLoader
from test_modelnet_VAE import evaluate
from train_modelnet_category import make_config


def train(
        learning_rate=1e-4,
        config=None, dataset_path=None,
        load_path=None,
        missing_pr=0.3,
        learn='train', batch_size=72, max_iter=None,
):
    import src.module.nolbo as nolbo
    model = nolbo.nolboSingleObject_modelnet_category_only(nolbo_structure=config, learning_rate=learning_rate)
    loader = dataLoader(data_path=dataset_path, trainortest='test', voxel=config['encoder']['input_shape'][0])
    if load_path is not None:
        print('load weights...')
        model.loadModel(load_path=load_path)
        print('done!')
    return evaluate(model, loader, missing_pr, batch_size, max_iter)


if __name__ == '__main__':
    a = C.parse(__doc__)
    voxvae.set_default_dtype(a.dtype)
    sys.exit(0 if train(learning_rate=a.lr, config=make_config(a.latent, a.voxel), dataset_path=a.dataset_path, load_path=a.load_path,
                        missing_pr=a.missing_pr, batch_size=a.batch, max_iter=a.max_iter) is not None else 1)

"""Entry point mirroring the reference's train_modelnet_category_dr.py: train_modelnet_category.py with
`model.fit(inputs, dropout=True)` (random-rate dropout on the mixed latent, nolbo.py:1649-1651)."""
import sys

import _entry_common as C
import voxvae
from train_modelnet_category import make_config, train

if __name__ == '__main__':
    a = C.parse(__doc__, train=True)
    voxvae.set_default_dtype(a.dtype)        # 'f32': exact-f32 parity mode; 'bf16': mixed precision (float32 master weights)
    sys.exit(0 if train(training_epoch=a.epochs, learning_rate=a.lr, batch_size=a.batch, config=make_config(a.latent, a.voxel),
                        dataset_path=a.dataset_path, save_path=a.save_path, load_path=a.load_path, max_iter=a.max_iter,
                        dropout=True) is not None else 1)

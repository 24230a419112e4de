"""Entry point mirroring the reference's train_modelnet_category_AE_dr.py (AE class, dropout=True)."""
import sys

import _entry_common as C
import voxvae
from train_modelnet_category_VAE import train

latent_dim = 64
config = C.make_config(latent_dim, 64, False)

if __name__ == '__main__':
    a = C.parse(__doc__, train=True)
    voxvae.set_default_dtype(a.dtype)        # 'f32': exact-f32 parity mode; 'bf16': mixed precision (float32 master weights)
    train(training_epoch=a.epochs, learning_rate=a.lr, batch_size=a.batch, config=C.make_config(a.latent, a.voxel, False),
          dataset_path=a.dataset_path, save_path=a.save_path, load_path=a.load_path, max_iter=a.max_iter, model_class='AE', dropout=True)
    sys.exit(0)

"""Entry point mirroring the reference's test_modelnet_AE.py (the VAE test script with the AE class)."""
import sys

import _entry_common as C
import voxvae
from test_modelnet_VAE import train

latent_dim = 64
config = C.make_config(latent_dim, 64, False)

if __name__ == '__main__':
    a = C.parse(__doc__)
    voxvae.set_default_dtype(a.dtype)
    train(learning_rate=a.lr, config=C.make_config(a.latent, a.voxel, False), dataset_path=a.dataset_path, load_path=a.load_path,
          missing_pr=a.missing_pr, batch_size=a.batch, max_iter=a.max_iter, model_class='AE')
    sys.exit(0)

"""Entry point mirroring the reference's test_modelnet_VAE.py (loop :104-156, printed fields :141-150, config
:169-192): one epoch of getEval over the test split with missing-latent correction.
`python test_modelnet_VAE.py --voxel 32 --batch 256 --dtype bf16 --missing-pr 0.9`."""
import os
import sys
import time

import numpy as np

import _entry_common as C
import voxvae
from src.dataset_loader.modelnet_dataset import dataLoader, deviceDataLoader


def train(
        training_epoch=1000,
        learning_rate=1e-4,
        config=None, dataset_path=None,
        save_path=None, load_path=None,
        missing_pr=0.3,
        learn='train', batch_size=72, max_iter=None, model_class='VAE', device_data=False,
):
    import src.module.nolbo as nolbo
    cls = nolbo.nolboSingleObject_modelnet_category_VAE if model_class == 'VAE' else nolbo.nolboSingleObject_modelnet_category_AE
    model = cls(nolbo_structure=config, learning_rate=learning_rate)
    voxel = config['encoder']['input_shape'][0]
    # device_data: the split stays in HBM as packed bits and batches are gathered + unpacked on the GPU (no per-iteration copy)
    data_loader_test = (deviceDataLoader if device_data else dataLoader)(data_path=dataset_path, trainortest='test', voxel=voxel)

    category_vectors = None
    if load_path != None:
        print('load weights...')
        model.loadModel(load_path=load_path)
        print('done!')
        cv = os.path.join(load_path, 'category_vectors.npy')
        if os.path.exists(cv):
            category_vectors = np.load(cv).astype('float32')     # reference :123 (re-read every iteration there)
    if category_vectors is None:                                  # no prototypes on disk: seeded stand-ins
        from voxvae import synthetic as syn
        category_vectors = syn.make_category_vectors(40, config['z_category_dim'])

    loss = np.zeros(8)
    epoch, epoch_curr = 0., 0.
    iteration, run_time = 0., 0.

    print('start training...')
    while epoch < 1:
        start_time = time.time()
        epoch_curr = data_loader_test.epoch
        data_start = data_loader_test.batchStart
        data_length = data_loader_test.dataLength

        batch_data = data_loader_test.getNextBatch(batchSize=batch_size)
        category_list, input_images, output_images = batch_data['class_list'], batch_data['input_images'], batch_data['input_images']
        inputs = input_images, output_images, category_list

        if epoch != epoch_curr and iteration != 0:
            break
        epoch = epoch_curr

        output_images_pred, loss_shape, pr, rc, acc_cat, \
            output_images_pred_corrected, loss_shape_corrected, pr_corrected, rc_corrected, acc_cat_corrected = model.getEval(
                inputs=inputs, category_vectors=category_vectors, missing_prob=missing_pr)

        loss_temp = loss_shape, pr, rc, acc_cat, loss_shape_corrected, pr_corrected, rc_corrected, acc_cat_corrected
        loss_temp = [float(v) for v in loss_temp]
        end_time = time.time()

        loss = (loss * iteration + np.array(loss_temp)) / (iteration + 1.0)
        run_time = (run_time * iteration + (end_time - start_time)) / (iteration + 1.0)

        sys.stdout.write(
            "it:{:04d} rt:{:.2f} Ep_o:{:03d} ".format(int(iteration + 1), run_time, int(epoch + 1)))
        sys.stdout.write("cur_o/tot_o:{:05d}/{:05d} ".format(data_start, data_length))
        sys.stdout.write(
            "loss:{:.4f}, pr:{:.4f}, rc:{:.4f}, c:{:.4f}, ".format(
                loss[0], loss[1], loss[2], loss[3]))
        sys.stdout.write(
            "closs:{:.4f}, cpr:{:.4f}, crc:{:.4f}, cc:{:.4f}  \r".format(
                loss[4], loss[5], loss[6], loss[7]))
        sys.stdout.flush()

        if np.sum(loss) != np.sum(loss):
            print('')
            print('NaN')
            return
        iteration += 1.0
        if max_iter is not None and iteration >= max_iter:
            break
    print('')
    return loss


latent_dim = 64
config = C.make_config(latent_dim, 64, True)

if __name__ == '__main__':
    a = C.parse(__doc__)
    voxvae.set_default_dtype(a.dtype)
    sys.exit(0 if train(
        learning_rate=a.lr, config=C.make_config(a.latent, a.voxel, True), dataset_path=a.dataset_path,
        load_path=a.load_path, missing_pr=a.missing_pr, batch_size=a.batch, max_iter=a.max_iter, device_data=a.device_data,
    ) is not None else 1)

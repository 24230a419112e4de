"""Entry point mirroring the reference's train_modelnet_category.py (loop :47-107, config :109-139): the VAE with a
learned class-conditional prior (nolboSingleObject_modelnet_category_only).
`python train_modelnet_category.py --voxel 32 --batch 64 --max-iter 10`."""
import sys
import time

import numpy as np

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
    data_loader_train = dataLoader(data_path=dataset_path, trainortest='train', voxel=voxel)
    data_loader_test = dataLoader(data_path=dataset_path, trainortest='test', voxel=voxel)
    if load_path != None:
        print('load weights...')
        model.loadModel(load_path=load_path)
        print('done!')

    loss = np.zeros(5)
    loss_train, loss_test = np.zeros(4), np.zeros(4)
    epoch = 0.
    iteration, run_time, total_iter = 0., 0., 0
    print('start training...')
    while epoch < training_epoch:
        start_time = time.time()
        epoch_curr = data_loader_train.epoch
        data_start = data_loader_train.batchStart
        data_length = data_loader_train.dataLength
        batch_data = data_loader_train.getNextBatch(batchSize=batch_size)
        batch_data_test = data_loader_test.getNextBatch(batchSize=batch_size)
        inputs = batch_data['input_images'], batch_data['input_images'], batch_data['class_list']
        inputs_test = batch_data_test['input_images'], batch_data_test['input_images'], batch_data_test['class_list']
        if epoch != epoch_curr and iteration != 0:
            print('')
            iteration = 0
            loss, loss_train, loss_test = loss * 0., loss_train * 0., loss_test * 0.
            run_time = 0.
            if save_path != None:
                print('save model...')
                model.saveModel(save_path=save_path)
        epoch = epoch_curr

        loss_temp = [float(v) for v in model.fit(inputs=inputs, dropout=dropout)]
        loss_train_temp = [float(v) for v in model.getEval(inputs=inputs)[1:5]]
        loss_test_temp = [float(v) for v in model.getEval(inputs=inputs_test)[1:5]]
        end_time = time.time()
        loss = (loss * iteration + np.array(loss_temp)) / (iteration + 1.0)
        loss_train = (loss_train * iteration + np.array(loss_train_temp)) / (iteration + 1.0)
        loss_test = (loss_test * iteration + np.array(loss_test_temp)) / (iteration + 1.0)
        run_time = (run_time * iteration + (end_time - start_time)) / (iteration + 1.0)
        sys.stdout.write("it:{:04d} rt:{:.2f} Ep_o:{:03d} ".format(int(iteration + 1), run_time, int(epoch + 1)))
        sys.stdout.write("cur_o/tot_o:{:04d}/{:04d} ".format(data_start, data_length))
        sys.stdout.write("kl:{:.4f}, shape:{:.4f}, reg:{:.4f}, pr:{:.4f}, rc:{:.4f}, c:{:.4f} ".format(
            loss[0], loss[1], loss[2], loss[3], loss[4], loss_train[3]))
        sys.stdout.write("shape:{:.4f}, pr:{:.4f}, rc:{:.4f}, c:{:.4f}  \r".format(
            loss_test[0], loss_test[1], loss_test[2], loss_test[3]))
        sys.stdout.flush()
        if np.sum(loss) != np.sum(loss):
            print('')
            print('NaN')
            return
        iteration += 1.0
        total_iter += 1
        if max_iter is not None and total_iter >= max_iter:
            break
    print('')
    if save_path != None:
        model.saveModel(save_path=save_path)
    return loss, loss_train, loss_test


latent_dim = 64
config = make_config(latent_dim, 64)

if __name__ == '__main__':
    a = C.parse(__doc__, train=True)
    voxvae.set_default_dtype(a.dtype)        # 'f32': exact-f32 parity mode; 'bf16': mixed precision (float32 master weights)
    sys.exit(0 if train(training_epoch=a.epochs, learning_rate=a.lr, batch_size=a.batch, config=make_config(a.latent, a.voxel),
                        dataset_path=a.dataset_path, save_path=a.save_path, load_path=a.load_path, max_iter=a.max_iter) is not None else 1)

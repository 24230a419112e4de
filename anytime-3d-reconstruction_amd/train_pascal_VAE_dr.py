"""Entry point mirroring the reference's train_pascal_VAE_dr.py (loop :119-170): nolboSingleObject_VAE with
dropout=True (random-rate latent dropout, nolbo.py:801-803) trained on (image, voxel) pairs; every iteration also
evaluates the current train and validation batches with the legacy getEval(inputs)[1:] form.
`python train_pascal_VAE_dr.py --batch 8 --image 128 --voxel 32 --max-iter 3`."""
import sys
import time

import numpy as np

import voxvae
import src.dataset_loader.pascal3D as pascal3D
import src.net_core.darknet as Darknet
from test_pascal_VAE_dr import make_config


def train(
        training_epoch=1000,
        learning_rate=1e-4,
        config=None,
        save_path=None, load_path=None,
        load_encoder_backbone_path=None, load_encoder_backbone_name=None,
        load_decoder_path=None, load_decoder_name=None,
        batch_size=72, image_size=(256, 256), max_iter=None, dataset_path=None,
):
    import src.module.nolbo as nolbo
    model = nolbo.nolboSingleObject_VAE(nolbo_structure=config, backbone_style=Darknet.Darknet19, learning_rate=learning_rate,
                                        dropout=True)
    voxel = config['decoder']['output_shape'][0]
    data_loader_pascal_train = pascal3D.dataLoaderSingleObject(trainOrVal='train', Pascal3DDataPath=dataset_path, voxel=voxel)
    data_loader_pascal_test = pascal3D.dataLoaderSingleObject(trainOrVal='val', Pascal3DDataPath=dataset_path, voxel=voxel)
    if load_path != None:
        print('load weights...')
        model.loadModel(load_path=load_path)
        print('done!')
    if load_encoder_backbone_path != None:
        model.loadEncoderBackbone(load_path=load_encoder_backbone_path, file_name=load_encoder_backbone_name)
    if load_decoder_path != None:
        model.loadDecoder(load_path=load_decoder_path, file_name=load_decoder_name)

    loss = np.zeros(4)
    loss_train, loss_test = np.zeros(3), np.zeros(3)
    epoch = 0.
    iteration, run_time, total_it = 0., 0., 0
    print('start training...')
    while epoch < training_epoch:
        start_time = time.time()
        epoch_curr = data_loader_pascal_train.epoch
        data_start = data_loader_pascal_train.dataStart
        data_length = data_loader_pascal_train.dataLength
        batch_data = data_loader_pascal_train.getNextBatch(batchSizeof3DShape=batch_size, imageSize=image_size)
        batch_data_test = data_loader_pascal_test.getNextBatch(batchSizeof3DShape=batch_size, imageSize=image_size, augmentation=False)
        inst_list, category_list, sin, cos, input_images, output_images = batch_data
        inputs = input_images, output_images
        inputs_test = batch_data_test[4], batch_data_test[5]
        if epoch != epoch_curr and iteration != 0:
            print('')
            iteration = 0
            loss, loss_train, loss_test = loss * 0., loss_train * 0., loss_test * 0.
            run_time = 0.
            if save_path != None:
                print('save model...')
                model.saveModel(save_path=save_path)
        epoch = epoch_curr

        loss_temp = model.fit(inputs=inputs)
        loss_train_temp = [float(v) for v in model.getEval(inputs=inputs)[1:]]
        loss_test_temp = [float(v) for v in model.getEval(inputs=inputs_test)[1:]]
        end_time = time.time()
        loss = (loss * iteration + np.array(loss_temp)) / (iteration + 1.0)
        loss_train = (loss_train * iteration + np.array(loss_train_temp)) / (iteration + 1.0)
        loss_test = (loss_test * iteration + np.array(loss_test_temp)) / (iteration + 1.0)
        run_time = (run_time * iteration + (end_time - start_time)) / (iteration + 1.0)
        sys.stdout.write("it:{:04d} rt:{:.2f} Ep_o:{:03d} ".format(int(iteration + 1), run_time, int(epoch + 1)))
        sys.stdout.write("cur_o/tot_o:{:04d}/{:04d} ".format(data_start, data_length))
        sys.stdout.write("kl:{:.4f}, shape:{:.4f}, pr:{:.4f}, rc:{:.4f} ".format(loss[0], loss_train[0], loss_train[1], loss_train[2]))
        sys.stdout.write("shape:{:.4f}, pr:{:.4f}, rc:{:.4f}  \r".format(loss_test[0], loss_test[1], loss_test[2]))
        sys.stdout.flush()
        if np.sum(loss) != np.sum(loss):
            print('')
            print('NaN')
            return
        iteration += 1.0
        total_it += 1
        if max_iter is not None and total_it >= max_iter:
            break
    print('')
    return loss, loss_train, loss_test


if __name__ == '__main__':
    import argparse
    ap = argparse.ArgumentParser(description=__doc__)
    ap.add_argument('--voxel', type=int, default=64)
    ap.add_argument('--latent', type=int, default=16)
    ap.add_argument('--batch', type=int, default=72)
    ap.add_argument('--image', type=int, default=256)
    ap.add_argument('--save-path', default=None)
    ap.add_argument('--load-path', default=None)
    ap.add_argument('--max-iter', type=int, default=None)
    ap.add_argument('--epochs', type=int, default=1000)
    ap.add_argument('--lr', type=float, default=1e-4)
    ap.add_argument('--dataset-path', default=None)
    a = ap.parse_args()
    voxvae.set_default_dtype('f32')          # fit() runs the exact-f32 path this round
    sys.exit(0 if train(training_epoch=a.epochs, learning_rate=a.lr, config=make_config(a.latent, a.voxel), save_path=a.save_path,
                        load_path=a.load_path, batch_size=a.batch, image_size=(a.image, a.image), max_iter=a.max_iter,
                        dataset_path=a.dataset_path) is not None else 1)

"""Entry point mirroring the reference's test_pascal_VAE_dr.py (loop :80-144, config :186-212): one epoch of
nolboSingleObject_VAE.getEval over the validation split -- Darknet19 + head2D image encoder (stock PyTorch ops),
latent 16, 64^3 decoder, missing-latent correction against the class prototypes.
`python test_pascal_VAE_dr.py --batch 16 --image 128 --missing-pr 0.9 --max-iter 2`."""
import os
import sys
import time

import numpy as np

import _entry_common as C
import voxvae
import src.dataset_loader.pascal3D as pascal3D
import src.net_core.darknet as Darknet


def make_config(latent_dim=16, voxel=64):
    return {
        'encoder_backbone': {'name': 'nolbo_backbone', 'z_dim': latent_dim, 'activation': 'elu'},
        'encoder_head': {'name': 'nolbo_head', 'output_dim': 2 * latent_dim, 'filter_num_list': [], 'filter_size_list': [],
                         'activation': 'elu'},
        'decoder': C.make_config(latent_dim, voxel, True)['decoder'],
    }


def train(
        learning_rate=1e-4,
        config=None,
        load_path=None,
        load_encoder_backbone_path=None, load_encoder_backbone_name=None,
        load_decoder_path=None, load_decoder_name=None,
        missing_pr=0.3,
        learn='train', batch_size=72, image_size=(256, 256), max_iter=None, dataset_path=None,
):
    import src.module.nolbo as nolbo
    model = nolbo.nolboSingleObject_VAE(nolbo_structure=config, backbone_style=Darknet.Darknet19, learning_rate=learning_rate)
    voxel = config['decoder']['output_shape'][0]
    data_loader_pascal = pascal3D.dataLoaderSingleObject(trainOrVal='val', Pascal3DDataPath=dataset_path, voxel=voxel)

    category_vectors = None
    if load_path != None:
        print('load weights...')
        model.loadModel(load_path=load_path)
        print('done!')
        cv = os.path.join(load_path, 'category_vectors.npy')
        if os.path.exists(cv):
            category_vectors = np.load(cv).astype('float32')
    if load_encoder_backbone_path != None:
        model.loadEncoderBackbone(load_path=load_encoder_backbone_path, file_name=load_encoder_backbone_name)
    if load_decoder_path != None:
        model.loadDecoder(load_path=load_decoder_path, file_name=load_decoder_name)
    if category_vectors is None:
        from voxvae import synthetic as syn
        category_vectors = syn.make_category_vectors(pascal3D.CLASSES, config['encoder_backbone']['z_dim'])

    loss = np.zeros(8)
    epoch, epoch_curr = 0., 0.
    iteration, run_time = 0., 0.
    print('start training...')
    while epoch < 1:
        start_time = time.time()
        epoch_curr = data_loader_pascal.epoch
        data_start = data_loader_pascal.dataStart
        data_length = data_loader_pascal.dataLength
        batch_data = data_loader_pascal.getNextBatch(batchSizeof3DShape=batch_size, imageSize=image_size, augmentation=False)
        inst_list, category_list, sin, cos, input_images, output_images = batch_data
        inputs = input_images, output_images, category_list
        if epoch != epoch_curr and iteration != 0:
            break
        epoch = epoch_curr

        output_images_pred, loss_shape, pr, rc, acc_cat, \
            output_images_pred_corrected, loss_shape_corrected, pr_corrected, rc_corrected, acc_cat_corrected = model.getEval(
                inputs=inputs, category_vectors=category_vectors, missing_prob=missing_pr)
        loss_temp = [float(v) for v in (loss_shape, pr, rc, acc_cat, loss_shape_corrected, pr_corrected, rc_corrected, acc_cat_corrected)]
        end_time = time.time()
        loss = (loss * iteration + np.array(loss_temp)) / (iteration + 1.0)
        run_time = (run_time * iteration + (end_time - start_time)) / (iteration + 1.0)
        sys.stdout.write("it:{:04d} rt:{:.2f} Ep_o:{:03d} ".format(int(iteration + 1), run_time, int(epoch + 1)))
        sys.stdout.write("cur_o/tot_o:{:05d}/{:05d} ".format(data_start, data_length))
        sys.stdout.write("loss:{:.4f}, pr:{:.4f}, rc:{:.4f}, c:{:.4f}, ".format(loss[0], loss[1], loss[2], loss[3]))
        sys.stdout.write("closs:{:.4f}, cpr:{:.4f}, crc:{:.4f}, cc:{:.4f}  \r".format(loss[4], loss[5], loss[6], loss[7]))
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


latent_dim = 16
config = make_config(latent_dim, 64)

if __name__ == '__main__':
    import argparse
    ap = argparse.ArgumentParser(description=__doc__)
    ap.add_argument('--voxel', type=int, default=64)
    ap.add_argument('--latent', type=int, default=16)
    ap.add_argument('--batch', type=int, default=72)
    ap.add_argument('--image', type=int, default=256)
    ap.add_argument('--dtype', default='f32', choices=['f32', 'bf16'])
    ap.add_argument('--load-path', default=None)
    ap.add_argument('--missing-pr', type=float, default=0.3)
    ap.add_argument('--max-iter', type=int, default=None)
    ap.add_argument('--dataset-path', default=None)
    a = ap.parse_args()
    voxvae.set_default_dtype(a.dtype)
    sys.exit(0 if train(config=make_config(a.latent, a.voxel), load_path=a.load_path, missing_pr=a.missing_pr, batch_size=a.batch,
                        image_size=(a.image, a.image), max_iter=a.max_iter, dataset_path=a.dataset_path) is not None else 1)

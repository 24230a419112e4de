"""Entry point mirroring the reference's test_modelnet_category.py (loop :94-150): one epoch of
nolboSingleObject_modelnet_category_only.getEval over the test split; prototypes = the prior network's class means.
`python test_modelnet_category.py --voxel 32 --batch 64 --missing-pr 0.5 --load-path <dir>`."""
import sys
import time

import numpy as np

import _entry_common as C
import voxvae
from src.dataset_loader.modelnet_dataset import dataLoader
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
    voxel = config['encoder']['input_shape'][0]
    data_loader = dataLoader(data_path=dataset_path, trainortest='test', voxel=voxel)
    if load_path != None:
        print('load weights...')
        model.loadModel(load_path=load_path)
        print('done!')
    loss = np.zeros(8)
    epoch, epoch_curr = 0., 0.
    iteration, run_time = 0., 0.
    print('start training...')
    while epoch < 1:
        start_time = time.time()
        epoch_curr = data_loader.epoch
        data_start = data_loader.batchStart
        data_length = data_loader.dataLength
        batch_data = data_loader.getNextBatch(batchSize=batch_size)
        inputs = batch_data['input_images'], batch_data['input_images'], batch_data['class_list']
        if epoch != epoch_curr and iteration != 0:
            break
        epoch = epoch_curr
        output_images_pred, loss_shape, pr, rc, acc_cat, \
            output_images_pred_corrected, loss_shape_corrected, pr_corrected, rc_corrected, acc_cat_corrected = model.getEval(
                inputs=inputs, missing_prob=missing_pr)
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


if __name__ == '__main__':
    a = C.parse(__doc__)
    voxvae.set_default_dtype(a.dtype)
    sys.exit(0 if train(learning_rate=a.lr, config=make_config(a.latent, a.voxel), dataset_path=a.dataset_path, load_path=a.load_path,
                        missing_pr=a.missing_pr, batch_size=a.batch, max_iter=a.max_iter) is not None else 1)

"""Entry point mirroring the reference's test_modelnet_3D.py (:100-143): for missing_pr in {0.3,0.5,0.7,0.9} evaluate one
test batch with the AE and the VAE (legacy 4-tuple getEval form, :125-126) and dump every sample as a (D*D, D) text
matrix -- the project's de-facto output format for external rendering.  The third model of the reference script
(`nolboSingleObject_modelnet_category_only`, prior-net class) is outside the hot-path scope."""
import os
import sys

import numpy as np

import _entry_common as C
import voxvae
from src.dataset_loader.modelnet_dataset import dataLoader


def test(dataset_path=None, batch_size=72, save_dir='./data/eval/image/modelnet/', voxel=32, latent=64,
         load_path_AE=None, load_path_VAE=None, missing_prs=(0.30, 0.50, 0.70, 0.90)):
    import src.module.nolbo as nolbo
    model_AE = nolbo.nolboSingleObject_modelnet_category_AE(nolbo_structure=C.make_config(latent, voxel, False))
    model_VAE = nolbo.nolboSingleObject_modelnet_category_VAE(nolbo_structure=C.make_config(latent, voxel, True))
    if load_path_AE:
        model_AE.loadModel(load_path_AE)
    if load_path_VAE:
        model_VAE.loadModel(load_path_VAE)
    data_loader_test = dataLoader(data_path=dataset_path, trainortest='test', voxel=voxel)
    batch_data = data_loader_test.getNextBatch(batchSize=batch_size)
    data = batch_data['input_images'], batch_data['input_images'], batch_data['class_list']
    os.makedirs(save_dir, exist_ok=True)
    D = voxel
    for missing_pr in missing_prs:
        print(missing_pr)
        input_images, output_images = data[0], data[1]
        output_images_pred_AE, _, _, _ = model_AE.getEval(inputs=data[0:2], missing_prob=missing_pr)
        output_images_pred_VAE, _, _, _ = model_VAE.getEval(inputs=data[0:2], missing_prob=missing_pr)
        pred_AE, pred_VAE = np.array(output_images_pred_AE), np.array(output_images_pred_VAE)
        for i, (gt, p_ae, p_vae) in enumerate(zip(output_images, pred_AE, pred_VAE)):
            file_name = '{:03d}'.format(int(i))
            np.savetxt(os.path.join(save_dir, file_name + '_' + str(missing_pr) + '_gt.txt'), np.reshape(gt, (D * D, D)))
            np.savetxt(os.path.join(save_dir, file_name + '_' + str(missing_pr) + '_AE.txt'), np.reshape(p_ae, (D * D, D)))
            np.savetxt(os.path.join(save_dir, file_name + '_' + str(missing_pr) + '_VAE.txt'), np.reshape(p_vae, (D * D, D)))


if __name__ == '__main__':
    a = C.parse(__doc__)
    voxvae.set_default_dtype(a.dtype)
    sys.exit(test(dataset_path=a.dataset_path, batch_size=a.batch, voxel=a.voxel, latent=a.latent, load_path_VAE=a.load_path,
                  save_dir=a.save_path or './data/eval/image/modelnet/'))

"""Child process of tests/test_gpu_rccl.py: a one-rank RCCL process group on the 1-GPU box.

Started fresh (nothing has touched the GPU before init_process_group), WORLD_SIZE = 1, RANK = 0.  One training step is run
twice from the same weights: once with no collective (the plain single-process path), once with every gradient bucket
sent through `torch.distributed.all_reduce` on backend 'nccl' (= RCCL) as the backward pass reports its gradients -- the
code path the 8-GPU run takes.  A one-rank sum is the identity, so the two steps must agree bit for bit.  Prints one JSON line."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'anytime-3d-reconstruction_amd'))


def main():
    import contextlib
    import numpy as np
    import torch
    import torch.distributed as dist
    dev = 'cuda:0'
    torch.cuda.set_device(0)
    dist.init_process_group('nccl', device_id=torch.device(dev))            # RCCL on ROCm
    import voxvae
    from voxvae import synthetic as syn
    from voxvae import train as T
    dtype = sys.argv[1] if len(sys.argv) > 1 else 'bf16'
    voxvae.set_default_dtype(dtype)
    voxvae.set_default_device(dev)
    import src.module.nolbo as nolbo
    cfg = syn.make_config(32, 64, True)
    ep = syn.make_encoder_params(cfg['encoder'], seed=42, nontrivial_affine=True)
    dp = syn.make_decoder_params(cfg['decoder'], seed=43, nontrivial_affine=True, final_gain=2.0)
    B = 8
    x = torch.from_numpy(syn.make_voxels(B, 32, seed=31)).to(dev)
    eps = torch.from_numpy(syn.make_eps(B, 64, seed=32)).to(dev)
    res = {}
    for tag, collective in (('plain', False), ('rccl', True)):
        with contextlib.redirect_stdout(sys.stderr):
            model = nolbo.nolboSingleObject_modelnet_category_VAE(nolbo_structure=cfg, learning_rate=1e-3)
        model._encoder.set_weights_dict(ep)
        model._decoder.set_weights_dict(dp)
        tr = T.Trainer(model._enc_eng, model._dec_eng, True, 1e-3, world_size=dist.get_world_size())
        tr.grads.always_reduce = collective
        tr.step(x, x, eps)
        torch.cuda.synchronize()
        res[tag] = {'grads': {n: tr.grads.views[n].clone() for n, _ in tr.order},
                    'w': {k: v.clone() for k, v in list(model._enc_eng.params.items()) + list(model._dec_eng.params.items())},
                    'launch_order': list(tr.grads.launch_order),
                    'works': [type(w).__name__ for w in tr.grads._works]}
    same_g = all(torch.equal(res['plain']['grads'][n], res['rccl']['grads'][n]) for n in res['plain']['grads'])
    same_w = all(torch.equal(a, b) for a, b in zip(res['plain']['w'].values(), res['rccl']['w'].values()))
    # and a collective whose result is not the identity of its input layout: all_gather of a rank-stamped tensor
    t = torch.full((4,), 7.0, device=dev)
    out = [torch.empty_like(t)]
    dist.all_gather(out, t)
    print(json.dumps({'backend': dist.get_backend(), 'world': dist.get_world_size(), 'grads_bit_identical': bool(same_g),
                      'weights_bit_identical': bool(same_w), 'buckets': len(res['rccl']['launch_order']),
                      'launch_order': res['rccl']['launch_order'], 'work_types': res['rccl']['works'],
                      'plain_work_types': res['plain']['works'], 'all_gather_ok': bool(torch.equal(out[0], t)), 'dtype': dtype}))
    dist.destroy_process_group()


if __name__ == '__main__':
    main()

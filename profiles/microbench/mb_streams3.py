"""2 vs 3 vs 4 streams, interleaved repetitions in one process (ONE model: the engines keep a workspace per stream since round 3)."""
import contextlib, json, sys, time, os
_R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, _R); sys.path.insert(0, os.path.join(_R, 'anytime-3d-reconstruction_amd'))
import numpy as np, torch
import voxvae
from voxvae import synthetic as syn
voxvae.set_default_dtype('bf16'); voxvae.set_default_device('cuda:0')
import src.module.nolbo as nolbo
cfg = syn.make_config(32, 64, True)
ep, dp = syn.make_encoder_params(cfg['encoder']), syn.make_decoder_params(cfg['decoder'])
with contextlib.redirect_stdout(sys.stderr):
    m = nolbo.nolboSingleObject_modelnet_category_VAE(nolbo_structure=cfg)
m._encoder.set_weights_dict(ep); m._decoder.set_weights_dict(dp)
B = 256
x = torch.from_numpy(syn.make_voxels(B, 32, seed=1234)).cuda(); eps = torch.from_numpy(syn.make_eps(B, 64, seed=7)).cuda()
m.eval_forward_device(x, x, eps); torch.cuda.synchronize()
streams = [torch.cuda.Stream() for _ in range(4)]
def run(NS, steps):
    for i in range(steps):
        with torch.cuda.stream(streams[i % NS]):
            m.eval_forward_device(x, x, eps)
for rep in range(4):
    for NS in (2, 3, 4, 1):
        run(NS, 40); torch.cuda.synchronize()
        t0 = time.perf_counter(); run(NS, 400); torch.cuda.synchronize(); el = time.perf_counter() - t0
        print(json.dumps({'rep': rep, 'streams': NS, 'ms_per_step': round(1e3 * el / 400, 4), 'rec_s': round(B * 400 / el)}), flush=True)

"""Two-stream step rate with equal and with unequal stream priorities."""
import contextlib, json, sys, time
import os; _R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, _R); sys.path.insert(0, os.path.join(_R, 'anytime-3d-reconstruction_amd'))
import numpy as np, torch
import voxvae
from voxvae import synthetic as syn
voxvae.set_default_dtype('bf16'); voxvae.set_default_device('cuda:0')
import src.module.nolbo as nolbo
cfg = syn.make_config(32, 64, True)
ep, dp = syn.make_encoder_params(cfg['encoder']), syn.make_decoder_params(cfg['decoder'])
def build():
    with contextlib.redirect_stdout(sys.stderr):
        m = nolbo.nolboSingleObject_modelnet_category_VAE(nolbo_structure=cfg)
    m._encoder.set_weights_dict(ep); m._decoder.set_weights_dict(dp)
    return m
B = 256
x = torch.from_numpy(syn.make_voxels(B, 32, seed=1234)).cuda(); eps = torch.from_numpy(syn.make_eps(B, 64, seed=7)).cuda()
print('priority range', torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, 'priority_range') else None)
models = [build() for _ in range(4)]
for rnd in range(4):
    for name, prios in (('two', (0, 0)), ('three', (0, 0, 0)), ('four', (0, 0, 0, 0))):
        streams = [torch.cuda.Stream(priority=p) for p in prios]
        NS = len(streams)
        def run(steps):
            for i in range(steps):
                with torch.cuda.stream(streams[i % NS]):
                    models[i % NS].eval_forward_device(x, x, eps)
        torch.cuda.synchronize()
        run(40); torch.cuda.synchronize()
        t0 = time.perf_counter(); run(400); torch.cuda.synchronize(); el = time.perf_counter() - t0
        print(json.dumps({'mode': name, 'ms_per_step': round(1e3 * el / 400, 4)}), flush=True)

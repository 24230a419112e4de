"""Is the two-stream step bound by the clock the chip holds under load?  Same launches on all-zero parameters and inputs (the
MFMAs toggle nothing -> the chip keeps its clock) against random ones."""
import contextlib, json, sys, time
import os; _R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, _R); sys.path.insert(0, os.path.join(_R, 'anytime-3d-reconstruction_amd'))
import numpy as np, torch
import voxvae
from voxvae import synthetic as syn
voxvae.set_default_dtype('bf16'); voxvae.set_default_device('cuda:0')
import src.module.nolbo as nolbo
cfg = syn.make_config(32, 64, True)
ep, dp = syn.make_encoder_params(cfg['encoder']), syn.make_decoder_params(cfg['decoder'])
def build(zero):
    with contextlib.redirect_stdout(sys.stderr):
        m = nolbo.nolboSingleObject_modelnet_category_VAE(nolbo_structure=cfg)
    e = {k: (np.zeros_like(v) if zero and 'variance' not in k else v) for k, v in ep.items()}
    d = {k: (np.zeros_like(v) if zero and 'variance' not in k else v) for k, v in dp.items()}
    m._encoder.set_weights_dict(e); m._decoder.set_weights_dict(d)
    return m
B = 256
xr = torch.from_numpy(syn.make_voxels(B, 32, seed=1234)).cuda(); eps = torch.from_numpy(syn.make_eps(B, 64, seed=7)).cuda()
for rnd in range(2):
    for zero in (False, True):
        x = torch.zeros_like(xr) if zero else xr
        e = torch.zeros_like(eps) if zero else eps
        for NS in (1, 2):
            models = [build(zero) for _ in range(NS)]
            streams = [torch.cuda.Stream() for _ in range(NS)]
            def run(steps):
                for i in range(steps):
                    with torch.cuda.stream(streams[i % NS]):
                        models[i % NS].eval_forward_device(x, x, e)
            torch.cuda.synchronize()
            run(50); torch.cuda.synchronize()
            t0 = time.perf_counter(); run(400); torch.cuda.synchronize(); el = time.perf_counter() - t0
            print(json.dumps({'zero': zero, 'streams': NS, 'ms_per_step': round(1e3 * el / 400, 4)}), flush=True)

"""Round 4: where the reference-convention call (numpy in, numpy out, test_modelnet_VAE.py:114-130) spends its time.
getEval on host arrays per chunk count (VV_HOST_CHUNKS), float32 / bit-packed input, float32 / uint8 prediction; and the same
call's pieces: device-resident compute alone, download alone.  Run on the GPU box: python profiles/microbench/mb_host_chunks.py"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'anytime-3d-reconstruction_amd'))
import contextlib
import numpy as np, torch
import voxvae
from voxvae import synthetic as syn, hostio
voxvae.set_default_dtype('bf16'); voxvae.set_default_device('cuda:0')
import src.module.nolbo as nolbo
B = 256
cfg = syn.make_config(32, 64, True)
with contextlib.redirect_stdout(sys.stderr):
    m = nolbo.nolboSingleObject_modelnet_category_VAE(nolbo_structure=cfg)
m._encoder.set_weights_dict(syn.make_encoder_params(cfg['encoder'])); m._decoder.set_weights_dict(syn.make_decoder_params(cfg['decoder']))
xh, epsh = syn.make_voxels(B, 32), syn.make_eps(B, 64)
oh, cats = syn.make_onehot(B, 40), syn.make_category_vectors(40, 64)
xp = hostio.pack_voxels(xh)

def timeit(fn, n=30, w=8):
    for _ in range(w): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3

res = {}
for chunks in (1, 2, 3, 4, 6, 8):
    os.environ['VV_HOST_CHUNKS'] = str(chunks)
    for inp, xin in (('f32_in', xh), ('packed_in', xp)):
        for outdt in ('float32', 'uint8'):
            hostio.set_prediction_host_dtype(outdt)
            def call():
                out = m.getEval(inputs=(xin, xin, oh), category_vectors=cats, missing_prob=0.0, _eps=epsh)
                return np.array(out[0]), float(out[1])
            res['chunks%d_%s_%s' % (chunks, inp, outdt)] = round(timeit(call), 4)
hostio.set_prediction_host_dtype('float32')
xd, epsd = torch.from_numpy(xh).cuda(), torch.from_numpy(epsh).cuda()
res['device_resident_one_stream_ms'] = round(timeit(lambda: m.eval_forward_device(xd, xd, epsd)), 4)
pred = torch.empty(B, 32, 32, 32, 1, device='cuda'); host = torch.empty(B, 32, 32, 32, 1).pin_memory()
res['d2h_33MB_pinned_ms'] = round(timeit(lambda: host.copy_(pred, non_blocking=True)), 4)
p8 = torch.empty(B, 32, 32, 32, 1, device='cuda', dtype=torch.uint8); h8 = torch.empty(B, 32, 32, 32, 1, dtype=torch.uint8).pin_memory()
res['d2h_8MB_pinned_ms'] = round(timeit(lambda: h8.copy_(p8, non_blocking=True)), 4)
res['h2d_33MB_pageable_ms'] = round(timeit(lambda: torch.from_numpy(xh).to('cuda')), 4)
# the Python side alone: how long the launches of one device-resident step take to ENQUEUE (no sync inside)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(50): m.eval_forward_device(xd, xd, epsd)
res['enqueue_ms_per_step'] = round((time.perf_counter() - t0) / 50 * 1e3, 4)
torch.cuda.synchronize()
print(json.dumps(res, indent=1))

"""Round 4: the pipelined host loop (voxvae.streams.HostPipeline) by depth and data format; ms per 256-batch.
python profiles/microbench/mb_host_pipeline.py"""
import collections, contextlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'anytime-3d-reconstruction_amd'))
import numpy as np, torch
import voxvae
from voxvae import synthetic as syn, hostio
from voxvae.streams import HostPipeline
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

def run(xin, depth, n=60, consume_pred=True, consume_scalar=True):
    pipe, pend = HostPipeline(m, depth), collections.deque()
    def consume(p):
        out = p.get()
        a = np.array(out[0]) if consume_pred else None
        s = float(out[1]) if consume_scalar else None
        return a, s
    for _ in range(2 * depth):
        pend.append(pipe.submit(inputs=(xin, xin, oh), category_vectors=cats, _eps=epsh))
    while pend: consume(pend.popleft())
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        pend.append(pipe.submit(inputs=(xin, xin, oh), category_vectors=cats, _eps=epsh))
        if len(pend) == depth: consume(pend.popleft())
    while pend: consume(pend.popleft())
    return round((time.perf_counter() - t0) / n * 1e3, 4)

res = {}
for pdt in ('float32', 'uint8'):
    hostio.set_prediction_host_dtype(pdt)
    for name, xin in (('f32_in', xh), ('packed_in', xp)):
        for depth in (1, 2, 3, 4, 6):
            res['%s_%s_out_depth%d' % (name, pdt, depth)] = run(xin, depth)
hostio.set_prediction_host_dtype('float32')
res['packed_in_float32_out_depth3_no_scalar_read'] = run(xp, 3, consume_scalar=False)
res['packed_in_float32_out_depth3_no_pred_read'] = run(xp, 3, consume_pred=False)
# submit cost alone (host side): time to enqueue one batch with nothing consumed until the end
pipe = HostPipeline(m, 3); ps = []
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(6): ps.append(pipe.submit(inputs=(xp, xp, oh), category_vectors=cats, _eps=epsh))
res['submit_ms_host_side_packed'] = round((time.perf_counter() - t0) / 6 * 1e3, 4)
for p in ps: p.get()
print(json.dumps(res, indent=1))

"""Round 4: the four BatchNorm sweeps of the training step at the shapes of the 32^3 model (rows x channels, bf16), back to back:
achieved bytes/s against the 6.29 TB/s copy rate.  Optional hooks (hook build only): VV_BN_NB = block cap of the two reductions,
VV_BN_SWEEP = block cap of the two element-wise sweeps.  python profiles/microbench/mb_bn.py [tag]"""
import ctypes, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'anytime-3d-reconstruction_amd'))
import torch
from voxvae import lib as L
lib = L.load()
st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
shapes = [('E1/D4 out', 256 * 4096, 64), ('E2/D3 out', 256 * 512, 128), ('E3/D2 out', 256 * 64, 256), ('E4/D1 out', 256 * 8, 512)]
res = {}
for name, R, C in shapes:
    x = torch.randn(R, C, device='cuda').bfloat16(); dy = torch.randn(R, C, device='cuda').bfloat16()
    y = torch.empty_like(x); dx = torch.empty_like(x)
    f = lambda: torch.empty(C, device='cuda')
    gamma, beta = torch.ones(C, device='cuda'), torch.zeros(C, device='cuda')
    mean, var, rstd, scale, shift, mm, mv, dg, db = f(), f(), f(), f(), f(), torch.zeros(C, device='cuda'), torch.ones(C, device='cuda'), f(), f()
    ws = torch.empty(max(lib.vv_bn_workspace_bytes(R, C), 16), dtype=torch.uint8, device='cuda')
    def stats():
        L.call('vv_bn_train_stats', L.ptr(x), R, C, L.ptr(gamma), L.ptr(beta), 1e-3, 0.99, L.ptr(mean), L.ptr(var), L.ptr(rstd), L.ptr(scale),
               L.ptr(shift), L.ptr(mm), L.ptr(mv), L.VV_BF16, L.ptr(ws), ws.numel(), st())
    def fwd():
        L.call('vv_bn_act_fwd', L.ptr(x), L.ptr(scale), L.ptr(shift), L.ptr(y), R, C, 1, L.VV_BF16, st())
    def bwd():
        L.call('vv_bn_act_bwd', L.ptr(x), L.ptr(dy), L.ptr(scale), L.ptr(shift), L.ptr(mean), L.ptr(rstd), L.ptr(dg), L.ptr(db), L.ptr(dx), R, C, 1,
               L.VV_BF16, L.ptr(ws), ws.numel(), st())
    nbytes = R * C * 2
    for label, fn, moved in (('stats (reduce + finalize)', stats, nbytes), ('apply', fwd, 2 * nbytes), ('backward (reduce + finalize + apply)', bwd, 5 * nbytes)):
        for _ in range(5): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(50): fn()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 50 * 1e3
        res['%s [%d x %d] %s' % (name, R, C, label)] = {'us': round(us, 2), 'TBps': round(moved / us / 1e6, 3)}
print(json.dumps({'hooks': {k: os.environ.get(k) for k in ('VV_BN_NB', 'VV_BN_SWEEP', 'VV_BN_ROWS')}, 'results': res}, indent=1))

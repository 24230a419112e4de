"""Back-to-back time of E4 / D2 (pg_kernel + pg_reduce_kernel, batch 256) for the ablation builds of pg_ablate.py, interleaved in one process.
usage: mb_pg_ablate.py <mask> ..."""
import ctypes, json, os, sys, time
import torch
_R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, _R); sys.path.insert(0, os.path.join(_R, 'anytime-3d-reconstruction_amd'))
from voxvae import lib as L
lib0 = L.load()
DEV = 'cuda:0'; B = 256
cs = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
torch.manual_seed(0)
sc = torch.ones(512, device=DEV); sh = torch.zeros(512, device=DEV)
def packed(fn, wshape, cin, cout):
    w = (torch.randn(*wshape, device=DEV) / (27 * cin) ** 0.5).float().contiguous()
    o = torch.empty(64 * cin * cout, dtype=torch.bfloat16, device=DEV)
    L.call(fn, L.ptr(w), L.ptr(o), cin, cout, cs); return o
w3 = packed('vv_pack_conv_k4_skip', (4, 4, 4, 256, 512), 256, 512); x3 = torch.randn(B, 4, 4, 4, 256, device=DEV).to(torch.bfloat16); y3 = torch.empty(B, 2, 2, 2, 512, dtype=torch.bfloat16, device=DEV)
w4 = packed('vv_pack_convT_k4s2_skip', (4, 4, 4, 256, 512), 512, 256); x4 = torch.randn(B, 2, 2, 2, 512, device=DEV).to(torch.bfloat16); y4 = torch.empty(B, 4, 4, 4, 256, dtype=torch.bfloat16, device=DEV)
ws = torch.empty(64 << 20, dtype=torch.uint8, device=DEV)
names = sys.argv[1:]
libs = {n: ctypes.CDLL(os.path.join(_R, 'scratch/abl/libpg_%s.so' % n)) for n in names}
def e4(n):
    f = libs[n].vv_conv3d_k4s2_pos_fwd; f.restype = ctypes.c_int
    assert f(L.ptr(x3), L.ptr(w3), L.ptr(sc), L.ptr(sh), L.ptr(y3), B, 4, 256, 512, 1, L.VV_BF16, L.ptr(ws), ctypes.c_size_t(ws.numel()), cs) == 0
def d2(n):
    f = libs[n].vv_convT3d_k4s2_pos_fwd; f.restype = ctypes.c_int
    assert f(L.ptr(x4), L.ptr(w4), L.ptr(sc), L.ptr(sh), L.ptr(y4), B, 2, 512, 256, 1, L.VV_BF16, L.ptr(ws), ctypes.c_size_t(ws.numel()), cs) == 0
what = {'0': 'full (kernel + reduce)', '1': 'no MFMA', '2': 'no fragment reads', '3': 'no MFMA, no fragment reads', '4': 'no DMA in the loop', '8': 'no counted wait + barrier in the loop',
        '12': 'no DMA, no barriers', '15': 'loop skeleton + epilogue + reduce only'}
N = 300
for name, fn in (('E4', e4), ('D2', d2)):
    res = {n: [] for n in names}
    for rep in range(2):
        for n in names:
            for i in range(20): fn(n)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for i in range(N): fn(n)
            torch.cuda.synchronize(); res[n].append(round(1e6 * (time.perf_counter() - t0) / N, 2))
    for n in names: print(json.dumps({'layer': name, 'abl': n, 'what': what.get(n), 'us': res[n]}), flush=True)

"""D4 (ctw16) and E2 (conv_direct16): tree vs s_setprio-1-for-waves-4..7 build, back to back, interleaved."""
import ctypes, json, os, sys, time
import torch
_R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, _R); sys.path.insert(0, os.path.join(_R, 'anytime-3d-reconstruction_amd'))
from voxvae import lib as L
new = L.load()
libs = {'tree': new, 'prio': ctypes.CDLL(os.path.join(_R, 'scratch/prio/libvoxvae_prio.so'))}
DEV = 'cuda:0'; B = 256
cs = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
torch.manual_seed(0)
# D4
w4 = (torch.randn(4, 4, 4, 64, 128, device=DEV) / 32).float().contiguous()
wk = torch.empty(64 * 128 * 64, dtype=torch.bfloat16, device=DEV)
L.call('vv_pack_convT_k4s2_skip', L.ptr(w4), L.ptr(wk), 128, 64, cs)
x4 = torch.randn(B, 8, 8, 8, 128, device=DEV).to(torch.bfloat16)
y4 = {k: torch.empty(B, 16, 16, 16, 64, dtype=torch.bfloat16, device=DEV) for k in libs}
sc = torch.rand(128, device=DEV) + 0.5; sh = torch.randn(128, device=DEV) * 0.3
# E2
w2 = (torch.randn(4, 4, 4, 64, 128, device=DEV) / 64).float().contiguous()
wp = torch.empty(128, 64 * 64, dtype=torch.bfloat16, device=DEV)
L.call('vv_pack_conv_k4', L.ptr(w2), L.ptr(wp), 64, 128, L.VV_BF16, cs)
x2 = torch.randn(B, 16, 16, 16, 64, device=DEV).to(torch.bfloat16)
y2 = {k: torch.empty(B, 8, 8, 8, 128, dtype=torch.bfloat16, device=DEV) for k in libs}
def d4(k):
    f = libs[k].vv_convT3d_k4s2_whole_fwd; f.restype = ctypes.c_int
    assert f(L.ptr(x4), L.ptr(wk), L.ptr(sc), L.ptr(sh), L.ptr(y4[k]), B, 8, 128, 64, 1, L.VV_BF16, cs) == 0
def e2(k):
    f = libs[k].vv_conv3d_k4s2_direct_fwd; f.restype = ctypes.c_int
    assert f(L.ptr(x2), L.ptr(wp), L.ptr(sc), L.ptr(sh), L.ptr(y2[k]), B, 16, 64, 128, 1, L.VV_BF16, cs) == 0
for k in libs: d4(k); e2(k)
torch.cuda.synchronize()
print('D4 equal', torch.equal(y4['tree'], y4['prio']), 'E2 equal', torch.equal(y2['tree'], y2['prio']))
N = 300
for rep in range(3):
    for name, fn in (('D4', d4), ('E2', e2)):
        for k in libs:
            for _ in range(20): fn(k)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(N): fn(k)
            torch.cuda.synchronize()
            print(json.dumps({'layer': name, 'lib': k, 'us': round(1e6 * (time.perf_counter() - t0) / N, 2)}), flush=True)

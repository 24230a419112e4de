"""Round 4: does a consumer kernel run faster when its input is resident in the 256 MB Infinity Cache?  On ONE device, HIP events around every
launch: the widest encoder layer (E2) and the last layer (D5) (a) back to back on the same buffers (input re-read every iteration: resident),
(b) right after their producer (E1 / D4) wrote the input -- the order of the real path, (c) after a 512 MB fill that evicts everything.
python profiles/microbench/mb_residency.py"""
import ctypes, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'anytime-3d-reconstruction_amd'))
import numpy as np, torch
from voxvae import lib as L
lib = L.load()
DEV = 'cuda:0'
B = 256
st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
g = torch.Generator(device=DEV).manual_seed(1)
bt = torch.bfloat16
# E1: 32^3 x 1 -> 16^3 x 64
x = (torch.rand(B, 32, 32, 32, 1, device=DEV, generator=g) < 0.15).float().contiguous()
w1 = (torch.randn(4, 4, 4, 1, 64, device=DEV, generator=g) / 8).contiguous()
w1p = torch.empty(64, 64, dtype=bt, device=DEV); L.call('vv_pack_conv_k4', L.ptr(w1), L.ptr(w1p), 1, 64, L.VV_BF16, st())
sc64, sh64 = torch.rand(64, device=DEV) + 0.5, torch.randn(64, device=DEV) * 0.3
h1 = torch.empty(B, 16, 16, 16, 64, dtype=bt, device=DEV)
def e1(): L.call('vv_conv3d_first_fwd', L.ptr(x), L.ptr(w1p), L.ptr(sc64), L.ptr(sh64), L.ptr(h1), B, 32, 64, 1, L.VV_BF16, st())
# E2: 16^3 x 64 -> 8^3 x 128
w2 = (torch.randn(4, 4, 4, 64, 128, device=DEV, generator=g) / 64).contiguous()
w2p = torch.empty(128, 64 * 64, dtype=bt, device=DEV); L.call('vv_pack_conv_k4', L.ptr(w2), L.ptr(w2p), 64, 128, L.VV_BF16, st())
sc128, sh128 = torch.rand(128, device=DEV) + 0.5, torch.randn(128, device=DEV) * 0.3
h2 = torch.empty(B, 8, 8, 8, 128, dtype=bt, device=DEV)
def e2(): L.call('vv_conv3d_k4s2_direct_fwd', L.ptr(h1), L.ptr(w2p), L.ptr(sc128), L.ptr(sh128), L.ptr(h2), B, 16, 64, 128, 1, L.VV_BF16, st())
# D4: 8^3 x 128 -> 16^3 x 64
w4 = (torch.randn(4, 4, 4, 64, 128, device=DEV, generator=g) / 32).contiguous()
w4k = torch.empty(64 * 128 * 64, dtype=bt, device=DEV); L.call('vv_pack_convT_k4s2_skip', L.ptr(w4), L.ptr(w4k), 128, 64, st())
d3 = torch.randn(B, 8, 8, 8, 128, device=DEV, generator=g).to(bt)
d4o = torch.empty(B, 16, 16, 16, 64, dtype=bt, device=DEV)
def d4(): L.call('vv_convT3d_k4s2_whole_fwd', L.ptr(d3), L.ptr(w4k), L.ptr(sc64), L.ptr(sh64), L.ptr(d4o), B, 8, 128, 64, 1, L.VV_BF16, st())
# D5: 16^3 x 64 -> 32^3 x 1 + losses
w5 = (torch.randn(4, 4, 4, 1, 64, device=DEV, generator=g) / 16).contiguous()
probs = torch.empty(B, 32, 32, 32, 1, device=DEV); stats = torch.empty(B, 4, device=DEV)
ws5 = torch.empty(max(lib.vv_convT3d_final_bce_workspace_bytes(B, 16), 16), dtype=torch.uint8, device=DEV)
def d5(): L.call('vv_convT3d_final_bce_fwd', L.ptr(d4o), L.ptr(w5), L.ptr(x), L.ptr(probs), None, L.ptr(stats), B, 16, 64, 0.6, 1e-7, L.VV_BF16, L.ptr(ws5), ws5.numel(), st())
junk = torch.empty(512 << 20, dtype=torch.uint8, device=DEV)
def flush(): junk.fill_(1)

def timed(fn, pre=None, n=40, w=8):
    ts = []
    for i in range(n + w):
        if pre is not None: pre()
        e0, e1_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1_.record()
        ts.append((e0, e1_))
    torch.cuda.synchronize()
    v = [a.elapsed_time(b) * 1e3 for a, b in ts[w:]]
    return {'median_us': round(float(np.median(v)), 2), 'min_us': round(min(v), 2)}
e1(); d4(); torch.cuda.synchronize()
res = {}
for rep in range(2):
    res['E2 back to back (input resident)  #%d' % rep] = timed(e2)
    res['E2 right after E1 wrote its input  #%d' % rep] = timed(e2, pre=e1)
    res['E2 after a 512 MB fill (cold)     #%d' % rep] = timed(e2, pre=flush)
    res['D5 back to back (input resident)  #%d' % rep] = timed(d5)
    res['D5 right after D4 wrote its input  #%d' % rep] = timed(d5, pre=d4)
    res['D5 after a 512 MB fill (cold)     #%d' % rep] = timed(d5, pre=flush)
    res['D4 back to back                   #%d' % rep] = timed(d4)
    res['D4 after a 512 MB fill (cold)     #%d' % rep] = timed(d4, pre=flush)
    res['E1 back to back                   #%d' % rep] = timed(e1)
    res['E1 after a 512 MB fill (cold)     #%d' % rep] = timed(e1, pre=flush)
print(json.dumps(res, indent=1))

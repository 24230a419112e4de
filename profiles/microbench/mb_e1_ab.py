import ctypes, json, os, sys, time
import torch
_R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, _R); sys.path.insert(0, os.path.join(_R, 'anytime-3d-reconstruction_amd'))
from voxvae import lib as L
new = L.load()
# usage: mb_e1_ab.py [other lib .so] [name]   (default: the nontemporal-store build of round 3 under scratch/nt)
OTHER = sys.argv[2] if len(sys.argv) > 2 else 'nt'
libs = {'tree': new, OTHER: ctypes.CDLL(os.path.join(_R, sys.argv[1] if len(sys.argv) > 1 else 'scratch/nt/libvoxvae_nt.so'))}
DEV = 'cuda:0'; B = 256
x = (torch.rand(B, 32, 32, 32, 1, device=DEV) < 0.1).float().contiguous()
w = (torch.randn(4, 4, 4, 1, 64, device=DEV) / 8).float().contiguous()
sc = torch.rand(64, device=DEV) + 0.5; sh = torch.randn(64, device=DEV) * 0.3
wp = torch.empty(64, 64, dtype=torch.bfloat16, device=DEV)
cs = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
L.call('vv_pack_conv_k4', L.ptr(w), L.ptr(wp), 1, 64, L.VV_BF16, cs)
ys = {k: torch.empty(B, 16, 16, 16, 64, dtype=torch.bfloat16, device=DEV) for k in libs}
def launch(k):
    f = libs[k].vv_conv3d_first_fwd; f.restype = ctypes.c_int
    assert f(L.ptr(x), L.ptr(wp), L.ptr(sc), L.ptr(sh), L.ptr(ys[k]), B, 32, 64, 1, L.VV_BF16, cs) == 0
for k in libs: launch(k)
torch.cuda.synchronize(); print('equal', torch.equal(ys['tree'], ys[OTHER]))
big = torch.empty(512 << 20, dtype=torch.uint8, device=DEV)
N = 200
for rep in range(3):
    for k in libs:
        for i in range(20): launch(k)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(N): launch(k)
        torch.cuda.synchronize(); b2b = 1e6 * (time.perf_counter() - t0) / N
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(50)]
        for a, b in ev:
            big.zero_(); a.record(); launch(k); b.record()       # cold caches: 512 MB written in between
        torch.cuda.synchronize(); t = sorted(a.elapsed_time(b) for a, b in ev)
        print(json.dumps({'lib': k, 'back_to_back_us': round(b2b, 1), 'cold_median_us': round(1e3 * t[25], 1)}), flush=True)

"""Back-to-back throughput of the E2 kernel (conv_direct.hip) in its two MFMA shapes."""
import ctypes, json, os, sys, time
import torch
import os; _R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, _R); sys.path.insert(0, os.path.join(_R, 'anytime-3d-reconstruction_amd'))
os.environ.setdefault('VOXVAE_TEST_HOOKS', '1')   # the kernel-form overrides live in lib/libvoxvae_hooks.so (voxvae/lib.py)
from voxvae import lib as L
L.load()
DEV = 'cuda:0'
B, cin, cout = 256, 64, 128
w = (torch.randn(4, 4, 4, cin, cout, device=DEV) / (64 * cin) ** 0.5).float().contiguous()
sc = torch.rand(cout, device=DEV) + 0.5; sh = torch.randn(cout, device=DEV) * 0.3
wp = torch.empty(cout, 64 * cin, dtype=torch.bfloat16, device=DEV)
cs = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
L.call('vv_pack_conv_k4', L.ptr(w), L.ptr(wp), cin, cout, L.VV_BF16, cs)
x = torch.randn(B, 16, 16, 16, cin, device=DEV).to(torch.bfloat16)
ys = {k: torch.empty(B, 8, 8, 8, cout, dtype=torch.bfloat16, device=DEV) for k in ('32', '16')}
torch.cuda.synchronize()
def launch(kind):
    L.call('vv_conv3d_k4s2_direct_fwd', L.ptr(x), L.ptr(wp), L.ptr(sc), L.ptr(sh), L.ptr(ys[kind]), B, 16, cin, cout, 1, L.VV_BF16, cs)
N = 400
for rnd in range(3):
    for kind in ('32', '16'):
        os.environ['VV_CD_SHAPE'] = kind
        for i in range(20):
            launch(kind)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(N):
            launch(kind)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        print(json.dumps({'shape': kind, 'us_per_launch': round(1e6 * el / N, 2)}), flush=True)
print('max_abs_diff', (ys['32'].float() - ys['16'].float()).abs().max().item())

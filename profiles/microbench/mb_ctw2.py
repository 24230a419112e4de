"""Back-to-back throughput of the two D4 kernels on one and on two streams (no events between launches)."""
import ctypes, json, os, sys, time
import torch
import os; _R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, _R); sys.path.insert(0, os.path.join(_R, 'anytime-3d-reconstruction_amd'))
os.environ.setdefault('VOXVAE_TEST_HOOKS', '1')   # the kernel-form overrides live in lib/libvoxvae_hooks.so (voxvae/lib.py)
from voxvae import lib as L
L.load()
DEV = 'cuda:0'
B, cin, cout = 256, 128, 64
w = (torch.randn(4, 4, 4, cout, cin, device=DEV) / (8 * cin) ** 0.5).float().contiguous()
sc = torch.rand(cout, device=DEV) + 0.5; sh = torch.randn(cout, device=DEV) * 0.3
wf = torch.empty(64 * cin * cout, dtype=torch.bfloat16, device=DEV)
wk = torch.empty(64 * cin * cout, dtype=torch.bfloat16, device=DEV)
cs = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
L.call('vv_pack_convT_k4s2_frag', L.ptr(w), L.ptr(wf), cin, cout, cs)
L.call('vv_pack_convT_k4s2_skip', L.ptr(w), L.ptr(wk), cin, cout, cs)
xs = [torch.randn(B, 8, 8, 8, cin, device=DEV).to(torch.bfloat16) for _ in range(2)]
ys = [torch.empty(B, 16, 16, 16, cout, dtype=torch.bfloat16, device=DEV) for _ in range(2)]
torch.cuda.synchronize()
def launch(kind, i, st):
    if kind == 'halo':
        L.call('vv_convT3d_k4s2_direct_fwd', L.ptr(xs[i]), L.ptr(wf), L.ptr(sc), L.ptr(sh), L.ptr(ys[i]), B, 8, cin, cout, 1, L.VV_BF16, st)
    else:
        L.call('vv_convT3d_k4s2_whole_fwd', L.ptr(xs[i]), L.ptr(wk), L.ptr(sc), L.ptr(sh), L.ptr(ys[i]), B, 8, cin, cout, 1, L.VV_BF16, st)
streams = [torch.cuda.Stream() for _ in range(2)]
sp = [ctypes.c_void_p(s.cuda_stream) for s in streams]
N = 400
for rnd in range(2):
    for kind in ('halo', 'whole32', 'whole16', 'whole4'):
        os.environ['VV_CTW_SHAPE'] = kind[5:] if kind != 'halo' else '16'
        for ns in (1, 2):
            for i in range(20):
                launch(kind, i % ns, sp[i % ns])
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(N):
                launch(kind, i % ns, sp[i % ns])
            torch.cuda.synchronize()
            el = time.perf_counter() - t0
            print(json.dumps({'kernel': kind, 'streams': ns, 'us_per_launch': round(1e6 * el / N, 2)}), flush=True)

"""Back-to-back time of the first layer (first_conv_plane_kernel) under VV_FC_DBG ablations."""
import ctypes, json, os, sys, time
import torch
import os; _R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, _R); sys.path.insert(0, os.path.join(_R, 'anytime-3d-reconstruction_amd'))
from voxvae import lib as L
L.load()
DEV = 'cuda:0'
B = 256
x = (torch.rand(B, 32, 32, 32, 1, device=DEV) < 0.1).float().contiguous()
w = (torch.randn(4, 4, 4, 1, 64, device=DEV) / 8).float().contiguous()
sc = torch.rand(64, device=DEV) + 0.5; sh = torch.randn(64, device=DEV) * 0.3
wp = torch.empty(64, 64, dtype=torch.bfloat16, device=DEV)
cs = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
L.call('vv_pack_conv_k4', L.ptr(w), L.ptr(wp), 1, 64, L.VV_BF16, cs)
y = torch.empty(B, 16, 16, 16, 64, dtype=torch.bfloat16, device=DEV)
def launch():
    L.call('vv_conv3d_first_fwd', L.ptr(x), L.ptr(wp), L.ptr(sc), L.ptr(sh), L.ptr(y), B, 32, 64, 1, L.VV_BF16, cs)
N = 400
for dbg in [int(v) for v in (sys.argv[1:] or ['0'])]:
    if dbg: os.environ['VV_FC_DBG'] = str(dbg)
    else: os.environ.pop('VV_FC_DBG', None)
    for i in range(20): launch()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(N): launch()
    torch.cuda.synchronize()
    print(json.dumps({'dbg': dbg, 'us_per_launch': round(1e6 * (time.perf_counter() - t0) / N, 2)}), flush=True)

"""Back-to-back time of the last layer (final_bce_sweep_kernel + reduce) under VV_SW_DBG ablations."""
import ctypes, json, os, sys, time
import torch
import os; _R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, _R); sys.path.insert(0, os.path.join(_R, 'anytime-3d-reconstruction_amd'))
from voxvae import lib as L
lib = L.load()
DEV = 'cuda:0'
B = 256
x = torch.randn(B, 16, 16, 16, 64, device=DEV).to(torch.bfloat16)
w = (torch.randn(4, 4, 4, 1, 64, device=DEV) / 16).float().contiguous()
tgt = (torch.rand(B, 32, 32, 32, 1, device=DEV) < 0.1).float().contiguous()
probs = torch.empty(B, 32, 32, 32, 1, device=DEV)
stats = torch.empty(B, 4, device=DEV); met = torch.empty(4, device=DEV)
ws = torch.empty(max(lib.vv_convT3d_final_bce_workspace_bytes(B, 16), 16), dtype=torch.uint8, device=DEV)
cs = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
def launch():
    L.call('vv_convT3d_final_bce_metrics_fwd', L.ptr(x), L.ptr(w), L.ptr(tgt), L.ptr(probs), None, L.ptr(stats), L.ptr(met), B, 16, 64, 0.6, 1e-7, L.VV_BF16,
           L.ptr(ws), ws.numel(), cs)
N = 400
for dbg in [int(v) for v in (sys.argv[1:] or ['0'])]:
    if dbg: os.environ['VV_SW_DBG'] = str(dbg)
    else: os.environ.pop('VV_SW_DBG', None)
    for i in range(20): launch()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(N): launch()
    torch.cuda.synchronize()
    print(json.dumps({'dbg': dbg, 'us_per_launch': round(1e6 * (time.perf_counter() - t0) / N, 2)}), flush=True)

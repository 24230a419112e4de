import ctypes, json, sys
import torch
import os; _R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, _R); sys.path.insert(0, os.path.join(_R, 'anytime-3d-reconstruction_amd'))
from voxvae import lib as L
L.load()
DEV = 'cuda:0'
st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
B, K5, Lz, lin, n1 = 256, 4096, 64, 64, 4096
E = 2 * Lz
bt = torch.bfloat16
h = torch.randn(B, K5, device=DEV).to(bt); w5 = (torch.randn(E, K5, device=DEV) / 64).to(bt)
eps = torch.randn(B, Lz, device=DEV); wd = (torch.randn(lin, Lz, device=DEV) / 8).to(bt); w1 = (torch.randn(n1, lin, device=DEV) / 8).to(bt)
scd = torch.ones(lin, device=DEV); shd = torch.zeros(lin, device=DEV); sc1 = torch.ones(n1, device=DEV); sh1 = torch.zeros(n1, device=DEV)
z = torch.empty(B, Lz, device=DEV); zb = torch.empty(B, Lz, dtype=bt, device=DEV); kl = torch.empty(B, device=DEV); h1 = torch.empty(B, n1, dtype=bt, device=DEV)
ws = torch.empty(max(L.load().vv_latent_tail_workspace_bytes(B, K5, E, n1), 16), dtype=torch.uint8, device=DEV)
fn = lambda: L.call('vv_latent_tail_fwd', L.ptr(h), L.ptr(w5), None, L.ptr(eps), L.ptr(wd), L.ptr(scd), L.ptr(shd), L.ptr(w1), L.ptr(sc1), L.ptr(sh1),
                    None, L.ptr(z), L.ptr(zb), L.ptr(kl), L.ptr(h1), B, K5, E, Lz, lin, n1, 1, 1, L.VV_BF16, L.ptr(ws), ws.numel(), st())
for _ in range(10): fn()
torch.cuda.synchronize()
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(50)]
for a, b in ev:
    a.record(); fn(); b.record()
torch.cuda.synchronize()
t = sorted(a.elapsed_time(b) for a, b in ev)
print(json.dumps({'latent_tail_ms': round(t[len(t)//2], 4), 'min': round(t[0], 4), 'ws_MB': ws.numel() / 1e6}))

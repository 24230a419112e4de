"""D5 A/B: old (scratch/old/libvoxvae_old.so) vs new (in-tree) last-layer kernel, interleaved, back to back; also checks the outputs agree."""
import ctypes, json, os, sys, time
import torch
_R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, _R); sys.path.insert(0, os.path.join(_R, 'anytime-3d-reconstruction_amd'))
from voxvae import lib as L
new = L.load()
old = ctypes.CDLL(os.path.join(_R, 'scratch/old/libvoxvae_old.so'))
DEV = 'cuda:0'; B = 256
torch.manual_seed(0)
x = torch.randn(B, 16, 16, 16, 64, device=DEV).to(torch.bfloat16)
w = (torch.randn(4, 4, 4, 1, 64, device=DEV) / 16).float().contiguous()
tgt = (torch.rand(B, 32, 32, 32, 1, device=DEV) < 0.1).float().contiguous()
cs = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
def mk():
    return dict(probs=torch.empty(B, 32, 32, 32, 1, device=DEV), logits=torch.empty(B, 32, 32, 32, 1, device=DEV), stats=torch.empty(B, 4, device=DEV),
                met=torch.empty(4, device=DEV), ws=torch.empty(max(new.vv_convT3d_final_bce_workspace_bytes(B, 16), 16), dtype=torch.uint8, device=DEV))
def launch(lib, o, want_logits=False):
    f = lib.vv_convT3d_final_bce_metrics_fwd
    f.restype = ctypes.c_int
    rc = f(L.ptr(x), L.ptr(w), L.ptr(tgt), L.ptr(o['probs']), L.ptr(o['logits']) if want_logits else None, L.ptr(o['stats']), L.ptr(o['met']), B, 16, 64,
           ctypes.c_float(0.6), ctypes.c_float(1e-7), L.VV_BF16, L.ptr(o['ws']), ctypes.c_size_t(o['ws'].numel()), cs)
    assert rc == 0, rc
a, b = mk(), mk()
launch(old, a, True); launch(new, b, True); torch.cuda.synchronize()
print('logits equal', torch.equal(a['logits'], b['logits']), 'probs equal', torch.equal(a['probs'], b['probs']), 'stats max diff', (a['stats'] - b['stats']).abs().max().item(),
      'metrics', a['met'].tolist(), b['met'].tolist())
N = 300
for rep in range(3):
    for name, lib, o in (('old', old, a), ('new', new, b)):
        for i in range(20): launch(lib, o)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(N): launch(lib, o)
        torch.cuda.synchronize()
        print(json.dumps({'kernel': name, 'us_per_launch': round(1e6 * (time.perf_counter() - t0) / N, 2)}), flush=True)

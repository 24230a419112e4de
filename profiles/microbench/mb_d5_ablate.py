"""Back-to-back time of the last layer (kernel + metric reduce, batch 256) for the ablation builds of d5_ablate.py, interleaved in one
process.  usage: mb_d5_ablate.py <mask> ..."""
import ctypes, json, os, sys, time
import torch
_R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, _R); sys.path.insert(0, os.path.join(_R, 'anytime-3d-reconstruction_amd'))
from voxvae import lib as L
L.load()
DEV = 'cuda:0'; B = 256
cs = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
x = torch.randn(B, 16, 16, 16, 64, device=DEV).to(torch.bfloat16)
w = (torch.randn(4, 4, 4, 1, 64, device=DEV) / 16).float().contiguous()
tgt = (torch.rand(B, 32, 32, 32, 1, device=DEV) < 0.1).float().contiguous()
probs = torch.empty(B, 32, 32, 32, 1, device=DEV)
stats = torch.empty(B, 4, device=DEV); met = torch.empty(4, device=DEV)
ws = torch.empty(1 << 22, dtype=torch.uint8, device=DEV)
names = sys.argv[1:]
libs = {n: ctypes.CDLL(os.path.join(_R, 'scratch/abl/libabl_%s.so' % n)) for n in names}
def launch(n):
    f = libs[n].vv_convT3d_final_bce_metrics_fwd; f.restype = ctypes.c_int
    rc = f(L.ptr(x), L.ptr(w), L.ptr(tgt), L.ptr(probs), None, L.ptr(stats), L.ptr(met), B, 16, 64, ctypes.c_float(0.6), ctypes.c_float(1e-7), L.VV_BF16, L.ptr(ws), ctypes.c_size_t(ws.numel()), cs)
    assert rc == 0, rc
N = 300
res = {n: [] for n in names}
for rep in range(3):
    for n in names:
        for i in range(20): launch(n)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(N): launch(n)
        torch.cuda.synchronize(); res[n].append(round(1e6 * (time.perf_counter() - t0) / N, 2))
what = {'0': 'full kernel', '1': 'no exp/log/rcp', '2': 'no gather (LDS reads of P)', '4': 'no publish (LDS writes of P)', '8': 'no probability store',
        '16': 'no MFMA (operand reads kept)', '32': 'no MFMA, no operand reads', '7': 'no loss math, no gather, no publish', '63': 'DMA + target load + barriers only'}
for n in names: print(json.dumps({'abl': n, 'what': what.get(n), 'us': res[n]}), flush=True)

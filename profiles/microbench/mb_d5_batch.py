"""Back-to-back time of the last layer (sweep kernel + metric reduce) against the batch: 4 workgroups per sample, 3 resident per CU
(768 slots) -- batch 192 is exactly one round, batch 256 is 1.33.  usage: mb_d5_batch.py [batch ...]"""
import ctypes, json, os, sys, time
import torch
_R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, _R); sys.path.insert(0, os.path.join(_R, 'anytime-3d-reconstruction_amd'))
from voxvae import lib as L
lib = L.load()
DEV = 'cuda:0'
cs = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
BM = 512
x = torch.randn(BM, 16, 16, 16, 64, device=DEV).to(torch.bfloat16)
w = (torch.randn(4, 4, 4, 1, 64, device=DEV) / 16).float().contiguous()
tgt = (torch.rand(BM, 32, 32, 32, 1, device=DEV) < 0.1).float().contiguous()
probs = torch.empty(BM, 32, 32, 32, 1, device=DEV)
stats = torch.empty(BM, 4, device=DEV); met = torch.empty(4, device=DEV)
ws = torch.empty(max(lib.vv_convT3d_final_bce_workspace_bytes(BM, 16), 16), dtype=torch.uint8, device=DEV)
N = 300
for B in [int(v) for v in (sys.argv[1:] or ['64', '128', '192', '256', '320', '384', '512'])]:
    def launch():
        L.call('vv_convT3d_final_bce_metrics_fwd', L.ptr(x), L.ptr(w), L.ptr(tgt), L.ptr(probs), None, L.ptr(stats), L.ptr(met), B, 16, 64, 0.6, 1e-7,
               L.VV_BF16, L.ptr(ws), ws.numel(), cs)
    ts = []
    for rep in range(3):
        for i in range(20): launch()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(N): launch()
        torch.cuda.synchronize()
        ts.append(1e6 * (time.perf_counter() - t0) / N)
    t = min(ts)
    print(json.dumps({'batch': B, 'workgroups': 4 * B, 'us_per_launch': round(t, 2), 'us_per_64_samples': round(t * 64 / B, 2),
                      'TB_per_s': round(B * (16 ** 3 * 128 + 2 * 32 ** 3 * 4) / t / 1e6, 2)}), flush=True)

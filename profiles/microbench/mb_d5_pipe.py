"""D5 in the pipeline: each timed launch follows a D4 launch that has just WRITTEN its input (134 MB), as in the real step.
Variants: libs given on the command line (name=path); HIP events around the D5 launch only."""
import ctypes, json, os, sys, time
import torch
_R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, _R); sys.path.insert(0, os.path.join(_R, 'anytime-3d-reconstruction_amd'))
from voxvae import lib as L
new = L.load()
libs = {'tree': new}
for a in sys.argv[1:]:
    k, v = a.split('=')
    libs[k] = ctypes.CDLL(v)
DEV = 'cuda:0'; B = 256
torch.manual_seed(0)
x3 = torch.randn(B, 8, 8, 8, 128, device=DEV).to(torch.bfloat16)
w4 = (torch.randn(4, 4, 4, 64, 128, device=DEV) / 32).contiguous()
ww = torch.empty(64 * 128 * 64, dtype=torch.bfloat16, device=DEV)
cs = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
L.call('vv_pack_convT_k4s2_skip', L.ptr(w4), L.ptr(ww), 128, 64, cs)
sc = torch.rand(64, device=DEV) + 0.5; sh = torch.randn(64, device=DEV) * 0.3
x = torch.empty(B, 16, 16, 16, 64, dtype=torch.bfloat16, device=DEV)
w = (torch.randn(4, 4, 4, 1, 64, device=DEV) / 16).float().contiguous()
tgt = (torch.rand(B, 32, 32, 32, 1, device=DEV) < 0.1).float().contiguous()
probs = torch.empty(B, 32, 32, 32, 1, device=DEV); stats = torch.empty(B, 4, device=DEV); met = torch.empty(4, device=DEV)
ws = torch.empty(max(new.vv_convT3d_final_bce_workspace_bytes(B, 16), 16), dtype=torch.uint8, device=DEV)
def d4():
    L.call('vv_convT3d_k4s2_whole_fwd', L.ptr(x3), L.ptr(ww), L.ptr(sc), L.ptr(sh), L.ptr(x), B, 8, 128, 64, 1, L.VV_BF16, cs)
def d5(lib):
    f = lib.vv_convT3d_final_bce_metrics_fwd; f.restype = ctypes.c_int
    rc = f(L.ptr(x), L.ptr(w), L.ptr(tgt), L.ptr(probs), None, L.ptr(stats), L.ptr(met), B, 16, 64, ctypes.c_float(0.6), ctypes.c_float(1e-7), L.VV_BF16,
           L.ptr(ws), ctypes.c_size_t(ws.numel()), cs)
    assert rc == 0
N = 100
for rep in range(2):
    for name, lib in libs.items():
        for _ in range(10): d4(); d5(lib)
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(N)]
        torch.cuda.synchronize()
        for a, b in ev:
            d4(); a.record(); d5(lib); b.record()
        torch.cuda.synchronize()
        t = sorted(a.elapsed_time(b) for a, b in ev)
        # back to back, nothing in between
        t0 = time.perf_counter()
        for _ in range(N): d5(lib)
        torch.cuda.synchronize()
        print(json.dumps({'lib': name, 'after_D4_median_us': round(1e3 * t[N // 2], 1), 'after_D4_p10_us': round(1e3 * t[N // 10], 1),
                          'back_to_back_us': round(1e6 * (time.perf_counter() - t0) / N, 1)}), flush=True)

"""D4 (ctw16_kernel): workgroup length in shader cycles and the in-kernel clock, from one stamp pair around the kernel
(diagnostic builds scratch/libvv_clock_<name>.so written by ctw_clock.py).  usage: mb_ctw_clock.py name [name ...]"""
import ctypes, json, os, sys, time
import numpy as np
import torch
_R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, _R); sys.path.insert(0, os.path.join(_R, 'anytime-3d-reconstruction_amd'))
from voxvae import lib as L
L.load()
DEV = 'cuda:0'; B = 256
cs = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
torch.manual_seed(0)
w4 = (torch.randn(4, 4, 4, 64, 128, device=DEV) / 32).float().contiguous()
wk = torch.empty(64 * 128 * 64, dtype=torch.bfloat16, device=DEV)
L.call('vv_pack_convT_k4s2_skip', L.ptr(w4), L.ptr(wk), 128, 64, cs)
x4 = torch.randn(B, 8, 8, 8, 128, device=DEV).to(torch.bfloat16)
y = torch.empty(B, 16, 16, 16, 64, dtype=torch.bfloat16, device=DEV)
sc = torch.rand(128, device=DEV) + 0.5; sh = torch.randn(128, device=DEV) * 0.3
dbg = torch.zeros(B * 8 * 4, dtype=torch.int64, device=DEV)
os.environ['VV_CTW_STAMP_PTR'] = str(dbg.data_ptr())
for name in sys.argv[1:]:
    lib = ctypes.CDLL(os.path.join(_R, 'scratch/libvv_clock_%s.so' % name))
    f = lib.vv_convT3d_k4s2_whole_fwd; f.restype = ctypes.c_int
    a = (L.ptr(x4), L.ptr(wk), L.ptr(sc), L.ptr(sh), L.ptr(y), B, 8, 128, 64, 1, L.VV_BF16, cs)
    for rnd in range(2):
        t_end = time.perf_counter() + 2.5                       # >= 2 s of back-to-back launches: the clock the kernel holds under load
        n = 0
        while time.perf_counter() < t_end:
            for i in range(200):
                assert f(*a) == 0
            torch.cuda.synchronize(); n += 200
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(400):
            f(*a)
        torch.cuda.synchronize()
        us = 1e6 * (time.perf_counter() - t0) / 400
        d = dbg.cpu().numpy().reshape(B, 8, 4).astype(np.float64)
        cyc, real = d[..., 0], d[..., 1]
        ghz = cyc / real * 0.1
        print(json.dumps({'build': name, 'us_per_launch': round(us, 2), 'workgroup_cycles_median': float(np.median(cyc.max(axis=1))),
                          'in_kernel_clock_GHz_median': round(float(np.median(ghz)), 3), 'cycles_per_kstep_incl_prologue_epilogues': round(float(np.median(cyc.max(axis=1))) / 256, 1),
                          'ideal_cycles_per_kstep': 512}), flush=True)

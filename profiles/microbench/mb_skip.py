"""A/B timing of the skip-direct kernels against the implicit GEMM on the four small-grid layers (interleaved rounds, one process)."""
import ctypes
import json
import sys

import torch

import os; _R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, _R); sys.path.insert(0, os.path.join(_R, 'anytime-3d-reconstruction_amd'))
from voxvae import lib as L  # noqa: E402

L.load()
DEV = 'cuda:0'
st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)  # noqa: E731
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256


def timeit(fn, n=30):
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) for a, b in ev)
    return t[len(t) // 2], t[0]


def layer(kind, cin, cout):
    side = 8 if kind == 'conv' else 4
    oside = 4 if kind == 'conv' else 8
    x = torch.randn(B, side, side, side, cin, device=DEV).to(torch.bfloat16)
    wshape = (4, 4, 4, cin, cout) if kind == 'conv' else (4, 4, 4, cout, cin)
    w = (torch.randn(*wshape, device=DEV) / (64 * cin) ** 0.5).float().contiguous()
    sc = torch.ones(cout, device=DEV); sh = torch.zeros(cout, device=DEV)
    y0 = torch.empty(B, oside, oside, oside, cout, dtype=torch.bfloat16, device=DEV)
    y1 = torch.empty_like(y0)
    lib = L.load()
    if kind == 'conv':
        wp = torch.empty(cout, 64 * cin, dtype=torch.bfloat16, device=DEV)
        L.call('vv_pack_conv_k4', L.ptr(w), L.ptr(wp), cin, cout, L.VV_BF16, st())
        ws = torch.empty(max(lib.vv_conv3d_k4s2_workspace_bytes(B, side, cin, cout, L.VV_BF16), 16), dtype=torch.uint8, device=DEV)
        old = lambda: L.call('vv_conv3d_k4s2_fwd', L.ptr(x), L.ptr(wp), L.ptr(sc), L.ptr(sh), L.ptr(y0), B, side, cin, cout, 1, L.VV_BF16, L.ptr(ws), ws.numel(), st())  # noqa: E731
        wk = torch.empty(64 * cin * cout, dtype=torch.bfloat16, device=DEV)
        L.call('vv_pack_conv_k4_skip', L.ptr(w), L.ptr(wk), cin, cout, st())
        new = lambda: L.call('vv_conv3d_k4s2_skip_fwd', L.ptr(x), L.ptr(wk), L.ptr(sc), L.ptr(sh), L.ptr(y1), B, side, cin, cout, 1, L.VV_BF16, st())  # noqa: E731
        valid = (14 / 16) ** 3
    else:
        wp = torch.empty(8, cout, 8 * cin, dtype=torch.bfloat16, device=DEV)
        L.call('vv_pack_convT_k4s2', L.ptr(w), L.ptr(wp), cin, cout, L.VV_BF16, st())
        ws = torch.empty(max(lib.vv_convT3d_k4s2_workspace_bytes(B, side, cin, cout, L.VV_BF16), 16), dtype=torch.uint8, device=DEV)
        old = lambda: L.call('vv_convT3d_k4s2_fwd', L.ptr(x), L.ptr(wp), L.ptr(sc), L.ptr(sh), L.ptr(y0), B, side, cin, cout, 1, L.VV_BF16, L.ptr(ws), ws.numel(), st())  # noqa: E731
        wk = torch.empty(64 * cin * cout, dtype=torch.bfloat16, device=DEV)
        L.call('vv_pack_convT_k4s2_skip', L.ptr(w), L.ptr(wk), cin, cout, st())
        new = lambda: L.call('vv_convT3d_k4s2_skip_fwd', L.ptr(x), L.ptr(wk), L.ptr(sc), L.ptr(sh), L.ptr(y1), B, side, cin, cout, 1, L.VV_BF16, st())  # noqa: E731
        valid = (14 / 16) ** 3
    for _ in range(5):
        old(); new()
    torch.cuda.synchronize()
    diff = (y0.float() - y1.float()).abs().max().item()
    res = {}
    for rnd in range(3):
        for name, fn in (('igemm', old), ('skip', new)):
            res.setdefault(name, []).append(timeit(fn))
    dense = 2.0 * B * 64 * 64 * cin * cout
    out = {'layer': '%s %d->%d B=%d' % (kind, cin, cout, B), 'max_abs_diff_vs_igemm': diff}
    for name, v in res.items():
        med = sorted(m for m, _ in v)[1]
        out[name + '_ms'] = round(med, 4)
        out[name + '_valid_TF'] = round(dense * valid / (med * 1e-3) / 1e12, 1)
    print(json.dumps(out), flush=True)


import os
shapes = (('conv', 128, 256), ('convT', 256, 128)) if os.environ.get('MB_SHORT') else (('conv', 128, 256), ('convT', 256, 128), ('conv', 256, 512), ('convT', 512, 256))
for kind, cin, cout in shapes:
    layer(kind, cin, cout)

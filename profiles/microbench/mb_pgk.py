import os, subprocess, sys, json
# A/B over VV_PG_FORM in separate processes is not needed: getenv is read per call
import ctypes, time, torch
_R = '/root/repo'; sys.path.insert(0, _R); sys.path.insert(0, os.path.join(_R, 'anytime-3d-reconstruction_amd'))
from voxvae import lib as L
lib = L.load(); DEV = 'cuda:0'; B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
cs = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
torch.manual_seed(0)
def layer(kind, cin, cout):
    side = 4 if kind == 'conv' else 2; oside = 2 if kind == 'conv' else 4
    x = torch.randn(B, side, side, side, cin, device=DEV).to(torch.bfloat16)
    wshape = (4, 4, 4, cin, cout) if kind == 'conv' else (4, 4, 4, cout, cin)
    w = (torch.randn(*wshape, device=DEV) / (27 * cin) ** 0.5).float().contiguous()
    sc = torch.rand(cout, device=DEV) + 0.5; sh = torch.randn(cout, device=DEV) * 0.3
    wk = torch.empty(64 * cin * cout, dtype=torch.bfloat16, device=DEV)
    if kind == 'conv':
        L.call('vv_pack_conv_k4_skip', L.ptr(w), L.ptr(wk), cin, cout, cs)
        ws = torch.empty(max(lib.vv_conv3d_k4s2_pos_workspace_bytes(B, cin, cout), 16), dtype=torch.uint8, device=DEV)
        fn = lambda y: L.call('vv_conv3d_k4s2_pos_fwd', L.ptr(x), L.ptr(wk), L.ptr(sc), L.ptr(sh), L.ptr(y), B, side, cin, cout, 1, L.VV_BF16, L.ptr(ws), ws.numel(), cs)
    else:
        L.call('vv_pack_convT_k4s2_skip', L.ptr(w), L.ptr(wk), cin, cout, cs)
        ws = torch.empty(max(lib.vv_convT3d_k4s2_pos_workspace_bytes(B, cin, cout), 16), dtype=torch.uint8, device=DEV)
        fn = lambda y: L.call('vv_convT3d_k4s2_pos_fwd', L.ptr(x), L.ptr(wk), L.ptr(sc), L.ptr(sh), L.ptr(y), B, side, cin, cout, 1, L.VV_BF16, L.ptr(ws), ws.numel(), cs)
    ys = {}
    for form in ('slab', 'k'):
        os.environ['VV_PG_FORM'] = form
        y = torch.full((B, oside, oside, oside, cout), float('nan'), dtype=torch.bfloat16, device=DEV)
        fn(y); torch.cuda.synchronize(); ys[form] = y
    d = (ys['k'].float() - ys['slab'].float()).abs()
    out = {'layer': '%s %d->%d B=%d' % (kind, cin, cout, B), 'max_abs_diff': d.max().item(), 'nan': int(torch.isnan(ys['k'].float()).sum().item()), 'max_abs': ys['slab'].float().abs().max().item()}
    y = torch.empty_like(ys['k']); N = 300
    for rep in range(3):
        for form in ('slab', 'k'):
            os.environ['VV_PG_FORM'] = form
            for i in range(20): fn(y)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for i in range(N): fn(y)
            torch.cuda.synchronize(); out.setdefault(form + '_us', []).append(round(1e6 * (time.perf_counter() - t0) / N, 2))
    print(json.dumps(out), flush=True)
layer('conv', 256, 512); layer('convT', 512, 256)

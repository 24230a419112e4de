"""A/B of two library builds on the chip-wide MFMA layers at batch 256, back to back and interleaved: D4 (vv_convT3d_k4s2_whole_fwd),
E2 (vv_conv3d_k4s2_direct_fwd; E2fp8: vv_conv3d_k4s2_direct_fp8_fwd), E3 / D3 (vv_conv3d_k4s2_skip_fwd / vv_convT3d_k4s2_skip_fwd).  Outputs must be bit-identical.
usage: mb_ab_lib.py <other lib .so> [name]      (the tree's library is 'tree')"""
import ctypes, json, os, sys, time
import torch
_R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, _R); sys.path.insert(0, os.path.join(_R, 'anytime-3d-reconstruction_amd'))
from voxvae import lib as L
other = sys.argv[2] if len(sys.argv) > 2 else 'base'
libs = {other: ctypes.CDLL(os.path.join(_R, sys.argv[1])), 'tree': L.load()}
DEV = 'cuda:0'; B = 256
cs = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
torch.manual_seed(0)
sc = torch.rand(256, device=DEV) + 0.5; sh = torch.randn(256, device=DEV) * 0.3


def bf(*shape):
    return torch.randn(*shape, device=DEV).to(torch.bfloat16)


def packed(fn, cin, cout, *extra):
    w = (torch.randn(4, 4, 4, cin, cout, device=DEV) / (8 * cin) ** 0.5).float().contiguous()
    o = torch.empty(64 * cin * cout, dtype=torch.bfloat16, device=DEV)
    L.call(fn, L.ptr(w), L.ptr(o), *extra, cs)
    return o


cases = {}
# D4: 8^3 x 128 -> 16^3 x 64 (Keras transposed kernel [4,4,4,cout,cin])
w = packed('vv_pack_convT_k4s2_skip', 64, 128, 128, 64)
x = bf(B, 8, 8, 8, 128)
cases['D4'] = ('vv_convT3d_k4s2_whole_fwd', lambda y: (L.ptr(x), L.ptr(w), L.ptr(sc), L.ptr(sh), L.ptr(y), B, 8, 128, 64, 1, L.VV_BF16, cs), (B, 16, 16, 16, 64))
# E2: 16^3 x 64 -> 8^3 x 128
w2 = packed('vv_pack_conv_k4', 64, 128, 64, 128, L.VV_BF16)
x2 = bf(B, 16, 16, 16, 64)
cases['E2'] = ('vv_conv3d_k4s2_direct_fwd', lambda y: (L.ptr(x2), L.ptr(w2), L.ptr(sc), L.ptr(sh), L.ptr(y), B, 16, 64, 128, 1, L.VV_BF16, cs), (B, 8, 8, 8, 128))
# E3: 8^3 x 128 -> 4^3 x 256
w3 = packed('vv_pack_conv_k4_skip', 128, 256, 128, 256)
x3 = bf(B, 8, 8, 8, 128)
cases['E3'] = ('vv_conv3d_k4s2_skip_fwd', lambda y: (L.ptr(x3), L.ptr(w3), L.ptr(sc), L.ptr(sh), L.ptr(y), B, 8, 128, 256, 1, L.VV_BF16, cs), (B, 4, 4, 4, 256))
# D3: 4^3 x 256 -> 8^3 x 128
w4 = packed('vv_pack_convT_k4s2_skip', 128, 256, 256, 128)
x4 = bf(B, 4, 4, 4, 256)
cases['D3'] = ('vv_convT3d_k4s2_skip_fwd', lambda y: (L.ptr(x4), L.ptr(w4), L.ptr(sc), L.ptr(sh), L.ptr(y), B, 4, 256, 128, 1, L.VV_BF16, cs), (B, 8, 8, 8, 128))

# E2 in fp8 (the direct fp8 kernel of the 'wide' policy): e4m3fn input and weights, bf16 output
w2f = torch.empty(64 * 64 * 128, dtype=torch.uint8, device=DEV)
_w2 = (torch.randn(4, 4, 4, 64, 128, device=DEV) / 64).float().contiguous()
L.call('vv_pack_conv_k4', L.ptr(_w2), L.ptr(w2f), 64, 128, L.VV_FP8, cs)
x2f = torch.randn(B, 16, 16, 16, 64, device=DEV).to(torch.float8_e4m3fn)
cases['E2fp8'] = ('vv_conv3d_k4s2_direct_fp8_fwd', lambda y: (L.ptr(x2f), L.ptr(w2f), L.ptr(sc), L.ptr(sh), L.ptr(y), B, 16, 64, 128, 1, L.VV_BF16, cs), (B, 8, 8, 8, 128))

only = os.environ.get('AB_ONLY', '').split(',') if os.environ.get('AB_ONLY') else list(cases)
N = 300
res = {}
for name in only:
    fn, args, oshape = cases[name]
    ys = {k: torch.empty(*oshape, dtype=torch.bfloat16, device=DEV) for k in libs}
    fs = {}
    for k in libs:
        f = getattr(libs[k], fn); f.restype = ctypes.c_int
        fs[k] = f
        assert f(*args(ys[k])) == 0
    torch.cuda.synchronize()
    same = bool(torch.equal(ys[other], ys['tree']))
    t = {k: [] for k in libs}
    for rnd in range(4):
        for k in libs:
            a = args(ys[k])
            for i in range(30):
                fs[k](*a)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(N):
                fs[k](*a)
            torch.cuda.synchronize()
            t[k].append(1e6 * (time.perf_counter() - t0) / N)
    res[name] = {'bit_identical': same, 'us_per_launch': {k: [round(v, 2) for v in t[k]] for k in libs},
                 'tree_over_%s' % other: round(min(t['tree']) / min(t[other]), 4)}
    print(json.dumps({name: res[name]}), flush=True)

"""Back-to-back time of the widest decoder layer (batch 256) for the diagnostic builds of c4_ablate.py and for the tree's library,
interleaved in one process.  usage: mb_c4_ablate.py <tag> ...   ('tree16' / 'tree4' = the tree's hook library with VV_CTW_SHAPE=16 / 4)"""
import ctypes, json, os, sys, time
import torch
_R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, _R); sys.path.insert(0, os.path.join(_R, 'anytime-3d-reconstruction_amd'))
os.environ['VOXVAE_TEST_HOOKS'] = '1'
from voxvae import lib as L
DEV = 'cuda:0'; B, cin, cout = 256, 128, 64
cs = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
w = (torch.randn(4, 4, 4, cout, cin, device=DEV) / (8 * cin) ** 0.5).float().contiguous()
sc = torch.rand(cout, device=DEV) + 0.5; sh = torch.randn(cout, device=DEV) * 0.3
wk = torch.empty(64 * cin * cout, dtype=torch.bfloat16, device=DEV)
L.call('vv_pack_convT_k4s2_skip', L.ptr(w), L.ptr(wk), cin, cout, cs)
x = torch.randn(B, 8, 8, 8, cin, device=DEV).to(torch.bfloat16)
y = torch.empty(B, 16, 16, 16, cout, dtype=torch.bfloat16, device=DEV)
names = sys.argv[1:]
libs = {}
for n in names:
    if n.startswith('tree'):
        os.environ['VV_CTW_SHAPE'] = n[4:]
        libs[n] = L.load()
    else:
        libs[n] = ctypes.CDLL(os.path.join(_R, 'scratch/abl/libc4_%s.so' % n))
def launch(n):
    os.environ['VV_CTW_SHAPE'] = n[4:] if n.startswith('tree') else os.environ.get('C4_SHAPE', '4')
    f = libs[n].vv_convT3d_k4s2_whole_fwd; f.restype = ctypes.c_int
    f.argtypes = [ctypes.c_void_p] * 5 + [ctypes.c_int] * 6 + [ctypes.c_void_p]
    rc = f(L.ptr(x), L.ptr(wk), L.ptr(sc), L.ptr(sh), L.ptr(y), B, 8, cin, cout, 1, L.VV_BF16, cs)
    assert rc == 0, rc
N = 300
res = {n: [] for n in names}
for rep in range(3):
    for n in names:
        for i in range(20): launch(n)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(N): launch(n)
        torch.cuda.synchronize(); res[n].append(round(1e6 * (time.perf_counter() - t0) / N, 2))
for n in names: print(json.dumps({'build': n, 'us': res[n]}), flush=True)

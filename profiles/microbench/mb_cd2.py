"""Back-to-back throughput of the E2 kernel (conv_direct.hip): MFMA shapes 32x32x16 / 16x16x32 (8 waves, one workgroup per CU) and
'8' = 16x16x32 as two independent 4-wave workgroups per CU (conv_direct16h_kernel, round 4)."""
import ctypes, json, os, sys, time
import torch
import os; _R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, _R); sys.path.insert(0, os.path.join(_R, 'anytime-3d-reconstruction_amd'))
os.environ.setdefault('VOXVAE_TEST_HOOKS', '1')   # the kernel-form overrides live in lib/libvoxvae_hooks.so (voxvae/lib.py)
from voxvae import lib as L
L.load()
DEV = 'cuda:0'
B, cin, cout = 256, 64, 128
w = (torch.randn(4, 4, 4, cin, cout, device=DEV) / (64 * cin) ** 0.5).float().contiguous()
sc = torch.rand(cout, device=DEV) + 0.5; sh = torch.randn(cout, device=DEV) * 0.3
wp = torch.empty(cout, 64 * cin, dtype=torch.bfloat16, device=DEV)
cs = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
L.call('vv_pack_conv_k4', L.ptr(w), L.ptr(wp), cin, cout, L.VV_BF16, cs)
x = torch.randn(B, 16, 16, 16, cin, device=DEV).to(torch.bfloat16)
ys = {k: torch.empty(B, 8, 8, 8, cout, dtype=torch.bfloat16, device=DEV) for k in ('32', '16', '8')}
torch.cuda.synchronize()
def launch(kind):
    L.call('vv_conv3d_k4s2_direct_fwd', L.ptr(x), L.ptr(wp), L.ptr(sc), L.ptr(sh), L.ptr(ys[kind]), B, 16, cin, cout, 1, L.VV_BF16, cs)
N = 400
VARIANTS = [('32', {}), ('16', {}), ('8', {}), ('8', {'VV_CDH_STAGGER': '0'}), ('8', {'VV_CDH_STAGGER': '2'}), ('8', {'VV_CDH_STAGGER': '10'}), ('8', {'VV_CDH_ABL': '1'})]
for rnd in range(2):
    for kind, env in VARIANTS:
        os.environ['VV_CD_SHAPE'] = kind
        for k_ in ('VV_CDH_STAGGER', 'VV_CDH_ABL'):
            os.environ.pop(k_, None)
        os.environ.update(env)
        for i in range(20):
            launch(kind)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(N):
            launch(kind)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        print(json.dumps({'shape': kind, 'env': env, 'us_per_launch': round(1e6 * el / N, 2)}), flush=True)
os.environ.pop('VV_CDH_ABL', None); os.environ['VV_CD_SHAPE'] = '8'; launch('8'); torch.cuda.synchronize()
print('max_abs_diff 32 vs 16', (ys['32'].float() - ys['16'].float()).abs().max().item(), ' 8 vs 16', (ys['8'].float() - ys['16'].float()).abs().max().item())

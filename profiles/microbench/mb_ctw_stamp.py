"""D4 (ctw16_kernel) with in-kernel s_memtime stamps (diagnostic build scratch/libvv_stamp.so from ctw_stamp.py): where a chunk
spends its cycles.  Four stamps per chunk at the points where lgkmcnt is 0 anyway (profiles/microbench/ctw_stamp.py writes the diagnostic source):
0 | group P: 16 MFMAs + the 8 reads of Q + next tap's addresses, wait Q | A | vmcnt wait (weight chunk issued two chunks ago) | B |
s_barrier | C | group Q: 16 MFMAs + LDS-DMA piece + the 8 reads of P, wait P | 0.
Shares, not lengths (the stamps cost ~40 cycles each and fence the schedule)."""
import ctypes, json, os, sys
import numpy as np
import torch
_R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, _R); sys.path.insert(0, os.path.join(_R, 'anytime-3d-reconstruction_amd'))
from voxvae import lib as L
tree = L.load()
lib = ctypes.CDLL(os.path.join(_R, 'scratch/libvv_stamp.so'))
DEV = 'cuda:0'; B = 256
cs = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
torch.manual_seed(0)
w4 = (torch.randn(4, 4, 4, 64, 128, device=DEV) / 32).float().contiguous()
wk = torch.empty(64 * 128 * 64, dtype=torch.bfloat16, device=DEV)
L.call('vv_pack_convT_k4s2_skip', L.ptr(w4), L.ptr(wk), 128, 64, cs)
x4 = torch.randn(B, 8, 8, 8, 128, device=DEV).to(torch.bfloat16)
y_ref = torch.empty(B, 16, 16, 16, 64, dtype=torch.bfloat16, device=DEV)
y = torch.empty_like(y_ref)
sc = torch.rand(128, device=DEV) + 0.5; sh = torch.randn(128, device=DEV) * 0.3
dbg = torch.zeros(B * 8 * 8, dtype=torch.int64, device=DEV)
f = lib.vv_convT3d_k4s2_whole_fwd; f.restype = ctypes.c_int
g = tree.vv_convT3d_k4s2_whole_fwd
assert g(L.ptr(x4), L.ptr(wk), L.ptr(sc), L.ptr(sh), L.ptr(y_ref), B, 8, 128, 64, 1, L.VV_BF16, cs) == 0
for i in range(300):                                             # clocks settle under load
    assert f(L.ptr(x4), L.ptr(wk), L.ptr(sc), L.ptr(sh), L.ptr(y), B, 8, 128, 64, 1, L.VV_BF16, cs) == 0
os.environ['VV_CTW_STAMP_PTR'] = str(dbg.data_ptr())
for i in range(3):
    assert f(L.ptr(x4), L.ptr(wk), L.ptr(sc), L.ptr(sh), L.ptr(y), B, 8, 128, 64, 1, L.VV_BF16, cs) == 0
torch.cuda.synchronize()
assert torch.equal(y, y_ref), 'the stamped build must compute the same tile'
d = dbg.cpu().numpy().reshape(B, 8, 8).astype(np.float64)
vm, bar, gp, pro, loop, gq = d[..., 0], d[..., 1], d[..., 2], d[..., 3], d[..., 4], d[..., 7]
work = gp + gq
nchunk = 128
out = {
    'chunks_per_wave': nchunk,
    'cycles_per_chunk_mean': float(loop.mean() / nchunk),
    'share_vmcnt_wait': float((vm / loop).mean()), 'share_barrier': float((bar / loop).mean()), 'share_work': float((work / loop).mean()),
    'cycles_per_chunk': {'vmcnt_wait': float(vm.mean() / nchunk), 'barrier': float(bar.mean() / nchunk), 'work': float(work.mean() / (nchunk - 1))},
    'prologue_cycles_mean': float(pro.mean()), 'prologue_share_of_kernel': float((pro / (pro + loop)).mean()),
    'per_wave_share_barrier': [float((bar[:, w] / loop[:, w]).mean()) for w in range(8)],
    'per_wave_share_vmcnt': [float((vm[:, w] / loop[:, w]).mean()) for w in range(8)],
    'per_wave_groupP_cycles_per_chunk': [float(gp[:, w].mean() / nchunk) for w in range(8)],
    'per_wave_groupQ_cycles_per_chunk': [float(gq[:, w].mean() / (nchunk - 1)) for w in range(8)],
    'per_wave_work_cycles_per_chunk': [float(work[:, w].mean() / (nchunk - 1)) for w in range(8)],
    'kernel_cycles_first_to_last_stamp': float(d[..., 6].max() - d[..., 5].min()),
    'workgroup_cycles_mean': float((d[..., 6].max(axis=1) - d[..., 5].min(axis=1)).mean()),
    'ideal_mfma_cycles_per_chunk_two_waves_per_simd': 2 * 32 * 16,
}
print(json.dumps(out, indent=1))

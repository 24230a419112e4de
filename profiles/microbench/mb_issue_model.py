"""Runs scratch/abl/libissue.so (profiles/microbench/issue_model.hip built by hipcc -shared): cycles per MFMA slot with fillers."""
import ctypes, os, json
import torch  # noqa: F401  (one HIP runtime per process)
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
lib = ctypes.CDLL(os.path.join(R, 'scratch/abl/libissue.so'))
res = (ctypes.c_double * 128)()
n = lib.issue_model(res)
labels = []
def Rr(s, k, f):
    labels.append((s, k, f, 1)); labels.append((s, k, f, 2))
Rr(16,0,0); Rr(32,0,0)
for f in (1,2,3,4): Rr(16,1,f)
for f in (1,2,3,4): Rr(32,1,f)
for f in (1,2): Rr(16,2,f)
for f in (1,2): Rr(32,2,f)
for f in (1,2): Rr(16,3,f)
for f in (1,2): Rr(32,3,f)
for f in (1,2,3): Rr(16,4,f)
for f in (1,2,3): Rr(32,4,f)
kinds = {0: 'none', 1: 'v_add', 2: 'v_exp', 3: 'ds_read_b128', 4: 'xor+ds_read / 2 v_add mix'}
for (s, k, f, w), v in zip(labels, list(res)[:n]):
    print(json.dumps({'mfma': '16x16x32' if s == 16 else '32x32x16', 'filler': kinds[k], 'per_16x16_slot': f, 'waves_per_simd': w, 'cycles_per_slot': round(v, 2)}))

import ctypes, os, torch
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
lib = ctypes.CDLL(os.path.join(R, 'scratch/abl/libissue3.so'))
res = (ctypes.c_double * 16)()
n = lib.issue_model3(res)
labels = ['unroll 1 (0.8 KB) fma x2', 'unroll 16 (12 KB)', 'unroll 32 (25 KB)', 'unroll 64 (49 KB)', 'unroll 96 (74 KB)', 'unroll 128 (98 KB)', 'unroll 64, no fillers (16 KB)', 'unroll 128, no fillers (33 KB)']
for l, v in zip(labels, list(res)[:n]): print(l, round(v, 2))

res = (ctypes.c_double * 16)()
n = lib.issue_model3t(res)
for i, f in enumerate((0, 1, 2, 3, 4)):
    print('v_fma x%d per 16x16x32 gap: %.2f cycles per MFMA slot (s_memtime), clock held %.2f GHz (cycles / hipEvent time) -> %.2f ns per slot' % (f, res[2 * i], res[2 * i + 1], res[2 * i] / res[2 * i + 1]))

import ctypes, os, torch
lib = ctypes.CDLL('/root/repo/scratch/abl/libissue2.so')
res = (ctypes.c_double * 16)()
n = lib.issue_model2(res)
labels = ['none', 'fma x1', 'fma x2', 'fma x3', 'read-mix x1', 'read-mix x2', 'read-mix x3']
for l, v in zip(labels, list(res)[:n]): print(l, round(v, 2))

"""Writes scratch/abl/libsk_<mask>.so: the tree's skip_direct.hip with parts of sd_kernel compiled out (-DABL=<mask>, wrong results, timing
only), linked against the tree's other objects; then `gpurun -- python profiles/microbench/mb_sd_ablate.py 0 1 2 3 4 8 12 15`.  Mask bits:
1 no MFMA, 2 no fragment reads, 4 no LDS-DMA inside the loops (the prologue's stages stay), 8 no vmcnt(0) + barrier inside the loops.
DESIGN.md section 7 (3c)."""
import os, subprocess, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PKG = os.path.join(R, 'anytime-3d-reconstruction_amd')
s = open(os.path.join(PKG, 'csrc', 'skip_direct.hip')).read()


def sub(old, new, n=1):
    global s
    assert s.count(old) == n, (s.count(old), old[:60])
    s = s.replace(old, new)


sub('#include "common.h"', '#include "%s"\n#ifndef ABL\n#define ABL 0\n#endif' % os.path.join(PKG, 'csrc', 'common.h'))
sub("        if (vmask & (1 << i)) {\n", "        if ((vmask & (1 << i)) && !(ABL & 1)) {\n")
sub('#define SD_RD(DST, ADDR, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(DST) : "v"(ADDR), "n"(OFF) : "memory")',
    '#define SD_RD(DST, ADDR, OFF) do { if (!(ABL & 2)) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(DST) : "v"(ADDR), "n"(OFF) : "memory"); '
    'else asm volatile("" : "=v"(DST)); } while (0)')
sub("if (u + 2 < nunits) issue_w(u + 2, qb, cb);", "if (u + 2 < nunits && !(ABL & 4)) issue_w(u + 2, qb, cb);", 2)
sub("if (T + 2 < ntiles) issue_a(T + 2, qc, cc);", "if (T + 2 < ntiles && !(ABL & 4)) issue_a(T + 2, qc, cc);")
sub("if (u + 2 < nunits) issue_w(u + 2, 0, 0);", "if (u + 2 < nunits && !(ABL & 4)) issue_w(u + 2, 0, 0);")
sub("if (T + 2 < ntiles) issue_a(T + 2, 0, 0);", "if (T + 2 < ntiles && !(ABL & 4)) issue_a(T + 2, 0, 0);")
sub('                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");\n                __builtin_amdgcn_s_barrier();\n',
    '                if (!(ABL & 8)) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); }\n', 3)
out = os.path.join(R, 'scratch', 'abl'); os.makedirs(out, exist_ok=True)
open(os.path.join(out, 'skip_abl.hip'), 'w').write(s)
objs = [os.path.join(PKG, 'lib', f) for f in sorted(os.listdir(os.path.join(PKG, 'lib'))) if f.endswith('.o') and f != 'skip_direct.o']
for m in (sys.argv[1:] or ['0', '1', '2', '3', '4', '8', '12', '15']):
    o = os.path.join(out, 'sk_%s.o' % m)
    subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-fvisibility=hidden', '-Wno-unused-function',
                           '-Wno-shift-op-parentheses', '-DABL=%s' % m, '-c', os.path.join(out, 'skip_abl.hip'), '-o', o])
    subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-shared', '-fPIC', '-o', os.path.join(out, 'libsk_%s.so' % m), o] + objs)
    print('built', m)

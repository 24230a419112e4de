"""Writes scratch/abl/libpg_<mask>.so: the tree's posgemm.hip with parts of pg_kernel compiled out (-DABL=<mask>, wrong results, timing
only), linked against the tree's other objects; then `gpurun -- python profiles/microbench/mb_pg_ablate.py 0 1 2 3 4 8 12 15`.  Mask bits:
1 no MFMA, 2 no fragment reads, 4 no LDS-DMA inside the loop, 8 no counted wait + barrier inside the loop.  DESIGN.md section 7 (3c)."""
import os, subprocess, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PKG = os.path.join(R, 'anytime-3d-reconstruction_amd')
s = open(os.path.join(PKG, 'csrc', 'posgemm.hip')).read()


def sub(old, new, n=1):
    global s
    assert s.count(old) == n, (s.count(old), old[:50])
    s = s.replace(old, new)


sub('#include "common.h"', '#include "%s"\n#ifndef ABL\n#define ABL 0\n#endif' % os.path.join(PKG, 'csrc', 'common.h'))
sub("    auto mma = [&](const pg_u4 *F) {\n", "    auto mma = [&](const pg_u4 *F) {\n        if (ABL & 1) return;\n")
sub("    auto rd = [&](pg_u4 *F, int ks, unsigned stoff) {\n",
    "    auto rd = [&](pg_u4 *F, int ks, unsigned stoff) {\n        if (ABL & 2) { asm volatile(\"\" : \"=v\"(F[0]), \"=v\"(F[1]), \"=v\"(F[2]), \"=v\"(F[3])); return; }\n")
sub("        if (i + 3 < nmine) issue(buf);", "        if (i + 3 < nmine && !(ABL & 4)) issue(buf);")
sub("""        if (i + 2 < nmine) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPC) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (i + 3 < nmine""", """        if (!(ABL & 8)) {
        if (i + 2 < nmine) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPC) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier(); }
        if (i + 3 < nmine""")
out = os.path.join(R, 'scratch', 'abl'); os.makedirs(out, exist_ok=True)
open(os.path.join(out, 'pos_abl.hip'), 'w').write(s)
objs = [os.path.join(PKG, 'lib', f) for f in sorted(os.listdir(os.path.join(PKG, 'lib'))) if f.endswith('.o') and f != 'posgemm.o']
for m in (sys.argv[1:] or ['0', '1', '2', '3', '4', '8', '12', '15']):
    o = os.path.join(out, 'pg_%s.o' % m)
    subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-fvisibility=hidden', '-Wno-unused-function',
                           '-Wno-shift-op-parentheses', '-DABL=%s' % m, '-c', os.path.join(out, 'pos_abl.hip'), '-o', o])
    subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-shared', '-fPIC', '-o', os.path.join(out, 'libpg_%s.so' % m), o] + objs)
    print('built', m)

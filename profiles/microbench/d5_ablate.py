"""Writes scratch/abl/libabl_<mask>.so: the tree's first_last.hip with parts of final_bce_sweep_kernel compiled out (-DABL=<mask>, wrong
results, timing only) linked against the tree's other objects.  Run on the build box (CPU), then `gpurun -- python
profiles/microbench/mb_d5_ablate.py 0 1 2 4 8 16 32 7 63`.  Mask bits: 1 no exp / log / rcp, 2 no gather (LDS reads of P), 4 no publish
(LDS writes of P), 8 no probability store, 16 no MFMA (operand reads kept), 32 no MFMA and no operand reads.  DESIGN.md section 4f."""
import os, subprocess, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PKG = os.path.join(R, 'anytime-3d-reconstruction_amd')
out = os.path.join(R, 'scratch', 'abl'); os.makedirs(out, exist_ok=True)
s = open(os.path.join(PKG, 'csrc', 'first_last.hip')).read()


def sub(old, new):
    global s
    assert s.count(old) == 1, old[:60]
    s = s.replace(old, new)


sub('#include "common.h"', '#include "%s"\n#ifndef ABL\n#define ABL 0\n#endif' % os.path.join(PKG, 'csrc', 'common.h'))
sub("""                p[e] = __builtin_amdgcn_rcpf(1.0f + __expf(-l[e]));
                const float q = fminf(fmaxf(p[e], epsilon), hi);
                bce -= gamma * yy[e] * __logf(q) + (1.0f - gamma) * (1.0f - yy[e]) * __logf(1.0f - q);""",
    """                if (ABL & 1) { p[e] = l[e]; bce += l[e] * yy[e]; } else {
                p[e] = __builtin_amdgcn_rcpf(1.0f + __expf(-l[e]));
                const float q = fminf(fmaxf(p[e], epsilon), hi);
                bce -= gamma * yy[e] * __logf(q) + (1.0f - gamma) * (1.0f - yy[e]) * __logf(1.0f - q); }""")
sub("""        for (int ah = 0; ah < 2; ++ah) {
            const int zh = mh + ph - ah + 1, th = 1 - ph + 2 * ah;
            const int q = sl * 4 + th, rowb = (zh * 10 + mw) * SW_PP;""",
    """        for (int ah = 0; ah < ((ABL & 2) ? 0 : 2); ++ah) {
            const int zh = mh + ph - ah + 1, th = 1 - ph + 2 * ah;
            const int q = sl * 4 + th, rowb = (zh * 10 + mw) * SW_PP;""")
sub("            if (row < SW_ROWS) {\n                const int zwk = (row - (row / 10) * 10) & 7;",
    "            if (row < SW_ROWS && !(ABL & 4)) {\n                const int zwk = (row - (row / 10) * 10) & 7;")
sub("            if (probs) *reinterpret_cast<float2 *>(probs + o) = make_float2(p[0], p[1]);\n            if (logits) *reinterpret_cast<float2 *>(logits + o) = make_float2(l[0], l[1]);\n        }\n        acc[0] = acc_next[0];",
    "            if (probs && !(ABL & 8)) *reinterpret_cast<float2 *>(probs + o) = make_float2(p[0], p[1]);\n            if (logits) *reinterpret_cast<float2 *>(logits + o) = make_float2(l[0], l[1]);\n        }\n        acc[0] = acc_next[0];")
sub("""                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8 *>(&fb[ks]),
                                                                 *reinterpret_cast<const bf16x8 *>(&fa), acc[j], 0, 0, 0);
            }
        }
    };""", """                if (ABL & 16) acc[j][ks] += __builtin_bit_cast(float, fa.x); else
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8 *>(&fb[ks]),
                                                                 *reinterpret_cast<const bf16x8 *>(&fa), acc[j], 0, 0, 0);
            }
        }
    };""")
sub("        mfma_plane(nexth, acc_next);", "        if (ABL & 32) { acc_next[0] = acc[0]; acc_next[1] = acc[1]; } else mfma_plane(nexth, acc_next);")
src = os.path.join(out, 'first_last_abl.hip')
open(src, 'w').write(s)
objs = [os.path.join(PKG, 'lib', f) for f in sorted(os.listdir(os.path.join(PKG, 'lib'))) if f.endswith('.o') and f != 'first_last.o']
for m in (sys.argv[1:] or ['0', '1', '2', '4', '8', '16', '32', '7', '63']):
    o = os.path.join(out, 'fl_%s.o' % m)
    subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-fvisibility=hidden', '-Wno-unused-function',
                           '-Wno-shift-op-parentheses', '-DABL=%s' % m, '-c', src, '-o', o])
    subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-shared', '-fPIC', '-o', os.path.join(out, 'libabl_%s.so' % m), o] + objs)
    print('built', m)

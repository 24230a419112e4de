"""Writes scratch/csrc_clock/: ctw16_kernel with ONE pair of stamps around the whole kernel (s_memtime and s_memrealtime): the
workgroup's length in shader cycles and the in-kernel clock (MI355X_MICROARCH.md, DVFS give-back item 6).  No stamp in the loop.
usage: ctw_clock.py [git rev of csrc/convt_whole.hip, default: the tree's]"""
import os, shutil, subprocess, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
src, dst = os.path.join(R, 'anytime-3d-reconstruction_amd/csrc'), os.path.join(R, 'scratch/csrc_clock')
shutil.rmtree(dst, ignore_errors=True)
shutil.copytree(src, dst)
p = os.path.join(dst, 'convt_whole.hip')
if len(sys.argv) > 1:
    open(p, 'w').write(subprocess.check_output(['git', '-C', R, 'show', sys.argv[1] + ':anytime-3d-reconstruction_amd/csrc/convt_whole.hip']).decode())
s = open(p).read()
i = s.index('void ctw16_kernel(')
head, s = s[:i], s[i:]


def rep(old, new):
    global s
    assert old in s, old[:50]
    s = s.replace(old, new, 1)


rep('__bf16 *__restrict__ y, int npar) {', '__bf16 *__restrict__ y, int npar, unsigned long long *dbg) {')
rep('    const unsigned lds0 = (unsigned)(unsigned long long)(lptr_t)smem;\n',
    '    const unsigned lds0 = (unsigned)(unsigned long long)(lptr_t)smem;\n'
    '    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();\n    __builtin_amdgcn_s_waitcnt(0xC07F);\n')
rep('    cw_wait_vm<0>();\n}', '''    cw_wait_vm<0>();
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (dbg && lane == 0) {
        unsigned long long *o = dbg + ((size_t)blockIdx.x * 8 + wave) * 4;
        o[0] = c1 - c0; o[1] = r1 - r0; o[2] = c0; o[3] = r0;
    }
}''')
rep('reinterpret_cast<__bf16 *>(y), 8 / ps);\n        else', '''reinterpret_cast<__bf16 *>(y), 8 / ps,
                      getenv("VV_CTW_STAMP_PTR") ? reinterpret_cast<unsigned long long *>(strtoull(getenv("VV_CTW_STAMP_PTR"), nullptr, 0)) : nullptr);
        else''')
open(p, 'w').write(head + s)
print('wrote', p)

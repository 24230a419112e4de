"""Writes the diagnostic (stamped) copy of csrc/ into scratch/csrc_stamp/: ctw16_kernel with four s_memtime stamps per chunk at the
points where lgkmcnt is 0 anyway.  Build: hipcc ... -shared -o scratch/libvv_stamp.so scratch/csrc_stamp/*.hip; run mb_ctw_stamp.py."""
import os, shutil
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
src, dst = os.path.join(R, 'anytime-3d-reconstruction_amd/csrc'), os.path.join(R, 'scratch/csrc_stamp')
shutil.rmtree(dst, ignore_errors=True)
shutil.copytree(src, dst)
p = os.path.join(dst, 'convt_whole.hip')
s = open(p).read()


def rep(old, new, count=1):
    global s
    assert old in s, old[:60]
    s = s.replace(old, new, count)


rep('#define CW16_MFMA(F) ', '''#define STAMP(T)                                                                                                                   \\
    do {                                                                                                                            \\
        __builtin_amdgcn_sched_barrier(0);                                                                                          \\
        asm volatile("s_memtime %0\\n\\ts_waitcnt lgkmcnt(0)" : "=s"(T) : : "memory");                                                \\
        __builtin_amdgcn_sched_barrier(0);                                                                                          \\
    } while (0)
#define CW16_MFMA(F) ''')
i = s.index('void ctw16_kernel(')
head, s = s[:i], s[i:]
rep('__bf16 *__restrict__ y, int npar) {', '__bf16 *__restrict__ y, int npar, unsigned long long *dbg) {')
rep('    const unsigned lds0 = (unsigned)(unsigned long long)(lptr_t)smem;\n',
    '    const unsigned lds0 = (unsigned)(unsigned long long)(lptr_t)smem;\n'
    '    unsigned long long t0 = 0, tA = 0, tB = 0, tC = 0, sVm = 0, sBar = 0, sP = 0, sQ = 0, tStart = 0, tLoop = 0;\n    STAMP(tStart);\n')
rep('    CW16_LD(P, ua, ws);\n#pragma unroll 1', '    STAMP(tLoop);\n    CW16_LD(P, ua, ws);\n#pragma unroll 1')
rep('            CW16_WAIT(P, 0);\n            CW16_SB;', '            CW16_WAIT(P, 0);\n            STAMP(t0);\n            if (tC) sQ += t0 - tC;')
rep('            CW16_WAIT(Q, 0);\n            if (J < 2 && pi > 0) cw_wait_vm<9>();\n            else cw_wait_vm<1>();\n            __builtin_amdgcn_s_barrier();\n',
    '            CW16_WAIT(Q, 0);\n            STAMP(tA);\n            sP += tA - t0;\n            if (J < 2 && pi > 0) cw_wait_vm<9>();\n            else cw_wait_vm<1>();\n'
    '            STAMP(tB);\n            __builtin_amdgcn_s_barrier();\n            STAMP(tC);\n            sVm += tB - tA;\n            sBar += tC - tB;\n')
rep('    CW16_WAIT(P, 0);\n    cw_wait_vm<0>();\n}', '''    CW16_WAIT(P, 0);
    cw_wait_vm<0>();
    unsigned long long tEnd;
    STAMP(tEnd);
    if (dbg && lane == 0) {
        unsigned long long *o = dbg + ((size_t)blockIdx.x * 8 + wave) * 8;
        o[0] = sVm; o[1] = sBar; o[2] = sP; o[3] = tLoop - tStart; o[4] = tEnd - tLoop; o[5] = tStart; o[6] = tEnd; o[7] = sQ;
    }
}''')
rep('reinterpret_cast<__bf16 *>(y), 8 / ps);\n        else', '''reinterpret_cast<__bf16 *>(y), 8 / ps,
                      getenv("VV_CTW_STAMP_PTR") ? reinterpret_cast<unsigned long long *>(strtoull(getenv("VV_CTW_STAMP_PTR"), nullptr, 0)) : nullptr);
        else''')
open(p, 'w').write(head + s)
print('wrote', p)

// Closer to the real D4 stream: one wave per SIMD, 32 accumulators (AGPRs via "a"-constrained reads at the end), A/B operands that
// change with every MFMA (4 x 8 fragment registers), fillers: v_fma with three VGPR sources / v_xor+ds_read pairs (address one gap ahead).
#include <hip/hip_runtime.h>
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
#define SB __builtin_amdgcn_sched_barrier(0)

template <int KIND, int F>
__global__ __launch_bounds__(256, 1) void k2(unsigned long long *out, int iters, float seed) {
    __shared__ __attribute__((aligned(256))) char lds[65536];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) reinterpret_cast<float *>(lds)[i] = seed * i;
    __syncthreads();
    u32x4 fa[4], fb[8];
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 4; ++e) fa[i][e] = 0x3f803f80u + i + e + lane;
    for (int i = 0; i < 8; ++i) for (int e = 0; e < 4; ++e) fb[i][e] = 0x3f003f00u + i * 3 + e + lane;
    f32x4 acc[4][8];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
    float v[8], s1 = seed * 1.5f, s2 = seed * 0.25f;
    for (int i = 0; i < 8; ++i) v[i] = seed + i;
    u32x4 r[4] = {};
    const unsigned la = (unsigned)(unsigned long long)(__attribute__((address_space(3))) void *)lds + lane * 16;
    unsigned xa[2] = {la, la};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 32; ++m) {
            acc[m >> 3][m & 7] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8 *>(&fa[m >> 3]), *reinterpret_cast<const bf16x8 *>(&fb[m & 7]), acc[m >> 3][m & 7], 0, 0, 0);
            SB;
#pragma unroll
            for (int f = 0; f < F; ++f) {
                const int j = (m * 3 + f) & 7;
                if (KIND == 1) { asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[j]) : "v"(s1), "v"(s2)); }
                else if (KIND == 2) {       // every second gap: a read with the address prepared in the previous gap + the next xor
                    if ((m & 1) == 0 && f == 0) { asm volatile("ds_read_b128 %0, %1" : "=&v"(r[(m >> 1) & 3]) : "v"(xa[(m >> 1) & 1]) : "memory"); asm volatile("v_xor_b32 %0, %1, %2" : "=v"(xa[((m >> 1) + 1) & 1]) : "v"(la), "v"(64u * ((m >> 1) & 3))); }
                    else if (f > 0 || (m & 1)) { asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[j]) : "v"(s1), "v"(s2)); }
                }
            }
            SB;
        }
        if (KIND == 2) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]) : : "memory");
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 8; ++j) { float t; asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(t) : "a"(acc[i][j][0])); s += t; }
    for (int i = 0; i < 8; ++i) s += v[i];
    s += __builtin_bit_cast(float, r[0][0] ^ r[1][1] ^ r[2][2] ^ r[3][3]);
    if (lane == 0) out[(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 2] = t1 - t0;
    if (s == 12345.678f) out[1] = 1;
}
template <int KIND, int F>
static double run2(unsigned long long *out) {
    const int iters = 1000;
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL((k2<KIND, F>), dim3(256), dim3(256), 0, 0, out, iters, 1.0f); hipDeviceSynchronize(); }
    unsigned long long h[8];
    hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    return (double)h[0] / (iters * 32.0);
}
extern "C" __attribute__((visibility("default"))) int issue_model2(double *res) {
    unsigned long long *out;
    hipMalloc(&out, 1 << 16);
    int n = 0;
    res[n++] = run2<0, 0>(out);
    res[n++] = run2<1, 1>(out); res[n++] = run2<1, 2>(out); res[n++] = run2<1, 3>(out);
    res[n++] = run2<2, 1>(out); res[n++] = run2<2, 2>(out); res[n++] = run2<2, 3>(out);
    hipFree(out);
    return n;
}

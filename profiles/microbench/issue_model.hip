// What does one wave per SIMD overlap with its own MFMAs?  (scratch experiment behind the round-4 D4 kernel)
// For each MFMA shape, N independent MFMAs in a loop with F filler instructions after each one; cycles per MFMA by s_memtime.
//   filler kinds: 0 none, 1 v_add_f32 (independent registers), 2 v_exp_f32, 3 ds_read_b128 (conflict-free, waited once per iteration),
//                 4 v_xor + ds_read_b128 pairs alternating with 2 v_add (the mix of the real kernel)
#include <hip/hip_runtime.h>
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
#define SB __builtin_amdgcn_sched_barrier(0)

template <int SHAPE, int KIND, int F>
__global__ __launch_bounds__(512, 1) void k(unsigned long long *out, int iters, float seed) {
    __shared__ __attribute__((aligned(256))) char lds[65536];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) reinterpret_cast<float *>(lds)[i] = seed * i;
    __syncthreads();
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(seed + e + lane); b[e] = (__bf16)(seed * 2 + e - lane); }
    f32x4 c16[16];
    f32x16 c32[4];
    for (int i = 0; i < 16; ++i) c16[i] = f32x4{0, 0, 0, 0};
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) c32[i][e] = 0;
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = seed + i;
    u32x4 r[4] = {};
    const unsigned la = (unsigned)(unsigned long long)(__attribute__((address_space(3))) void *)lds + lane * 16;
    unsigned xa = la;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            if (SHAPE == 16) c16[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c16[m], 0, 0, 0);
            else if ((m & 1) == 0) c32[m >> 2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c32[m >> 2], 0, 0, 0);   // 8 per 16 slots: same FLOPs
            SB;
            const int nf = SHAPE == 16 ? F : ((m & 1) ? 2 * F : 0);      // the 32x32 form gets both slots' fillers in its one gap
#pragma unroll
            for (int f = 0; f < nf; ++f) {
                const int j = (m * 3 + f) & 7;
                if (KIND == 1) { asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[j]) : "v"(seed)); }
                else if (KIND == 2) { asm volatile("v_exp_f32 %0, %0" : "+v"(v[j])); }
                else if (KIND == 3) { asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(r[f & 3]) : "v"(la), "n"(1024) : "memory"); }
                else if (KIND == 4) {
                    if ((f & 1) == 0) { asm volatile("v_xor_b32 %0, %1, %2" : "=v"(xa) : "v"(la), "v"(64u * (m & 3))); asm volatile("ds_read_b128 %0, %1" : "=&v"(r[(f >> 1) & 3]) : "v"(xa) : "memory"); }
                    else { asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[j]) : "v"(seed)); asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[(j + 1) & 7]) : "v"(seed)); }
                }
            }
            SB;
        }
        if (KIND >= 3) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]) : : "memory");
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 16; ++i) s += c16[i][0];
    for (int i = 0; i < 4; ++i) s += c32[i][0];
    for (int i = 0; i < 8; ++i) s += v[i];
    s += __builtin_bit_cast(float, r[0][0] ^ r[1][1] ^ r[2][2] ^ r[3][3]);
    if (lane == 0) out[(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 2] = t1 - t0;
    if (s == 12345.678f) out[1] = 1;
}

template <int SHAPE, int KIND, int F>
static double run1(unsigned long long *out, int threads) {
    const int iters = 2000;
    hipLaunchKernelGGL((k<SHAPE, KIND, F>), dim3(256), dim3(threads), 0, 0, out, iters, 1.0f);
    hipDeviceSynchronize();
    hipLaunchKernelGGL((k<SHAPE, KIND, F>), dim3(256), dim3(threads), 0, 0, out, iters, 1.0f);
    hipDeviceSynchronize();
    unsigned long long h[8];
    hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    return (double)h[0] / (iters * 16.0);       // cycles per 16x16x32-equivalent MFMA slot (a 32x32x16 covers two slots)
}
extern "C" __attribute__((visibility("default"))) int issue_model(double *res) {
    unsigned long long *out;
    hipMalloc(&out, 1 << 16);
    int n = 0;
#define R(S, K, F) res[n++] = run1<S, K, F>(out, 256); res[n++] = run1<S, K, F>(out, 512);
    R(16, 0, 0) R(32, 0, 0)
    R(16, 1, 1) R(16, 1, 2) R(16, 1, 3) R(16, 1, 4) R(32, 1, 1) R(32, 1, 2) R(32, 1, 3) R(32, 1, 4)
    R(16, 2, 1) R(16, 2, 2) R(32, 2, 1) R(32, 2, 2)
    R(16, 3, 1) R(16, 3, 2) R(32, 3, 1) R(32, 3, 2)
    R(16, 4, 1) R(16, 4, 2) R(16, 4, 3) R(32, 4, 1) R(32, 4, 2) R(32, 4, 3)
    hipFree(out);
    return n;
}

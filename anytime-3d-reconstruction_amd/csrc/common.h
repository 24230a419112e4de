// Shared device/host helpers for libvoxvae (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/voxvae.h"

#define VV_EXPORT extern "C" __attribute__((visibility("default")))

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
typedef __attribute__((address_space(3))) void *lptr_t;     // LDS pointer (its integer value is the LDS byte address)

// One LDS-DMA instruction (buffer_load_dwordx4 ... lds: 16 B per lane straight into LDS at m0 + lane*16), issued from
// inline asm ON PURPOSE: hipcc counts its own LDS-DMA builtins as pending LDS writes and puts `s_waitcnt vmcnt(0)` in front
// of the next ds_read, which serialises the prefetch of chunk k+1 with the MFMAs of chunk k.  The asm form is invisible
// to that pass; the kernels wait for it by hand (s_waitcnt vmcnt(N) right before the barrier that publishes the buffer).
// Out-of-range lanes (voff >= num_records) deposit zeros.  m0 is written in the same statement (hipcc does not preserve
// it); `s_nop 4` covers the SGPR-write -> VMEM-read hazard on m0 / freshly produced descriptor words.
__device__ __forceinline__ void vv_dma16(u32x4 rsrc, unsigned voff, unsigned soff, unsigned lds_addr) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 4\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                 :
                 : "s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff)
                 : "memory");
}
__device__ __forceinline__ void vv_dma16(u32x4 rsrc, unsigned voff, unsigned lds_addr) { vv_dma16(rsrc, voff, 0u, lds_addr); }

// Raw buffer descriptor over [base, base + bytes): the range check is what turns padded taps / tails into zeros.
__device__ __forceinline__ u32x4 vv_make_rsrc(const void *base, unsigned bytes) {
    const unsigned long long b = reinterpret_cast<unsigned long long>(base);
    u32x4 r;
    r[0] = __builtin_amdgcn_readfirstlane((unsigned)b);
    r[1] = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32) & 0xFFFFu);   // stride 0: raw buffer
    r[2] = __builtin_amdgcn_readfirstlane(bytes);
    r[3] = 0x00020000u;
    return r;
}

static inline bool vv_is_pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }
static inline int vv_log2(int v) {
    int l = 0;
    while ((1 << l) < v) ++l;
    return l;
}
// Buffer descriptors carry 32-bit offsets: a launch covers at most 2 GiB of its largest per-sample-indexed tensor; larger batches
// go out as several launches over sample ranges (samples are independent).  VV_CHUNK_SAMPLES caps the range (tests).
#include <stdlib.h>
// Test hooks (kernel-form overrides for A/B tests and microbenchmarks) exist only in the build with -DVV_TEST_HOOKS
// (lib/libvoxvae_hooks.so, loaded by tests that set one of the VV_* variables); the release library reads no environment
// variable at all: vv_hook("...") is a null pointer there and the name does not reach the binary.
#ifdef VV_TEST_HOOKS
static inline const char *vv_hook(const char *name) { return getenv(name); }
#else
#define vv_hook(name) (static_cast<const char *>(nullptr))
#endif
static inline int vv_chunk_samples(size_t sample_bytes, int batch) {
    size_t per = sample_bytes ? 0x7FFFFFFFull / sample_bytes : (size_t)batch;
    if (const char *e = vv_hook("VV_CHUNK_SAMPLES")) {
        const long v = atol(e);
        if (v > 0 && (size_t)v < per) per = (size_t)v;
    }
    if (per > (size_t)batch) per = (size_t)batch;
    return (int)per;
}
static inline bool vv_aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
static inline size_t vv_dtype_size(int dt) { return dt == VV_BF16 ? 2 : (dt == VV_FP8 ? 1 : 4); }
// hipGetLastError() is per-thread and may hold a stale error from the host framework: clear it, then launch.
#define VV_LAUNCH(...)            \
    do {                          \
        (void)hipGetLastError();  \
        hipLaunchKernelGGL(__VA_ARGS__); \
    } while (0)

extern thread_local int vv_tls_last_hip_error;  // small.hip; read through vv_last_hip_error()
static inline int vv_launch_status() {
    const hipError_t e = hipGetLastError();
    if (e == hipSuccess) return VV_OK;
    vv_tls_last_hip_error = (int)e;
    return VV_ERR_LAUNCH;
}

// autoencoder3D.py:33-38: ELU(alpha 1) / ReLU / LeakyReLU(alpha 0.3, the Keras default)
__device__ __forceinline__ float vv_apply_act(float v, int act) {
    switch (act) {
        case VV_ACT_ELU: return v > 0.f ? v : expm1f(v);
        case VV_ACT_RELU: return v > 0.f ? v : 0.f;
        case VV_ACT_LRELU: return v > 0.f ? v : 0.3f * v;
        default: return v;
    }
}

// Same, for values about to be rounded to bf16 (8 significant bits): exp through v_exp_f32 instead of expm1f.
__device__ __forceinline__ float vv_apply_act_fast(float v, int act) {
    switch (act) {
        case VV_ACT_ELU: return v > 0.f ? v : __expf(v) - 1.f;
        case VV_ACT_RELU: return v > 0.f ? v : 0.f;
        case VV_ACT_LRELU: return v > 0.f ? v : 0.3f * v;
        default: return v;
    }
}

// Folded BN + activation of one accumulator quad (4 consecutive channels) for the bf16 epilogues: v * sc + sh, then the
// activation.  ELU is written as max(t, 0) + (exp2(min(t, 0) * log2 e) - 1): the same value as `t > 0 ? t : __expf(t) - 1`
// bit for bit (for t > 0 the second term is exp2(0) - 1 = 0 exactly), but every step except min / max / exp pairs up into
// v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32: 22 VALU instructions per quad instead of 26 (no compare + select).
typedef __attribute__((ext_vector_type(2))) float f32x2;
template <int ACT>
__device__ __forceinline__ f32x4 vv_bn_act4(f32x4 v, f32x4 sc, f32x4 sh) {
    f32x2 t0 = __builtin_elementwise_fma(f32x2{v[0], v[1]}, f32x2{sc[0], sc[1]}, f32x2{sh[0], sh[1]});
    f32x2 t1 = __builtin_elementwise_fma(f32x2{v[2], v[3]}, f32x2{sc[2], sc[3]}, f32x2{sh[2], sh[3]});
    if (ACT == VV_ACT_ELU) {
        const float L2E = 1.4426950408889634f;
        const f32x2 m0 = f32x2{fminf(t0[0], 0.f), fminf(t0[1], 0.f)} * L2E, m1 = f32x2{fminf(t1[0], 0.f), fminf(t1[1], 0.f)} * L2E;
        const f32x2 e0 = f32x2{__builtin_amdgcn_exp2f(m0[0]), __builtin_amdgcn_exp2f(m0[1])} - 1.f;
        const f32x2 e1 = f32x2{__builtin_amdgcn_exp2f(m1[0]), __builtin_amdgcn_exp2f(m1[1])} - 1.f;
        t0 = f32x2{fmaxf(t0[0], 0.f), fmaxf(t0[1], 0.f)} + e0;
        t1 = f32x2{fmaxf(t1[0], 0.f), fmaxf(t1[1], 0.f)} + e1;
    } else if (ACT == VV_ACT_RELU) {
        t0 = f32x2{fmaxf(t0[0], 0.f), fmaxf(t0[1], 0.f)};
        t1 = f32x2{fmaxf(t1[0], 0.f), fmaxf(t1[1], 0.f)};
    } else if (ACT == VV_ACT_LRELU) {
        t0 = f32x2{t0[0] > 0.f ? t0[0] : 0.3f * t0[0], t0[1] > 0.f ? t0[1] : 0.3f * t0[1]};
        t1 = f32x2{t1[0] > 0.f ? t1[0] : 0.3f * t1[0], t1[1] > 0.f ? t1[1] : 0.3f * t1[1]};
    }
    return f32x4{t0[0], t0[1], t1[0], t1[1]};
}

// OCP e4m3fn storage (gfx950's fp8; not MI300's fnuz).  Conversions saturate at +-448 by hand: the hardware convert maps
// larger magnitudes to NaN.
struct vv_fp8 { unsigned char v; };
__device__ __forceinline__ unsigned vv_pack_fp8x4(f32x4 v) {
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = fminf(fmaxf(v[e], -448.f), 448.f);
    int w = __builtin_amdgcn_cvt_pk_fp8_f32(v[0], v[1], 0, false);
    w = __builtin_amdgcn_cvt_pk_fp8_f32(v[2], v[3], w, true);
    return (unsigned)w;
}
__device__ __forceinline__ unsigned char vv_to_fp8(float v) {
    v = fminf(fmaxf(v, -448.f), 448.f);
    return (unsigned char)(__builtin_amdgcn_cvt_pk_fp8_f32(v, v, 0, false) & 0xFF);
}
__device__ __forceinline__ float vv_from_fp8(unsigned char b) { return __builtin_amdgcn_cvt_f32_fp8((int)b, 0); }

__device__ __forceinline__ float vv_load_f32(const float *p, size_t i) { return p[i]; }
__device__ __forceinline__ float vv_load_f32(const __bf16 *p, size_t i) { return static_cast<float>(p[i]); }
__device__ __forceinline__ void vv_store(float *p, size_t i, float v) { p[i] = v; }
__device__ __forceinline__ void vv_store(__bf16 *p, size_t i, float v) { p[i] = static_cast<__bf16>(v); }
__device__ __forceinline__ void vv_store(vv_fp8 *p, size_t i, float v) { p[i].v = vv_to_fp8(v); }
__device__ __forceinline__ float vv_load_f32(const vv_fp8 *p, size_t i) { return vv_from_fp8(p[i].v); }

__device__ __forceinline__ float vv_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// wgrad_phase.hip (internal): phase-form weight gradient of the stride-2 layers; the caller sums *splits slabs.
bool vv_wgrad_phase_ok(const void *src, const void *g, int batch, int side, int cin, int cout);
size_t vv_wgrad_phase_ws(long rows, int cin, int cout);
void vv_wgrad_phase_launch(const void *src, const void *g, float *slabs, int batch, int side, int cin, int cout, int *splits,
                           hipStream_t st);

// final_bce_fp8.hip (internal): sweep-form last layer with an e4m3fn input; returns the partial blocks per sample.
int vv_final_bce_sweep_fp8_launch(const void *x, const float *w_keras, const float *target, float *probs, float *logits, float *partials,
                                  int batch, int side, float gamma, float epsilon, hipStream_t st);

// posgemm.hip (internal): the 4^3 -> 2^3 convolution written as float32 split-K slabs only, in the MFMA FRAGMENT order of the
// producing workgroup (no transpose on the producer's side).  Piece of (position p, share s, sample tile mt, channel tile nt) =
// ws + (((first[p] + s) * mtiles + mt) * ntn + nt) * 256 * 128 floats; inside a piece the f32x4 holding channels c .. c + 3 (c % 4 == 0,
// local to the 128-channel tile) of local row r sits at index vv_pg_frag_index(r, c).
struct VvPgSlabPlan { int npos, mtiles, nitems, rows_per_tile, ntn; unsigned char nsplit[8]; unsigned short first[8]; };
size_t vv_pg_conv_slab_bytes(int batch, int cin, int cout);
int vv_pg_conv_slabs(const void *x, const void *w, int batch, int cin, int cout, void *ws, size_t ws_bytes, hipStream_t st, VvPgSlabPlan *plan);
// wave = (r >> 6) * 2 + (c >> 6); nt_ = (c >> 5) & 1; mt_ = (r >> 5) & 1; g = (c >> 3) & 3; lane = ((c >> 2) & 1) * 32 + (r & 31)
__host__ __device__ inline int vv_pg_frag_index(int r, int c) {
    return ((((((r >> 6) * 2 + (c >> 6)) * 2 + ((c >> 5) & 1)) * 2 + ((r >> 5) & 1)) * 4 + ((c >> 3) & 3)) * 64) + ((c >> 2) & 1) * 32 + (r & 31);
}

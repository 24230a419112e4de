// The HBM-bound tail of the path (gfx950):
//   final_bce  : Conv3DTranspose k4 s2 SAME -> 1 channel, sigmoid, weighted BCE and TP/FP/FN, fused
//                (autoencoder3D.py:129-136; function.py:73-82, 100-115)
#include <stdlib.h>
#include <string.h>

#include <type_traits>

#include "common.h"

namespace {

// ---------------------------------------------------------------------------------------------------------------
// final_bce.  One workgroup = a 4x4x4 block of input-grid cells (-> 8x8x8 logits) of one sample.  The 6x6x6 input
// halo tile is staged in LDS as float32 rows (padded by 16 B against bank conflicts); wave w owns output parity
// (pd,ph) = (w>>1, w&1) and each lane both pw parities of its cell, so the 16 weight vectors a wave needs are
// wave-uniform and come through the scalar cache.  Loss terms are reduced by wave shuffles, then across the 4
// waves in LDS, and written as one partial per workgroup; final_reduce sums a sample's partials in block order.
constexpr int FB_CIN = 64;
constexpr int FB_ROW = FB_CIN + 4;  // floats per staged voxel row (+16 B pad)

template <typename T>
__global__ __launch_bounds__(256) void final_bce_kernel(const T *__restrict__ x, const float *__restrict__ w,
                                                        const float *__restrict__ target, float *__restrict__ probs,
                                                        float *__restrict__ logits, float *__restrict__ partials,
                                                        int din_log2, float gamma, float epsilon) {
    __shared__ __attribute__((aligned(16))) float tile[216 * FB_ROW];
    __shared__ float red[4][4];
    const int li = din_log2, n = 1 << li, nb = n >> 2;  // blocks per axis
    const int blk = blockIdx.x, b = blockIdx.y;
    const int bw = blk % nb, bh = (blk / nb) % nb, bd = blk / (nb * nb);
    const int m0d = bd * 4, m0h = bh * 4, m0w = bw * 4;
    const T *xb = x + ((size_t)b << (3 * li)) * FB_CIN;

    // stage the halo tile: 216 voxels x 64 channels, 16 B (= 4 f32 / 8 bf16 -> split) per lane per step
    constexpr int EPL = 16 / sizeof(T);          // elements per 16-byte load
    constexpr int LPV = FB_CIN / EPL;            // loads per voxel
    for (int i = threadIdx.x; i < 216 * LPV; i += 256) {
        const int vox = i / LPV, part = i % LPV;
        const int zw = vox % 6, zh = (vox / 6) % 6, zd = vox / 36;
        const int id = m0d - 1 + zd, ih = m0h - 1 + zh, iw = m0w - 1 + zw;
        float vals[EPL];
        if ((unsigned)id < (unsigned)n && (unsigned)ih < (unsigned)n && (unsigned)iw < (unsigned)n) {
            const T *src = xb + ((((size_t)id << li) + ih << li) + iw) * FB_CIN + part * EPL;
            const uint4 raw = *reinterpret_cast<const uint4 *>(src);
            const T *rv = reinterpret_cast<const T *>(&raw);
#pragma unroll
            for (int e = 0; e < EPL; ++e) vals[e] = static_cast<float>(rv[e]);
        } else {
#pragma unroll
            for (int e = 0; e < EPL; ++e) vals[e] = 0.f;
        }
        float *dst = tile + vox * FB_ROW + part * EPL;
#pragma unroll
        for (int e = 0; e < EPL; e += 4) *reinterpret_cast<f32x4 *>(dst + e) = f32x4{vals[e], vals[e + 1], vals[e + 2], vals[e + 3]};
    }
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int pd = wv >> 1, ph = wv & 1;
    const int mw = lane & 3, mh = (lane >> 2) & 3, md = lane >> 4;
    float acc0 = 0.f, acc1 = 0.f;
#pragma unroll
    for (int ad = 0; ad < 2; ++ad) {
#pragma unroll
        for (int ah = 0; ah < 2; ++ah) {
            const int zd = md + pd - ad + 1, zh = mh + ph - ah + 1;
            const int td = 1 - pd + 2 * ad, th = 1 - ph + 2 * ah;
            const float *r0 = tile + ((zd * 6 + zh) * 6 + mw) * FB_ROW;  // zw = mw, mw+1, mw+2
            const float *wt = w + (size_t)((td * 4 + th) * 4) * FB_CIN;  // [tw][ci], wave-uniform
#pragma unroll 4
            for (int c = 0; c < FB_CIN; c += 4) {
                const f32x4 x0 = *reinterpret_cast<const f32x4 *>(r0 + c);
                const f32x4 x1 = *reinterpret_cast<const f32x4 *>(r0 + FB_ROW + c);
                const f32x4 x2 = *reinterpret_cast<const f32x4 *>(r0 + 2 * FB_ROW + c);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    // pw = 0: i = mw (tw 1), mw-1 (tw 3);  pw = 1: i = mw+1 (tw 0), mw (tw 2)
                    acc0 = fmaf(x1[e], wt[1 * FB_CIN + c + e], acc0);
                    acc0 = fmaf(x0[e], wt[3 * FB_CIN + c + e], acc0);
                    acc1 = fmaf(x2[e], wt[0 * FB_CIN + c + e], acc1);
                    acc1 = fmaf(x1[e], wt[2 * FB_CIN + c + e], acc1);
                }
            }
        }
    }
    const int lo = li + 1;
    const int od = 2 * (m0d + md) + pd, oh = 2 * (m0h + mh) + ph, ow = 2 * (m0w + mw);
    const size_t o = ((((size_t)b << lo) + od << lo) + oh << lo) + ow;
    const float2 y = *reinterpret_cast<const float2 *>(target + o);
    const float l[2] = {acc0, acc1}, yy[2] = {y.x, y.y};
    float p[2], bce = 0.f, tp = 0.f, fp = 0.f, fn = 0.f;
    const float hi = 1.0f - epsilon;  // 0.99999988 in float32 (function.py:79)
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        p[e] = 1.0f / (1.0f + expf(-l[e]));                         // tf.sigmoid, autoencoder3D.py:136
        const float q = fminf(fmaxf(p[e], epsilon), hi);
        bce -= gamma * yy[e] * logf(q) + (1.0f - gamma) * (1.0f - yy[e]) * logf(1.0f - q);   // function.py:80
        const float yh = p[e] >= 0.5f ? 1.f : 0.f;                  // function.py:110
        tp += yy[e] * yh; fp += (1.f - yy[e]) * yh; fn += yy[e] * (1.f - yh);
    }
    if (probs) *reinterpret_cast<float2 *>(probs + o) = make_float2(p[0], p[1]);
    if (logits) *reinterpret_cast<float2 *>(logits + o) = make_float2(l[0], l[1]);
    bce = vv_wave_sum(bce); tp = vv_wave_sum(tp); fp = vv_wave_sum(fp); fn = vv_wave_sum(fn);
    if (lane == 0) { red[wv][0] = bce; red[wv][1] = tp; red[wv][2] = fp; red[wv][3] = fn; }
    __syncthreads();
    if (threadIdx.x < 4) {
        const float s = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
        partials[((size_t)b * gridDim.x + blk) * 4 + threadIdx.x] = s;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// final_bce on MFMA (bf16 activations), scatter form.  A transposed conv with one output channel is
//   P[i][t] = sum_ci x[i][ci] * w[t][ci]      (a [voxels x 64] x [64 x 64 taps] GEMM: v_mfma_f32_32x32x16_bf16)
//   logit[o] = sum_{(i,t) : o = 2i + t - 1} P[i][t]   (8 terms per output voxel)
// One workgroup = 4x4x4 input cells (+1 halo: 216 rows, padded to 224) -> P in LDS (f32, aliased over the
// staged operands) -> every lane gathers its 2 x 8 terms, then sigmoid / BCE / TP / FP / FN as in the VALU kernel.
constexpr int FM_ROWS = 224;            // 216 halo voxels padded to 7 MFMA row tiles
constexpr int FM_PPITCH = 33;           // floats per P row (32 taps of one half + 1: consecutive voxels on consecutive banks)

__device__ __forceinline__ int fm_lds_off(int row, int slot) { return row * 128 + ((slot ^ ((row >> 1) & 7)) << 4); }

__global__ __launch_bounds__(256) void final_bce_mfma_kernel(const __bf16 *__restrict__ x, const float *__restrict__ w,
                                                             const float *__restrict__ target, float *__restrict__ probs,
                                                             float *__restrict__ logits, float *__restrict__ partials,
                                                             int din_log2, unsigned x_bytes, float gamma, float epsilon) {
    // LDS: staged operands (36 KiB), later overwritten by ONE 32-tap half of P at a time (28 KiB): 4 workgroups per CU.
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *As = smem;                                   // [224][128 B] bf16 rows, slot-swizzled (source side)
    char *Ws = smem + FM_ROWS * 128;                   // [64 taps][128 B]
    float *P = reinterpret_cast<float *>(smem);        // [216][33] f32
    __shared__ float red[4][4];
    const int li = din_log2, n = 1 << li, nb = n >> 2, nblk = nb * nb * nb;
    // XCD-aware order (workgroups are dealt round-robin over 8 XCDs): block g takes item (g % 8) * (T / 8) + g / 8 of the
    // sample-major list, so the 6^3 halo tiles of neighbouring blocks of one sample are re-read from ONE XCD's L2.
    const int T = gridDim.x;
    const int wi = (T & 7) ? (int)blockIdx.x : (int)(blockIdx.x & 7) * (T >> 3) + (int)(blockIdx.x >> 3);
    const int blk = wi % nblk, b = wi / nblk;
    const int bw = blk % nb, bh = (blk / nb) % nb, bd = blk / (nb * nb);
    const int m0d = bd * 4, m0h = bh * 4, m0w = bw * 4;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);

    // stage A by LDS-DMA: 224 rows x 8 slots = 28 wave instructions (8 rows each); rows >= 216 and halo voxels outside
    // the grid come back as zeros (out-of-range buffer offsets)
    {
        const u32x4 rs = vv_make_rsrc(x, x_bytes);
        const unsigned lds0 = (unsigned)(unsigned long long)(lptr_t)As;
        const int pos = lane & 7, rsub = lane >> 3;
        for (int it = wv; it < FM_ROWS / 8; it += 4) {
            const int row = it * 8 + rsub;
            const int zw = row % 6, zh = (row / 6) % 6, zd = row / 36;
            const int id = m0d - 1 + zd, ih = m0h - 1 + zh, iw = m0w - 1 + zw;
            const bool ok = row < 216 && (unsigned)id < (unsigned)n && (unsigned)ih < (unsigned)n && (unsigned)iw < (unsigned)n;
            const int g = pos ^ ((row >> 1) & 7);
            const unsigned vo = ok ? (unsigned)((((((b << li) + id) << li) + ih) << li) + iw) * (FB_CIN * 2) + g * 16 : 0xFFFFFFF0u;
            vv_dma16(rs, vo, lds0 + it * 1024);
        }
    }
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int idx = tid + 256 * it, row = idx >> 3, slot = idx & 7;
        const f32x4 w0 = *reinterpret_cast<const f32x4 *>(w + row * FB_CIN + slot * 8);
        const f32x4 w1 = *reinterpret_cast<const f32x4 *>(w + row * FB_CIN + slot * 8 + 4);
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) { o[e] = static_cast<__bf16>(w0[e]); o[4 + e] = static_cast<__bf16>(w1[e]); }
        *reinterpret_cast<bf16x8 *>(Ws + fm_lds_off(row, slot)) = o;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // MFMA: wave -> tap half nt = wv & 1, row tiles mt = (wv >> 1) + 2 j
    const int fr = lane & 31, fh = lane >> 5;
    const int nt = wv & 1, mt0 = wv >> 1;
    uint4 fb[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) fb[ks] = *reinterpret_cast<const uint4 *>(Ws + fm_lds_off(nt * 32 + fr, ks * 2 + fh));
    f32x16 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[j][q] = 0.f;
        const int mt = mt0 + 2 * j;
        if (mt < 7) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const uint4 fa = *reinterpret_cast<const uint4 *>(As + fm_lds_off(mt * 32 + fr, ks * 2 + fh));
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8 *>(&fa),
                                                                 *reinterpret_cast<const bf16x8 *>(&fb[ks]), acc[j], 0, 0, 0);
            }
        }
    }

    // wave -> output parity (pd, ph); lane -> cell; both pw parities per lane.  Tap half h holds td = 2h, 2h+1, i.e. the
    // terms with ad = h of every output: two passes of {waves of that half publish P, everyone gathers its 4 terms}.
    const int pd = wv >> 1, ph = wv & 1;
    const int mw = lane & 3, mh = (lane >> 2) & 3, md = lane >> 4;
    float acc0 = 0.f, acc1 = 0.f;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        __syncthreads();   // operands (h = 0) / previous half (h = 1) no longer read by anyone
        if (nt == h) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int mt = mt0 + 2 * j;
                if (mt < 7) {
#pragma unroll
                    for (int q = 0; q < 16; ++q) {
                        const int row = mt * 32 + (q & 3) + 8 * (q >> 2) + 4 * fh;
                        if (row < 216) P[row * FM_PPITCH + fr] = acc[j][q];
                    }
                }
            }
        }
        __syncthreads();
        const int ad = h, td = 1 - pd + 2 * ad - 2 * h;          // tap row inside this half (0 or 1)
#pragma unroll
        for (int ah = 0; ah < 2; ++ah) {
            const int zd = md + pd - ad + 1, zh = mh + ph - ah + 1;
            const int th = 1 - ph + 2 * ah;
            const float *r = P + ((zd * 6 + zh) * 6 + mw) * FM_PPITCH + (td * 4 + th) * 4;
            acc0 += r[FM_PPITCH + 1] + r[3];                    // pw = 0: i = mw (tw 1), mw-1 (tw 3)
            acc1 += r[2 * FM_PPITCH + 0] + r[FM_PPITCH + 2];    // pw = 1: i = mw+1 (tw 0), mw (tw 2)
        }
    }
    const int lo = li + 1;
    const int od = 2 * (m0d + md) + pd, oh = 2 * (m0h + mh) + ph, ow = 2 * (m0w + mw);
    const size_t o = (((((size_t)b << lo) + od) << lo) + oh << lo) + ow;
    const float2 y = *reinterpret_cast<const float2 *>(target + o);
    const float l[2] = {acc0, acc1}, yy[2] = {y.x, y.y};
    float p[2], bce = 0.f, tp = 0.f, fp = 0.f, fn = 0.f;
    const float hi = 1.0f - epsilon;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        p[e] = 1.0f / (1.0f + expf(-l[e]));
        const float q = fminf(fmaxf(p[e], epsilon), hi);
        bce -= gamma * yy[e] * logf(q) + (1.0f - gamma) * (1.0f - yy[e]) * logf(1.0f - q);
        const float yh = p[e] >= 0.5f ? 1.f : 0.f;
        tp += yy[e] * yh; fp += (1.f - yy[e]) * yh; fn += yy[e] * (1.f - yh);
    }
    if (probs) *reinterpret_cast<float2 *>(probs + o) = make_float2(p[0], p[1]);
    if (logits) *reinterpret_cast<float2 *>(logits + o) = make_float2(l[0], l[1]);
    bce = vv_wave_sum(bce); tp = vv_wave_sum(tp); fp = vv_wave_sum(fp); fn = vv_wave_sum(fn);
    if (lane == 0) { red[wv][0] = bce; red[wv][1] = tp; red[wv][2] = fp; red[wv][3] = fn; }
    __syncthreads();
    if (tid < 4) partials[((size_t)b * nblk + blk) * 4 + tid] = red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
}

// ---------------------------------------------------------------------------------------------------------------
// final_bce, sweep form (bf16): the box kernel above stages a 6^3 halo for 4^3 cells, so it loads, multiplies and
// publishes every input voxel 3.4 times.  Here one workgroup owns an 8 x 8 tile of cells in (h, w) and sweeps the whole
// depth: per plane d it stages the 10 x 10 halo rows ONCE (LDS-DMA, double buffered), forms P_d = X_d W^T on MFMA, keeps
// the td in {2,3} half of P_d for the next step and combines the td in {0,1} half with the kept half of P_{d-1}:
//   od = 2d - 1 + s  <-  P_d[td = s] + P_{d-1}[td = 2 + s]            (s = 0, 1; 2 x 2 terms in h, w each)
// so a step finishes two output planes of 16 x 16 voxels (256 threads x one pw pair).  Amplification 1.56 (h, w halo
// only), P is published once per cell, and the four BCE / TP / FP / FN sums stay in registers for the whole sweep.
// The voxel math uses the hardware exp / log / rcp (relative error ~1e-7, far below the bf16 operand rounding) and
// thresholds on the logit (sigmoid(l) >= 0.5 <=> l >= 0, function.py:110).
// P rows are dense (32 taps = 8 quads of 16 B); quad q of the row of halo cell (zh, zw) sits at slot q ^ (zw & 7): the gather reads
// whole quads with ds_read_b128 and this slot key makes every one of its lane groups conflict-free (exhaustive search over
// a*zh + b*zw keys and pitches 32 / 36 / 40: profiles/microbench/d5_swz.py; the dword gathers of rounds 1-2 at pitch 36 were 4-way).
#ifndef VV_SW_DEPTH
#define VV_SW_DEPTH 1            // input planes in flight ahead of the one being multiplied (ring = depth + 1 slots)
#endif
constexpr int SW_DEPTH = VV_SW_DEPTH;
constexpr int SW_ROWS = 100, SW_XB = 13 * 1024, SW_NX = SW_DEPTH + 1, SW_PP = 32, SW_PSZ = 100 * SW_PP;   // X slot bytes (104 rows); P row pitch / buffer floats
// LDS: 2 plane slots + 1 KiB sink + PL + PH = 53,248 B (+ 64 B of static sums): THREE workgroups per CU (rounds 1-2: 73.6 KB, two).
// The 4th MFMA row tile reads 24 rows past a plane slot (into the next slot / the sink and the head of PL): whatever it finds only
// reaches accumulator rows >= 104, which are never published.
constexpr int SW_LDS = SW_NX * SW_XB + 1024 + 2 * SW_PSZ * 4;

__global__ __launch_bounds__(256, SW_DEPTH == 1 ? 3 : 2) void final_bce_sweep_kernel(const __bf16 *__restrict__ x, const float *__restrict__ w,
                                                                 const float *__restrict__ target, float *__restrict__ probs,
                                                                 float *__restrict__ logits, float *__restrict__ partials,
                                                                 int din_log2, unsigned x_bytes, float gamma, float epsilon) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *Xs = smem;                                             // ring of SW_NX planes x [104 rows][128 B], slot-swizzled; 1 KiB sink
    float *PL = reinterpret_cast<float *>(smem + SW_NX * SW_XB + 1024);   // P_d[td 0,1]  [100][32]
    float *PH = PL + SW_PSZ;                                     // P_d[td 2,3]  [100][32]: read in step d for the outputs of step d+1
    __shared__ float red[4][4];
    const int li = din_log2, n = 1 << li, nt8 = n >> 3, ntile = nt8 * nt8;
    const int T = gridDim.x;
    const int wi = (T & 7) ? (int)blockIdx.x : (int)(blockIdx.x & 7) * (T >> 3) + (int)(blockIdx.x >> 3);
    const int tile = wi % ntile, b = wi / ntile;
    const int h0 = (tile / nt8) * 8, w0 = (tile % nt8) * 8;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 31, fh = lane >> 5;

    const u32x4 rs = vv_make_rsrc(x, x_bytes);
    const unsigned ldsx = (unsigned)(unsigned long long)(lptr_t)Xs;
    // plane d -> ring slot d % 3: 13 pieces of 8 rows; every wave issues 4 (the 3 surplus ones go to the sink so that the
    // vector-memory counter advances uniformly); rows >= 100, voxels outside the grid and planes outside [0, n) arrive
    // as zeros (the virtual plane d = n closes the sweep).  The 4th MFMA row tile reads rows 96..127, i.e. 24 rows past
    // the slot: whatever it finds there only reaches accumulator rows >= 104, which are never published.
    // The lane part of a piece's source offset (sample, halo row, swizzled slot; out-of-range if the row is outside the grid
    // or past the 100 halo rows) is prepared once; the plane rides in soffset.
    unsigned sv[4], sdst[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int piece = wv * 4 + i, row = piece * 8 + (lane >> 3);
        const int zh = row / 10, zw = row - zh * 10;
        const int ih = h0 - 1 + zh, iw = w0 - 1 + zw;
        const bool ok = row < SW_ROWS && (unsigned)ih < (unsigned)n && (unsigned)iw < (unsigned)n;
        const int g = (lane & 7) ^ ((row >> 1) & 7);
        sv[i] = ok ? (unsigned)(((((b << li) << li) + ih) << li) + iw) * (FB_CIN * 2) + g * 16 : 0xFFFFFFF0u;
        sdst[i] = piece < 13 ? (unsigned)(piece * 1024) : (unsigned)(SW_NX * SW_XB);     // surplus pieces: the sink (ring-slot independent)
    }
    auto stage = [&](int d, int sp) {              // sp = d % SW_NX, passed so that the unrolled steps see a constant
        // a plane outside [0, n): every lane out of range by its OFFSET.  (A descriptor of zero records is not a substitute: the
        // zero-fill of the virtual plane d = n then went missing now and then and od = 2n - 1 read the stale slot -- found by the
        // B = 256 cross-check against the box form, profiles/microbench/chk_e1_d5.py.)
        const bool din = (unsigned)d < (unsigned)n;
        const unsigned soff = (unsigned)__builtin_amdgcn_readfirstlane(din ? (d << (2 * li)) * (FB_CIN * 2) : 0);
        const unsigned slot = ldsx + sp * SW_XB;
#pragma unroll
        for (int i = 0; i < 4; ++i) vv_dma16(rs, din ? sv[i] : 0xFFFFFFF0u, soff, wv * 4 + i < 13 ? slot + sdst[i] : ldsx + sdst[i]);
    };
    stage(0, 0);

    // weights of this wave's tap half as B fragments (lane: tap nt*32 + fr, k = ks*16 + 8 fh + j), straight from the
    // Keras array [64 taps][64 ci]
    const int nt = wv & 1, mt0 = wv >> 1;
    uint4 fb[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const float *wr = w + (nt * 32 + fr) * FB_CIN + ks * 16 + 8 * fh;
        const f32x4 w0v = *reinterpret_cast<const f32x4 *>(wr), w1v = *reinterpret_cast<const f32x4 *>(wr + 4);
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) { o[e] = static_cast<__bf16>(w0v[e]); o[4 + e] = static_cast<__bf16>(w1v[e]); }
        fb[ks] = *reinterpret_cast<const uint4 *>(&o);
    }

    // gather role: s = od parity slot, ohh = output row inside the tile, mw = cell column (both pw per lane)
    const int mw = tid & 7, ohh = (tid >> 3) & 15, sl = tid >> 7;
    const int mh = ohh >> 1, ph = ohh & 1;
    const int lo = li + 1, n2 = 2 * n;
    const int oh = 2 * h0 + ohh, ow = 2 * (w0 + mw);
    const float hi = 1.0f - epsilon;
    float bce = 0.f, tp = 0.f, fp = 0.f, fn = 0.f;
    float lo0 = 0.f, lo1 = 0.f;                                  // td in {2,3} contributions of P_{d-1} to this step's outputs (P_{-1} = 0)

    // P_d = X_d W^T for this wave's two row tiles and its tap half: D[tap][cell], weights-first
    auto mfma_plane = [&](int sp, f32x16 (&acc)[2]) {             // sp = ring slot of the plane
        const char *Xd = Xs + sp * SW_XB;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[j][q] = 0.f;
        // k-step outer, row tile inner: consecutive MFMAs go to different accumulators (the other order is two chains of four
        // dependent MFMAs)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int mt = mt0 + 2 * j;
                const uint4 fa = *reinterpret_cast<const uint4 *>(Xd + fm_lds_off(mt * 32 + fr, ks * 2 + fh));
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8 *>(&fb[ks]),
                                                                 *reinterpret_cast<const bf16x8 *>(&fa), acc[j], 0, 0, 0);
            }
        }
    };

    // Software pipeline: step d publishes P_d (computed during step d-1) and then runs the MFMAs of plane d+1 in the same
    // instruction stream as the gather / loss math of plane d (matrix pipe under the VALU and LDS work).
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // plane 0
    __syncthreads();                                             // ... for every wave
    f32x16 acc[2];
    mfma_plane(0, acc);
#pragma unroll
    for (int k = 1; k <= SW_DEPTH; ++k) stage(k, k);
    __syncthreads();                                             // slot 0 may be refilled from the first step on

    // (Unrolling this loop by two with the step parity as a compile-time constant -- ring slot and P buffer addresses folded
    // into the instructions -- is worth 2 % (53.5 vs 54.7 us) in the clean kernel; with the ablation switches still compiled in
    // it returned a low loss sum at B = 256 with exact logits and counts, which is not understood: not used.)
    int oldh = 0;                                                // ring slot of plane d
#pragma unroll 1
    for (int d = 0; d <= n; ++d) {
        // weights-first: lane = cell row, registers walk the taps of the half; quad g = taps 8g + 4fh .. +3 = the four tw
        // of one (td, th): one 16-byte store per quad
        float *Pw = nt == 0 ? PL : PH;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int row = (mt0 + 2 * j) * 32 + fr;
            if (row < SW_ROWS) {
                const int zwk = (row - (row / 10) * 10) & 7;              // slot key of this halo cell
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *reinterpret_cast<f32x4 *>(Pw + row * SW_PP + (((2 * g + fh) ^ zwk) << 2)) =
                        f32x4{acc[j][4 * g], acc[j][4 * g + 1], acc[j][4 * g + 2], acc[j][4 * g + 3]};
            }
        }
        const int od = 2 * d - 1 + sl;
        const bool ovalid = (unsigned)od < (unsigned)n2;
        const size_t o = ((((((size_t)b << lo) + (ovalid ? od : 0)) << lo) + oh) << lo) + ow;
        // The target pair is loaded by inline asm so that its wait can be counted: the vector-memory counter retires in
        // order, and a compiler-placed wait for this load would be vmcnt(0), i.e. it would also wait for the 4 pieces of
        // plane d+2 issued right after it -- the look-ahead.  In flight, oldest first:
        //   [plane d+1 x4][stores d-1] [y d][plane d+2 x4]
        // so "all but the newest 5" covers plane d+1 whatever the number of stores (more stores only wait for more).
        // An asm output is a READY value to the compiler: nothing in the language stops it from copying y or re-using its
        // registers while the load is in flight.  tests/test_isa_lint.py checks on the generated code that no instruction
        // names the pair between this load and the counted wait below that lands it ("+v"(y)); round 2's dead ends came from
        // exactly that (a look-ahead load whose last instance was DEAD: its registers went to the logit accumulators of the
        // last plane while it was in flight -- DESIGN.md section 4d).  The two compiler-managed alternatives were built in round 3
        // and are worse: a plain load of the noalias argument is moved by the compiler across the asm statements into the
        // `ovalid` branch (behind the counted wait, whose count then no longer holds: wrong logits), a volatile load becomes a
        // system-scope flat load with an immediate vmcnt(0).
        float2 y;
        asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(y) : "v"(target + o) : "memory");
        stage(d + 1 + SW_DEPTH, oldh);
        // depth 1: [plane d+1 x4][stores d-1][y d][plane d+2 x4] -> all but the newest 5.  depth 2: plane d+1 is followed by
        // stores d-2 (0..2), y d-1, plane d+2 x4, stores d-1 (0..2), y d, plane d+3 x4 = 10..14 operations -> all but the newest 10
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(SW_DEPTH == 1 ? 5 : 10) : "memory");         // plane d+1 has landed
        __syncthreads();                                         // ... for every wave; P_d is published

        f32x16 acc_next[2];
        const int nexth = oldh + 1 == SW_NX ? 0 : oldh + 1;      // ring slot of plane d+1
        mfma_plane(nexth, acc_next);

        // gather: per ah the tap quads (tw 0..3) of the three cells mw, mw+1, mw+2 -- ds_read_b128, conflict-free.  The td in {0,1}
        // half of P_d (PL) completes the output planes od = 2d - 1 + sl together with the td in {2,3} half of P_{d-1}, which was
        // gathered a step ago into (lo0, lo1): PH is read in the step that publishes it, so ONE buffer holds it.
        float l0 = lo0, l1 = lo1;
        lo0 = 0.f; lo1 = 0.f;
#pragma unroll
        for (int ah = 0; ah < 2; ++ah) {
            const int zh = mh + ph - ah + 1, th = 1 - ph + 2 * ah;
            const int q = sl * 4 + th, rowb = (zh * 10 + mw) * SW_PP;
            const int o0 = rowb + ((q ^ (mw & 7)) << 2), o1 = rowb + SW_PP + ((q ^ ((mw + 1) & 7)) << 2),
                      o2 = rowb + 2 * SW_PP + ((q ^ ((mw + 2) & 7)) << 2);
            const f32x4 a0 = *reinterpret_cast<const f32x4 *>(PL + o0), a1 = *reinterpret_cast<const f32x4 *>(PL + o1),
                        a2 = *reinterpret_cast<const f32x4 *>(PL + o2);
            const f32x4 b0 = *reinterpret_cast<const f32x4 *>(PH + o0), b1 = *reinterpret_cast<const f32x4 *>(PH + o1),
                        b2 = *reinterpret_cast<const f32x4 *>(PH + o2);
            l0 += a1[1] + a0[3];                                         // pw = 0: cell mw+1 (tw 1), cell mw (tw 3)
            l1 += a2[0] + a1[2];                                         // pw = 1: cell mw+2 (tw 0), cell mw+1 (tw 2)
            lo0 += b1[1] + b0[3];
            lo1 += b2[0] + b1[2];
        }
        asm volatile("s_waitcnt vmcnt(4)" : "+v"(y) : : "memory");   // y has landed; plane d+2 may still be in flight
        if (ovalid) {
            const float l[2] = {l0, l1}, yy[2] = {y.x, y.y};
            float p[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                p[e] = __builtin_amdgcn_rcpf(1.0f + __expf(-l[e]));
                const float q = fminf(fmaxf(p[e], epsilon), hi);
                bce -= gamma * yy[e] * __logf(q) + (1.0f - gamma) * (1.0f - yy[e]) * __logf(1.0f - q);
                const float yh = l[e] >= 0.f ? 1.f : 0.f;
                tp += yy[e] * yh; fp += (1.f - yy[e]) * yh; fn += yy[e] * (1.f - yh);
            }
            if (probs) *reinterpret_cast<float2 *>(probs + o) = make_float2(p[0], p[1]);
            if (logits) *reinterpret_cast<float2 *>(logits + o) = make_float2(l[0], l[1]);
        }
        acc[0] = acc_next[0];
        acc[1] = acc_next[1];
        oldh = nexth;
        __syncthreads();      // every gather of P_d / P_{d-1} and every read of plane d+1 is done: publish d+1, refill its slot
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // the last (all-zero) look-ahead planes
    bce = vv_wave_sum(bce); tp = vv_wave_sum(tp); fp = vv_wave_sum(fp); fn = vv_wave_sum(fn);
    if (lane == 0) { red[wv][0] = bce; red[wv][1] = tp; red[wv][2] = fp; red[wv][3] = fn; }
    __syncthreads();
    if (tid < 4) partials[((size_t)b * ntile + tile) * 4 + tid] = red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
}

// ---------------------------------------------------------------------------------------------------------------
// final_bce, sweep form with the w direction summed INSIDE the MFMA (bf16, round 3).  The sweep kernel above publishes P[halo cell][64 taps]
// (25.6 KB of float32 per plane) and every output gathers 8 terms from it; its ablations put the memory side of the
// layer at 31 us and the float32 round trip of P through LDS (written at the LDS write rate, gathered as whole tap quads) at most of the
// other 20.  The two w terms of an output column share their centre cell: out[2i] = x_i w[tw 1] + x_{i-1} w[tw 3], out[2i+1] = x_{i+1} w[tw 0]
// + x_i w[tw 2].  With K = 128 = (centre | left) resp. (right | centre) channels and the 16 (td, th) pairs as the MFMA's rows
// (v_mfma_f32_16x16x32_bf16), the matrix pipe delivers Q[centre cell][td][th][pw] -- the same FLOPs, float32 sums as before, but 80 cells x 32
// values = 10 KB per plane instead of 100 x 64, and an output pair reads ONE 8-byte granule per (ah, td) instead of three 16-byte quads:
// LDS written / 2.5, gathered / 6, 38 KB of LDS and <= 128 VGPRs = four workgroups per CU.  Staging, the counted waits and the voxel math
// are the sweep kernel's, unchanged.
constexpr int SWW_LDS = SW_NX * SW_XB + 1024 + 80 * 128;
__global__ __launch_bounds__(256, 4) void final_bce_sweepw_kernel(const __bf16 *__restrict__ x, const float *__restrict__ w,
                                                                 const float *__restrict__ target, float *__restrict__ probs,
                                                                 float *__restrict__ logits, float *__restrict__ partials,
                                                                 int din_log2, unsigned x_bytes, float gamma, float epsilon) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *Xs = smem;                                             // ring of SW_NX planes x [104 rows][128 B], slot-swizzled; 1 KiB sink
    char *Pq = smem + SW_NX * SW_XB + 1024;                      // Q_d [80 centre cells][td 4][th 4][pw 2] float32, 8-byte granule g at g ^ key(cell)
    __shared__ float red[4][4];
    const int li = din_log2, n = 1 << li, nt8 = n >> 3, ntile = nt8 * nt8;
    const int T = gridDim.x;
    const int wi = (T & 7) ? (int)blockIdx.x : (int)(blockIdx.x & 7) * (T >> 3) + (int)(blockIdx.x >> 3);
    const int tile = wi % ntile, b = wi / ntile;
    const int h0 = (tile / nt8) * 8, w0 = (tile % nt8) * 8;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);

    const u32x4 rs = vv_make_rsrc(x, x_bytes);
    const unsigned ldsx = (unsigned)(unsigned long long)(lptr_t)Xs;
    // plane d -> ring slot d % 2: 13 pieces of 8 rows; every wave issues 4 (the 3 surplus ones go to the sink so that the
    // vector-memory counter advances uniformly); rows >= 100, voxels outside the grid and planes outside [0, n) arrive
    // as zeros (the virtual plane d = n closes the sweep).  The operand reads stay inside rows 0 .. 99 (centre cells zw = 1 .. 8 and
    // their left / right neighbours).
    // The lane part of a piece's source offset (sample, halo row, swizzled slot; out-of-range if the row is outside the grid
    // or past the 100 halo rows) is prepared once; the plane rides in soffset.
    unsigned sv[4], sdst[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int piece = wv * 4 + i, row = piece * 8 + (lane >> 3);
        const int zh = row / 10, zw = row - zh * 10;
        const int ih = h0 - 1 + zh, iw = w0 - 1 + zw;
        const bool ok = row < SW_ROWS && (unsigned)ih < (unsigned)n && (unsigned)iw < (unsigned)n;
        const int g = (lane & 7) ^ (zw & 7);             // slot key zw & 7: conflict-free for the 16x16x32 operand reads of centre / left / right cells (d5w_swz.py)
        sv[i] = ok ? (unsigned)(((((b << li) << li) + ih) << li) + iw) * (FB_CIN * 2) + g * 16 : 0xFFFFFFF0u;
        sdst[i] = piece < 13 ? (unsigned)(piece * 1024) : (unsigned)(SW_NX * SW_XB);     // surplus pieces: the sink (ring-slot independent)
    }
    auto stage = [&](int d, int sp) {              // sp = d % SW_NX, passed so that the unrolled steps see a constant
        // a plane outside [0, n): every lane out of range by its OFFSET.  (A descriptor of zero records is not a substitute: the
        // zero-fill of the virtual plane d = n then went missing now and then and od = 2n - 1 read the stale slot -- found by the
        // B = 256 cross-check against the box form, profiles/microbench/chk_e1_d5.py.)
        const bool din = (unsigned)d < (unsigned)n;
        const unsigned soff = (unsigned)__builtin_amdgcn_readfirstlane(din ? (d << (2 * li)) * (FB_CIN * 2) : 0);
        const unsigned slot = ldsx + sp * SW_XB;
#pragma unroll
        for (int i = 0; i < 4; ++i) vv_dma16(rs, din ? sv[i] : 0xFFFFFFF0u, soff, wv * 4 + i < 13 ? slot + sdst[i] : ldsx + sdst[i]);
    };
    stage(0, 0);

    // Weights as the first MFMA operand (16 rows n = td * 4 + th, K = 128): for output-column parity pw the K halves are the taps
    // (tw 1 | tw 3) of (centre | left) for pw = 0 and (tw 0 | tw 2) of (right | centre) for pw = 1; lane (n = lane & 15, kq = lane >> 4)
    // holds 8 input channels of k-step ks, straight from the Keras array [64 taps][64 ci]
    const int c16 = lane & 15, kq = lane >> 4;
    uint4 wf[2][4];
#pragma unroll
    for (int pw = 0; pw < 2; ++pw)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int tw = pw == 0 ? (ks < 2 ? 1 : 3) : (ks < 2 ? 0 : 2);
            const float *wr = w + (c16 * 4 + tw) * FB_CIN + (ks & 1) * 32 + kq * 8;
            const f32x4 w0v = *reinterpret_cast<const f32x4 *>(wr), w1v = *reinterpret_cast<const f32x4 *>(wr + 4);
            bf16x8 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) { o[e] = static_cast<__bf16>(w0v[e]); o[4 + e] = static_cast<__bf16>(w1v[e]); }
            wf[pw][ks] = *reinterpret_cast<const uint4 *>(&o);
        }
    // this wave's row tiles: tile T = 16 centre cells (zh = 2T, 2T + 1; zw = 1 .. 8); wave w owns tile w, wave 0 tile 4 as well
    const int ntl = wv == 0 ? 2 : 1;
    const int rowC = (2 * wv + (c16 >> 3)) * 10 + 1 + (c16 & 7);     // halo row of this lane's centre cell in tile wv (tile 4: + 80)
    unsigned xo[3][2];                                           // [centre, left, right][channel half]: byte offset of this lane's operand slot
#pragma unroll
    for (int s3 = 0; s3 < 3; ++s3)
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            const int sh = s3 == 0 ? 0 : s3 == 1 ? -1 : 1, zwc = 1 + (c16 & 7) + sh;
            xo[s3][hf] = (unsigned)((rowC + sh) * 128 + (((hf * 4 + kq) ^ (zwc & 7)) << 4));
        }
    // Q row of a centre cell pr = zh * 8 + (zw - 1): 16 granules of 8 B = (pw 0, pw 1) of n = td * 4 + th, granule g stored at g ^ key,
    // key = ((zh + zw - 1) & 7) << 1 (bit 0 clear: the th pair of a 16-byte store stays adjacent): every ds_write_b128 lane group of the
    // publish and both 32-lane passes of every ds_read_b64 of the gather are conflict-free (profiles/microbench/d5w_swz.py: exhaustive over
    // linear keys under the guide's lane-group / bank model; the first key tried was 2-way on the stores)
    auto qkey = [](int pr) { return (((pr & 7) + (pr >> 3)) & 7) << 1; };

    // gather role: s = od parity slot, ohh = output row inside the tile, mw = cell column (both pw per lane)
    const int mw = tid & 7, ohh = (tid >> 3) & 15, sl = tid >> 7;
    const int mh = ohh >> 1, ph = ohh & 1;
    const int lo = li + 1, n2 = 2 * n;
    const int oh = 2 * h0 + ohh, ow = 2 * (w0 + mw);
    const float hi = 1.0f - epsilon;
    float bce = 0.f, tp = 0.f, fp = 0.f, fn = 0.f;
    float lo0 = 0.f, lo1 = 0.f;                                  // td in {2,3} contributions of Q_{d-1} to this step's outputs (Q_{-1} = 0)

    // Q_d[n][cell][pw] for this wave's tiles: D[n][cell], weights first, K = (centre | left) / (right | centre) channels
    auto mfma_plane = [&](int sp, f32x4 (&acc)[2][2]) {           // sp = ring slot of the plane; acc[tile][pw]
        const char *Xd = Xs + sp * SW_XB;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            acc[t][0] = f32x4{0.f, 0.f, 0.f, 0.f};
            acc[t][1] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (t < ntl) {
                const char *Xt = Xd + t * (80 * 128);             // tile 4 = tile 0 + 8 halo rows of 10 cells: same zw, same slot keys
                uint4 fc[2], fl[2], fr2[2];
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) {
                    fc[hf] = *reinterpret_cast<const uint4 *>(Xt + xo[0][hf]);
                    fl[hf] = *reinterpret_cast<const uint4 *>(Xt + xo[1][hf]);
                    fr2[hf] = *reinterpret_cast<const uint4 *>(Xt + xo[2][hf]);
                }
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {                  // the two accumulators alternate: no chain of dependent MFMAs
                    const uint4 &a0 = ks < 2 ? fc[ks] : fl[ks - 2], &a1 = ks < 2 ? fr2[ks] : fc[ks - 2];
                    acc[t][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8 *>(&wf[0][ks]),
                                                                        *reinterpret_cast<const bf16x8 *>(&a0), acc[t][0], 0, 0, 0);
                    acc[t][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8 *>(&wf[1][ks]),
                                                                        *reinterpret_cast<const bf16x8 *>(&a1), acc[t][1], 0, 0, 0);
                }
            }
        }
    };

    // Software pipeline: step d publishes Q_d (computed during step d-1) and then runs the MFMAs of plane d+1 in the same
    // instruction stream as the gather / loss math of plane d (matrix pipe under the VALU and LDS work).
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // plane 0
    __syncthreads();                                             // ... for every wave
    f32x4 acc[2][2];
    mfma_plane(0, acc);
#pragma unroll
    for (int k = 1; k <= SW_DEPTH; ++k) stage(k, k);
    __syncthreads();                                             // slot 0 may be refilled from the first step on

    int oldh = 0;                                                // ring slot of plane d
#pragma unroll 1
    for (int d = 0; d <= n; ++d) {
        // weights first: lane = centre cell (lane & 15), td = lane >> 4, registers walk th: (th 0, th 1) and (th 2, th 3) with both pw
        // are two 16-byte stores per tile
#pragma unroll
        for (int t = 0; t < 2; ++t)
            if (t < ntl) {
                const int pr = (wv + 4 * t) * 16 + c16, key = qkey(pr);
                char *row = Pq + pr * 128;
                *reinterpret_cast<f32x4 *>(row + (((4 * kq) ^ key) << 3)) = f32x4{acc[t][0][0], acc[t][1][0], acc[t][0][1], acc[t][1][1]};
                *reinterpret_cast<f32x4 *>(row + (((4 * kq + 2) ^ key) << 3)) = f32x4{acc[t][0][2], acc[t][1][2], acc[t][0][3], acc[t][1][3]};
            }
        const int od = 2 * d - 1 + sl;
        const bool ovalid = (unsigned)od < (unsigned)n2;
        const size_t o = ((((((size_t)b << lo) + (ovalid ? od : 0)) << lo) + oh) << lo) + ow;
        // The target pair is loaded by inline asm so that its wait can be counted (see final_bce_sweep_kernel for the history and the
        // compiler-managed forms that fail).  In flight, oldest first: [plane d+1 x4][stores d-1] [y d][plane d+2 x4]: "all but the newest
        // 5" covers plane d+1 whatever the number of stores.  tests/test_isa_lint.py checks on the generated code that no instruction
        // names the pair between this load and the counted wait below that lands it ("+v"(y)).
        float2 y;
        asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(y) : "v"(target + o) : "memory");
        stage(d + 1 + SW_DEPTH, oldh);
        // depth 1: [plane d+1 x4][stores d-1][y d][plane d+2 x4] -> all but the newest 5.  depth 2: plane d+1 is followed by
        // stores d-2 (0..2), y d-1, plane d+2 x4, stores d-1 (0..2), y d, plane d+3 x4 = 10..14 operations -> all but the newest 10
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(SW_DEPTH == 1 ? 5 : 10) : "memory");         // plane d+1 has landed
        __syncthreads();                                         // ... for every wave; Q_d is published

        f32x4 acc_next[2][2];
        const int nexth = oldh + 1 == SW_NX ? 0 : oldh + 1;      // ring slot of plane d+1
        mfma_plane(nexth, acc_next);

        // gather: the w direction is already summed inside Q, so an output pair (pw 0, pw 1) takes ONE 8-byte read per (ah, td): the
        // td = sl entries complete the output planes od = 2d - 1 + sl together with the td = 2 + sl entries of Q_{d-1}, which were read a
        // step ago into (lo0, lo1): Q is read in the step that publishes it, so one buffer holds it.
        float l0 = lo0, l1 = lo1;
        lo0 = 0.f; lo1 = 0.f;
#pragma unroll
        for (int ah = 0; ah < 2; ++ah) {
            const int zh = mh + ph - ah + 1, th = 1 - ph + 2 * ah;
            const int pr = zh * 8 + mw, key = qkey(pr);
            const char *row = Pq + pr * 128;
            const float2 a = *reinterpret_cast<const float2 *>(row + (((sl * 4 + th) ^ key) << 3));
            const float2 bq = *reinterpret_cast<const float2 *>(row + ((((2 + sl) * 4 + th) ^ key) << 3));
            l0 += a.x; l1 += a.y;
            lo0 += bq.x; lo1 += bq.y;
        }
        asm volatile("s_waitcnt vmcnt(4)" : "+v"(y) : : "memory");   // y has landed; plane d+2 may still be in flight
        if (ovalid) {
            const float l[2] = {l0, l1}, yy[2] = {y.x, y.y};
            float p[2];
            // Occupancy targets are 0 or 1: then exactly one of the two logarithms of binary_loss has a non-zero factor, and the other
            // term is +-0 -- one v_log_f32 per voxel instead of two, the same sum.  Any other target value in the wave takes the general form.
            const bool soft = __builtin_amdgcn_ballot_w64((yy[0] != 0.f && yy[0] != 1.f) || (yy[1] != 0.f && yy[1] != 1.f)) != 0;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                p[e] = __builtin_amdgcn_rcpf(1.0f + __expf(-l[e]));
                const float q = fminf(fmaxf(p[e], epsilon), hi);
                if (soft) bce -= gamma * yy[e] * __logf(q) + (1.0f - gamma) * (1.0f - yy[e]) * __logf(1.0f - q);
                else bce -= (yy[e] != 0.f ? gamma : 1.0f - gamma) * __logf(yy[e] != 0.f ? q : 1.0f - q);
                const float yh = l[e] >= 0.f ? 1.f : 0.f;
                tp += yy[e] * yh; fp += (1.f - yy[e]) * yh; fn += yy[e] * (1.f - yh);
            }
            if (probs) *reinterpret_cast<float2 *>(probs + o) = make_float2(p[0], p[1]);
            if (logits) *reinterpret_cast<float2 *>(logits + o) = make_float2(l[0], l[1]);
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) { acc[t][0] = acc_next[t][0]; acc[t][1] = acc_next[t][1]; }
        oldh = nexth;
        __syncthreads();      // every gather of Q_d and every read of plane d+1 is done: publish d+1, refill its slot
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // the last (all-zero) look-ahead planes
    bce = vv_wave_sum(bce); tp = vv_wave_sum(tp); fp = vv_wave_sum(fp); fn = vv_wave_sum(fn);
    if (lane == 0) { red[wv][0] = bce; red[wv][1] = tp; red[wv][2] = fp; red[wv][3] = fn; }
    __syncthreads();
    if (tid < 4) partials[((size_t)b * ntile + tile) * 4 + tid] = red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
}

// ---------------------------------------------------------------------------------------------------------------
__device__ float vv_zero_word = 0.f;

// first_conv (bf16): Conv3D k4 s2 SAME with ONE input channel -- a [rows x 64 taps] x [64 x 64] product per 128-row
// tile with the taps gathered from the float32 occupancy grid.  The layer is latency-bound (one K chunk per tile: gather
// -> LDS -> 8 MFMAs -> store, PMC: 68 % of wave time parked on vmcnt), so each workgroup walks several tiles and issues
// the NEXT tile's 32 gathers per thread before it multiplies and stores the current one.
__global__ __launch_bounds__(256) void first_conv_bf16_kernel(const float *__restrict__ x, const __bf16 *__restrict__ wp,
                                                              const float *__restrict__ scale, const float *__restrict__ shift,
                                                              __bf16 *__restrict__ y, int batch, int din_log2, int act) {
    constexpr int COUT = 64, EPITCH = COUT * 2 + 16;
    __shared__ __attribute__((aligned(16))) char Bs[64 * 128];         // [co][64 taps] bf16, slot-swizzled
    __shared__ __attribute__((aligned(16))) char As[128 * 128];        // [row][64 taps] bf16, slot-swizzled
    __shared__ __attribute__((aligned(16))) char Es[128 * EPITCH];     // output tile [row][64 co] bf16
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int li = din_log2, lo = li - 1, n = 1 << li, omsk = (1 << lo) - 1;
    const long M = (long)batch << (3 * lo);
    const int ntiles = (int)((M + 127) >> 7);
    const int pos = tid & 7, r0 = tid >> 3;
    const int gchunk = pos ^ ((r0 >> 1) & 7);

    for (int i = tid; i < 64 * 8; i += 256) {            // weights: 64 rows x 8 slots, once per workgroup
        const int row = i >> 3, slot = i & 7;
        *reinterpret_cast<uint4 *>(Bs + fm_lds_off(row, slot)) = *reinterpret_cast<const uint4 *>(wp + row * 64 + slot * 8);
    }

    // The gather only ISSUES loads (invalid taps read a zero word through a selected address, so nothing consumes the
    // values here); conversion to bf16 happens when the tile is written to LDS, one loop iteration later.
    float raw[4][8];
    auto gather = [&](int tile) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const long m = (long)tile * 128 + r0 + 32 * i;
            const int ow = (int)(m & omsk), oh = (int)((m >> lo) & omsk), od = (int)((m >> (2 * lo)) & omsk);
            const long b = m >> (3 * lo);
            const int d0 = 2 * od - 1, h0 = 2 * oh - 1, w0 = 2 * ow - 1;
#pragma unroll
            for (int r = 0; r < 2; ++r) {                 // slot = taps (td, th0 + r, tw 0..3): one run of 4 voxels along w
                const int td = gchunk >> 1, th = ((gchunk & 1) << 1) + r;
                const bool ok = m < M && (unsigned)(d0 + td) < (unsigned)n && (unsigned)(h0 + th) < (unsigned)n;
                const float *xr = x + ((((((b << li) + d0 + td) << li) + h0 + th) << li) + w0);
                const float *p0 = (ok && w0 >= 0) ? xr : &vv_zero_word, *p1 = ok ? xr + 1 : &vv_zero_word;
                const float *p2 = ok ? xr + 2 : &vv_zero_word, *p3 = (ok && w0 + 3 < n) ? xr + 3 : &vv_zero_word;
                raw[i][4 * r + 0] = *p0; raw[i][4 * r + 1] = *p1; raw[i][4 * r + 2] = *p2; raw[i][4 * r + 3] = *p3;
            }
        }
    };

    const int fr = lane & 31, fh = lane >> 5;
    int tile = blockIdx.x;
    if (tile < ntiles) gather(tile);
    for (; tile < ntiles; tile += gridDim.x) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            bf16x8 v;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = static_cast<__bf16>(raw[i][e]);
            *reinterpret_cast<bf16x8 *>(As + (r0 + 32 * i) * 128 + pos * 16) = v;
        }
        __syncthreads();
        if (tile + (int)gridDim.x < ntiles) gather(tile + gridDim.x);    // in flight during the MFMAs and the stores below

        f32x16 acc[2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[i][q] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const uint4 fb = *reinterpret_cast<const uint4 *>(Bs + fm_lds_off(wn * 32 + fr, ks * 2 + fh));
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const uint4 fa = *reinterpret_cast<const uint4 *>(As + fm_lds_off(wm * 64 + i * 32 + fr, ks * 2 + fh));
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8 *>(&fb),
                                                                 *reinterpret_cast<const bf16x8 *>(&fa), acc[i], 0, 0, 0);   // D[co][row]
            }
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int c = wn * 32 + 8 * g + 4 * fh;
            f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
            if (scale) sc = *reinterpret_cast<const f32x4 *>(scale + c);
            if (shift) sh = *reinterpret_cast<const f32x4 *>(shift + c);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                bf16x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float t = acc[i][4 * g + e] * sc[e] + sh[e];
                    if (act == VV_ACT_ELU) t = t > 0.f ? t : __expf(t) - 1.f;
                    else if (act == VV_ACT_RELU) t = fmaxf(t, 0.f);
                    else if (act == VV_ACT_LRELU) t = t > 0.f ? t : 0.3f * t;
                    o[e] = static_cast<__bf16>(t);
                }
                *reinterpret_cast<bf16x4 *>(Es + (wm * 64 + i * 32 + fr) * EPITCH + c * 2) = o;
            }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = tid + 256 * i, rl = idx >> 3, c = idx & 7;
            const long m = (long)tile * 128 + rl;
            if (m < M) *reinterpret_cast<uint4 *>(reinterpret_cast<char *>(y) + m * (COUT * 2) + c * 16) =
                           *reinterpret_cast<const uint4 *>(Es + rl * EPITCH + c * 16);
        }
    }
}

// first_conv, plane form (bf16, 32 <= D <= 256): one work item = 256 outputs of ONE output plane (all D/2 columns x
// 512/D rows) x 64 channels.  Its input is 4 consecutive occupancy planes (2 od - 1 .. 2 od + 2), full-width rows: a
// single contiguous stream of float4 loads per item (the gather form above re-reads every voxel 8x as scattered dwords).
// The planes are kept in LDS as bf16 with a one-voxel left pad, S[c + 1] = x[c], so that dword j of a row holds
// (x[2j-1], x[2j]): the 4 taps tw = 0..3 of output column ow are dwords ow, ow+1 -- the MFMA B fragment of a lane
// (k = 16 td + 8 fh + j  <->  th = 2 fh + (j>>2), tw = j&3) is two 8-byte LDS reads, and no im2col tile is ever written.
// Weights (64 x 64 taps) live in registers as A fragments for the whole persistent loop; the next item's planes are in
// flight (registers) while the current item multiplies, transposes through LDS and stores its contiguous 32 KiB.
template <int NI, bool OUT8>       // OUT8: store e4m3fn (64-byte rows) for an fp8 second layer instead of bf16
__global__ __launch_bounds__(256, 3) void first_conv_plane_kernel(const float *__restrict__ x, const __bf16 *__restrict__ wp,
                                                               const float *__restrict__ scale, const float *__restrict__ shift,
                                                               void *__restrict__ y, int batch, int din_log2, int act, int items_per_wg) {
    constexpr int COUT = 64, EPITCH = COUT * 2;            // output rows are 8 chunks of 16 B, chunk ^ (row & 7); fp8: 4 chunks, chunk ^ (row & 3)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = din_log2, D = 1 << li, lo = li - 1, OW = 1 << lo;
    const int loh = 8 - lo, OH = 1 << loh;                 // output rows per item
    const int R = 2 * OH + 2, PD = (D >> 1) + 2, PP = R * PD;   // tile rows per plane, dwords per row / per plane
    // bf16 output: the tile is double buffered and the outputs go from registers to memory (v_permlane32_swap pairs, 16-byte
    // stores), so an item costs ONE barrier (tile published) -- the stage transpose, its barrier and its 32 KiB are the fp8
    // output form's only (OUT8: [256][EPITCH] stage in place of the second tile buffer)
    unsigned *tile0 = reinterpret_cast<unsigned *>(smem);  // [4][R][PD] dwords of bf16 pairs
    const int tile_bytes = (4 * PP * 4 + 15) & ~15;
    char *stage = smem + tile_bytes;                       // OUT8 only
    float *ss = reinterpret_cast<float *>(smem + tile_bytes + (OUT8 ? 256 * EPITCH : tile_bytes));   // folded BN: scale[64], shift[64]
    uint4 *wl = reinterpret_cast<uint4 *>(ss + 128);       // weights as A fragments [ks][nt][lane]
    if (tid < 64) ss[tid] = scale ? scale[tid] : 1.f;
    else if (tid < 128) ss[tid] = shift ? shift[tid - 64] : 0.f;
    const int hblocks = OW >> loh;                         // items per output plane
    const long nitems = (long)batch * OW * hblocks;

    // ---- per-thread load slots (the same for every item): slot s = tid + 256 i -> (plane, tile row, float4 column)
    const int qpr = D >> 2, lq = li - 2;                   // float4 per row
    const int nslots = 4 * R * qpr;
    int sp[NI], srr[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int s = tid + 256 * i, row = s >> lq;        // row = plane * R + rr
        sp[i] = s < nslots ? row / R : -1;
        srr[i] = row - (s < nslots ? row / R : 0) * R;
    }
    const int m4 = tid & (qpr - 1);                        // float4 column (256 % qpr == 0)

    // ---- weights as A fragments: [ks][nt], lane (co = nt*32 + lane&31, k = ks*16 + 8*(lane>>5) + j)
    const int fr = lane & 31, fh = lane >> 5;
    for (int i = wave; i < 8; i += 4)                      // i = ks*2 + nt
        wl[i * 64 + lane] = *reinterpret_cast<const uint4 *>(wp + ((i & 1) * 32 + fr) * 64 + (i >> 1) * 16 + 8 * fh);

    float4 raw[NI];
    auto fetch = [&](long item) {
        const int hb = (int)(item % hblocks);
        const long t = item / hblocks;
        const int od = (int)(t & (OW - 1));
        const long b = t >> lo;
        const int d0 = 2 * od - 1, h0 = 2 * hb * OH - 1;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int id = d0 + sp[i], ih = h0 + srr[i];
            const bool ok = sp[i] >= 0 && (unsigned)id < (unsigned)D && (unsigned)ih < (unsigned)D;
            raw[i] = ok ? *reinterpret_cast<const float4 *>(x + ((((b << li) + id) << li) + ih << li) + 4 * m4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto pack2 = [](float a, float b) -> unsigned {
        const __bf16 ha = static_cast<__bf16>(a), hb = static_cast<__bf16>(b);
        return (unsigned)__builtin_bit_cast(unsigned short, ha) | ((unsigned)__builtin_bit_cast(unsigned short, hb) << 16);
    };

    const long item0 = (long)blockIdx.x * items_per_wg;
    const long item_end = item0 + items_per_wg < nitems ? item0 + items_per_wg : nitems;
    if (item0 < item_end) fetch(item0);
    auto run = [&](auto act_c) {
    constexpr int ACT = decltype(act_c)::value;
    int buf = 0;
    for (long item = item0; item < item_end; ++item) {
        unsigned *tile = OUT8 ? tile0 : tile0 + buf * (tile_bytes >> 2);
        buf ^= 1;
        // ---- planes -> LDS (bf16 pairs, left pad)
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            float left = __shfl_up(raw[i].w, 1);
            if (m4 == 0) left = 0.f;
            if (sp[i] >= 0) {
                unsigned *dst = tile + (sp[i] * R + srr[i]) * PD + 2 * m4;
                *reinterpret_cast<uint2 *>(dst) = make_uint2(pack2(left, raw[i].x), pack2(raw[i].y, raw[i].z));
                if (m4 == qpr - 1) dst[2] = pack2(raw[i].w, 0.f);
            }
        }
        __syncthreads();
        if (item + 1 < item_end) fetch(item + 1);

        // one 32-output row tile at a time (2 x 16 accumulator registers live): 8 MFMAs, then its epilogue
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            f32x16 acc[2];                                  // [nt]
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[c][e] = 0.f;
            const int o = (wave * 2 + mt) * 32 + fr, ohl = o >> lo, ow = o & (OW - 1);
            const unsigned *t0 = tile + (2 * ohl + 2 * fh) * PD + ow;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const unsigned *t1 = t0 + ks * PP;
                const uint4 xf = make_uint4(t1[0], t1[1], t1[PD], t1[PD + 1]);
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    const uint4 wf = wl[(ks * 2 + nt) * 64 + lane];
                    acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8 *>(&wf),
                                                                      *reinterpret_cast<const bf16x8 *>(&xf), acc[nt], 0, 0, 0);
                }
            }
            if constexpr (!OUT8) {
                // folded BN + activation; lanes fr / fr + 32 hold channels 8g + 0..3 / 8g + 4..7 of output o: swapping the upper
                // half of quad 2j with the lower half of quad 2j + 1 gives every lane 8 consecutive channels (guide T21)
                char *yo = reinterpret_cast<char *>(y) + item * (256 * COUT * 2) + o * (COUT * 2) + fh * 16;
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    u32x2 oq[4];
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int c = nt * 32 + 8 * g + 4 * fh;
                        const f32x4 sc = *reinterpret_cast<const f32x4 *>(ss + c), sh = *reinterpret_cast<const f32x4 *>(ss + 64 + c);
                        f32x4 tv = f32x4{acc[nt][4 * g], acc[nt][4 * g + 1], acc[nt][4 * g + 2], acc[nt][4 * g + 3]};
                        tv = vv_bn_act4<ACT>(tv, sc, sh);
                        bf16x4 ov;
#pragma unroll
                        for (int e = 0; e < 4; ++e) ov[e] = static_cast<__bf16>(tv[e]);
                        oq[g] = *reinterpret_cast<const u32x2 *>(&ov);
                    }
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        auto rx = __builtin_amdgcn_permlane32_swap(oq[2 * j][0], oq[2 * j + 1][0], false, false);
                        auto ry = __builtin_amdgcn_permlane32_swap(oq[2 * j][1], oq[2 * j + 1][1], false, false);
                        *reinterpret_cast<u32x4 *>(yo + nt * 64 + j * 32) = u32x4{rx[0], ry[0], rx[1], ry[1]};
                    }
                }
            } else
            // folded BN + activation, transpose through LDS
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int c = nt * 32 + 8 * g + 4 * fh;
                    const f32x4 sc = *reinterpret_cast<const f32x4 *>(ss + c), sh = *reinterpret_cast<const f32x4 *>(ss + 64 + c);
                    f32x4 tv = f32x4{acc[nt][4 * g], acc[nt][4 * g + 1], acc[nt][4 * g + 2], acc[nt][4 * g + 3]};
                    tv = vv_bn_act4<ACT>(tv, sc, sh);
                    if constexpr (OUT8) {                  // e4m3fn for an fp8 second layer: 64-byte rows
                        *reinterpret_cast<unsigned *>(stage + o * EPITCH + ((((c >> 4) ^ o) & 3) << 4) + (c & 12)) = vv_pack_fp8x4(tv);
                    } else {
                        bf16x4 ov;
#pragma unroll
                        for (int e = 0; e < 4; ++e) ov[e] = static_cast<__bf16>(tv[e]);
                        *reinterpret_cast<bf16x4 *>(stage + o * EPITCH + ((((c >> 3) ^ o) & 7) << 4) + (c & 4) * 2) = ov;
                    }
                }
        }
        if constexpr (OUT8) {
            __syncthreads();
            char *yo = reinterpret_cast<char *>(y) + item * (256 * COUT);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int idx = tid + 256 * i, rl = idx >> 2, c = idx & 3;
                *reinterpret_cast<uint4 *>(yo + (size_t)idx * 16) = *reinterpret_cast<const uint4 *>(stage + rl * EPITCH + (((c ^ rl) & 3) << 4));
            }
        }
    }
    };
    switch (act) {
        case VV_ACT_ELU: run(std::integral_constant<int, VV_ACT_ELU>{}); break;
        case VV_ACT_RELU: run(std::integral_constant<int, VV_ACT_RELU>{}); break;
        case VV_ACT_LRELU: run(std::integral_constant<int, VV_ACT_LRELU>{}); break;
        default: run(std::integral_constant<int, VV_ACT_NONE>{}); break;
    }
}

// first_conv, chained plane form (D = 32 or 64, bf16 or e4m3fn output, at least two consecutive items per workgroup).  One item =
// 256 outputs x 64 channels of ONE output plane od (D = 32: the whole 16 x 16 plane; D = 64: 8 of its 32 rows) from input planes
// 2 od - 1 .. 2 od + 2; the next output plane of the same sample (and row block) needs 2 od + 1 .. 2 od + 4: half of what is
// already in LDS.  The plane form above loads all four planes for every item (16 planes per four-item workgroup, every input plane
// fetched twice chip-wide); here the items are ordered with od fastest, the tile is a ring of four HALF tiles (two planes each), an
// item takes its first half from its predecessor's second one and only the two new planes travel (10 planes per four-item
// workgroup): 37 % fewer load instructions, conversions and LDS writes, 12 instead of 20 prefetch registers -- which is what lets
// FOUR workgroups per CU fit in 128 VGPRs without scratch (the plane form had drifted to 134 = three per CU under a launcher that
// still dealt the items for four: a 1.33-round grid).  Loads go through a buffer descriptor: a slot in the SAME padding (plane -1 / D,
// rows -1 / D) or past the slot list reads offset 0xFFFFFFF0 and comes back as zeros, no exec-masked branch per load.  The e4m3fn
// output goes from registers to memory as well (two v_permlane32_swap per 32 channels give a lane 16 consecutive channels = one
// 16-byte store): no 32 KiB transpose stage, no second barrier, four workgroups per CU instead of three.
// Ring safety with ONE barrier per item: item j reads halves (A_j, B_j); the halves written at the top of item j + 1 are the next one
// or two ring positions, never A_j or B_j (four positions), and nobody is behind item j (everyone passed barrier j + 1's predecessor).
template <int LI, bool OUT8>
__global__ __launch_bounds__(256, 4) void first_conv_chain_kernel(const float *__restrict__ x, const __bf16 *__restrict__ wp,
                                                                  const float *__restrict__ scale, const float *__restrict__ shift,
                                                                  void *__restrict__ y, int batch, int act, int items_per_wg) {
    constexpr int COUT = 64, D = 1 << LI, LO = LI - 1, OW = 1 << LO, OH = 256 / OW, LHB = LO - (8 - LO);   // hblocks = OW / OH = 2^LHB
    constexpr int R = 2 * OH + 2, PD = OW + 2, PP = R * PD, HALF = 2 * PP;   // tile rows per plane, dwords per row / plane / half tile
    constexpr int QPR = D / 4, LQ = LI - 2;                // float4 per input row
    constexpr int NIH = (2 * R * QPR + 255) / 256;         // float4 slots per thread and half (544 / 576 -> 3)
    constexpr int ROW = COUT * (OUT8 ? 1 : 2);             // bytes per output voxel
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    unsigned *ring = reinterpret_cast<unsigned *>(smem);   // [4 halves][2 planes][R][PD] dwords of bf16 pairs (left pad: dword j = (x[2j-1], x[2j]))
    float *ss = reinterpret_cast<float *>(smem + 4 * HALF * 4);
    uint4 *wl = reinterpret_cast<uint4 *>(ss + 128);       // weights as A fragments [ks][nt][lane]
    if (tid < 64) ss[tid] = scale ? scale[tid] : 1.f;
    else if (tid < 128) ss[tid] = shift ? shift[tid - 64] : 0.f;
    const long nitems = ((long)batch << LO) << LHB;

    const int m4 = tid & (QPR - 1);                        // float4 column of every slot of this thread
    int loff[NIH], pl[NIH], rr[NIH];                       // dword offset inside a half (-1: no slot); plane of the half (-4 D: none); tile row
    unsigned soff[NIH];                                    // byte offset from (first plane of the half, first tile row, column 0)
#pragma unroll
    for (int i = 0; i < NIH; ++i) {
        const int s = tid + 256 * i, row = s >> LQ, p = row >= R ? 1 : 0;
        const bool slot = s < 2 * R * QPR;
        rr[i] = row - p * R;
        loff[i] = slot ? (p * R + rr[i]) * PD + 2 * m4 : -1;
        pl[i] = slot ? p : -4 * D;
        soff[i] = (unsigned)((((p << LI) + rr[i]) << LI) + 4 * m4) * 4u;
    }
    const int fr = lane & 31, fh = lane >> 5;
    for (int i = wave; i < 8; i += 4)                      // i = ks*2 + nt
        wl[i * 64 + lane] = *reinterpret_cast<const uint4 *>(wp + ((i & 1) * 32 + fr) * 64 + (i >> 1) * 16 + 8 * fh);

    const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(x), 0, (int)(((unsigned)batch << (3 * LI)) * 4u), 0x00020000);
    // item -> (sample, row block, output plane), od fastest
    auto decode = [&](long item, int &b, int &hb, int &od) {
        od = (int)(item & (OW - 1));
        hb = (int)(item >> LO) & ((1 << LHB) - 1);
        b = (int)(item >> (LO + LHB));
    };
    auto load_half = [&](int b, int hb, int d, f32x4 (&r)[NIH]) {   // planes d, d + 1 of sample b, tile rows of row block hb
        const int h0 = 2 * hb * OH - 1;
        const unsigned base = (unsigned)(((((b << LI) + d) << LI) + h0) << LI) * 4u;     // wraps below zero for d / h0 = -1; valid slots land back in range
#pragma unroll
        for (int i = 0; i < NIH; ++i) {
            const bool ok = (unsigned)(d + pl[i]) < (unsigned)D && (unsigned)(h0 + rr[i]) < (unsigned)D;
            // (whole-vector bit cast: __builtin_bit_cast(float, v[k]) on a vector ELEMENT reads element 0 for every k with this compiler)
            r[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsx, (int)(ok ? base + soff[i] : 0xFFFFFFF0u), 0, 0));
        }
    };
    auto pack2 = [](float a, float b) -> unsigned {
        const __bf16 ha = static_cast<__bf16>(a), hb = static_cast<__bf16>(b);
        return (unsigned)__builtin_bit_cast(unsigned short, ha) | ((unsigned)__builtin_bit_cast(unsigned short, hb) << 16);
    };
    auto write_half = [&](int h, const f32x4 (&r)[NIH]) {
        unsigned *half = ring + h * HALF;
#pragma unroll
        for (int i = 0; i < NIH; ++i) {
            float left = __shfl_up(r[i][3], 1);
            if (m4 == 0) left = 0.f;
            if (loff[i] >= 0) {
                unsigned *dst = half + loff[i];
                *reinterpret_cast<uint2 *>(dst) = make_uint2(pack2(left, r[i][0]), pack2(r[i][1], r[i][2]));
                if (m4 == QPR - 1) dst[2] = pack2(r[i][3], 0.f);
            }
        }
    };

    const long item0 = (long)blockIdx.x * items_per_wg;
    const long item_end = item0 + items_per_wg < nitems ? item0 + items_per_wg : nitems;
    f32x4 raw[NIH];                                        // the second half (planes 2 od + 1, 2 od + 2) of the item about to run
    if (item0 < item_end) {
        int b, hb, od;
        decode(item0, b, hb, od);
        load_half(b, hb, 2 * od + 1, raw);
    }
    auto run = [&](auto act_c) {
    constexpr int ACT = decltype(act_c)::value;
    int nxt = 0, hA = 0, hB = 0;
    for (long item = item0; item < item_end; ++item) {
        int b, hb, od;
        decode(item, b, hb, od);
        if (item == item0 || od == 0) {                    // no predecessor in this workgroup / for this row block: planes 2 od - 1, 2 od as well
            f32x4 ra[NIH];
            load_half(b, hb, 2 * od - 1, ra);
            write_half(nxt, ra);
            hA = nxt;
            nxt = (nxt + 1) & 3;
        } else hA = hB;
        write_half(nxt, raw);
        hB = nxt;
        nxt = (nxt + 1) & 3;
        __syncthreads();
        if (item + 1 < item_end) {
            int nb, nhb, nod;
            decode(item + 1, nb, nhb, nod);
            load_half(nb, nhb, 2 * nod + 1, raw);
        }
        const unsigned *tA = ring + hA * HALF, *tB = ring + hB * HALF;
        const long oitem = ((((long)b << LO) + od) << LHB) + hb;      // the output is (sample, plane, row block) major
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            f32x16 acc[2];                                  // [nt]
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[c][e] = 0.f;
            const int o = (wave * 2 + mt) * 32 + fr, ohl = o >> LO, ow = o & (OW - 1);
            const int ti = (2 * ohl + 2 * fh) * PD + ow;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const unsigned *t1 = (ks < 2 ? tA + ks * PP : tB + (ks - 2) * PP) + ti;
                const uint4 xf = make_uint4(t1[0], t1[1], t1[PD], t1[PD + 1]);
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    const uint4 wf = wl[(ks * 2 + nt) * 64 + lane];
                    acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8 *>(&wf),
                                                                      *reinterpret_cast<const bf16x8 *>(&xf), acc[nt], 0, 0, 0);
                }
            }
            // folded BN + activation; lanes fr / fr + 32 hold channels 8g + 0..3 / 8g + 4..7 of output o
            char *yo = reinterpret_cast<char *>(y) + oitem * (256 * ROW) + o * ROW;
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                u32x2 oq[4];
                unsigned o8[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int c = nt * 32 + 8 * g + 4 * fh;
                    const f32x4 sc = *reinterpret_cast<const f32x4 *>(ss + c), sh = *reinterpret_cast<const f32x4 *>(ss + 64 + c);
                    f32x4 tv = f32x4{acc[nt][4 * g], acc[nt][4 * g + 1], acc[nt][4 * g + 2], acc[nt][4 * g + 3]};
                    tv = vv_bn_act4<ACT>(tv, sc, sh);
                    if constexpr (OUT8) o8[g] = vv_pack_fp8x4(tv);
                    else {
                        bf16x4 ov;
#pragma unroll
                        for (int e = 0; e < 4; ++e) ov[e] = static_cast<__bf16>(tv[e]);
                        oq[g] = *reinterpret_cast<const u32x2 *>(&ov);
                    }
                }
                if constexpr (OUT8) {
                    // dword g of lane half fh = channels 8g + 4fh .. + 3.  swap(g + 2, g): the upper half of dword g + 2 goes to the lower
                    // lanes' dword g and back -- lower lanes end with (g + 2: fh 0, fh 1) = 8 consecutive channels of group g + 2, upper
                    // lanes with those of group g: lane half 0 stores channels 16 .. 31, lane half 1 channels 0 .. 15 of this 32-block
                    auto r0 = __builtin_amdgcn_permlane32_swap(o8[2], o8[0], false, false);
                    auto r1 = __builtin_amdgcn_permlane32_swap(o8[3], o8[1], false, false);
                    *reinterpret_cast<u32x4 *>(yo + nt * 32 + (1 - fh) * 16) = u32x4{r0[0], r0[1], r1[0], r1[1]};
                } else {
                    // swapping the upper half of quad 2j with the lower half of quad 2j + 1 gives every lane 8 consecutive channels (guide T21)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        auto rx = __builtin_amdgcn_permlane32_swap(oq[2 * j][0], oq[2 * j + 1][0], false, false);
                        auto ry = __builtin_amdgcn_permlane32_swap(oq[2 * j][1], oq[2 * j + 1][1], false, false);
                        *reinterpret_cast<u32x4 *>(yo + fh * 16 + nt * 64 + j * 32) = u32x4{rx[0], ry[0], rx[1], ry[1]};
                    }
                }
            }
        }
    }
    };
    switch (act) {
        case VV_ACT_ELU: run(std::integral_constant<int, VV_ACT_ELU>{}); break;
        case VV_ACT_RELU: run(std::integral_constant<int, VV_ACT_RELU>{}); break;
        case VV_ACT_LRELU: run(std::integral_constant<int, VV_ACT_LRELU>{}); break;
        default: run(std::integral_constant<int, VV_ACT_NONE>{}); break;
    }
}

__global__ __launch_bounds__(64) void final_reduce_kernel(const float *__restrict__ partials, float *__restrict__ stats, int nblk) {
    const int b = blockIdx.x, lane = threadIdx.x;
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    for (int i = lane; i < nblk; i += 64) {
        const f32x4 v = *reinterpret_cast<const f32x4 *>(partials + ((size_t)b * nblk + i) * 4);
        s[0] += v[0]; s[1] += v[1]; s[2] += v[2]; s[3] += v[3];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) s[k] = vv_wave_sum(s[k]);
    if (lane == 0) *reinterpret_cast<f32x4 *>(stats + (size_t)b * 4) = f32x4{s[0], s[1], s[2], s[3]};
}

// final_reduce + shape_metrics (nolbo.py:1498-1501) in ONE launch for the usual case of a few partial blocks per sample: thread
// b sums its sample's partials in block order (as final_reduce does), keeps (bce, TP/(TP+FP+1e-10), TP/(TP+FN+1e-10), IoU), and
// the batch means are formed by a fixed tree (wave shuffles, then the waves in order): deterministic, independent of timing.
__global__ __launch_bounds__(256) void final_reduce_metrics_kernel(const float *__restrict__ partials, float *__restrict__ stats,
                                                                   float *__restrict__ out4, int nblk, int batch) {
    __shared__ float red[4][4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float m[4] = {0.f, 0.f, 0.f, 0.f};
    for (int b = tid; b < batch; b += 256) {
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        for (int i = 0; i < nblk; ++i) s += *reinterpret_cast<const f32x4 *>(partials + ((size_t)b * nblk + i) * 4);
        *reinterpret_cast<f32x4 *>(stats + (size_t)b * 4) = s;
        const float tp = s[1], fp = s[2], fn = s[3];
        m[0] += s[0];
        m[1] += tp / (tp + fp + 1e-10f);
        m[2] += tp / (tp + fn + 1e-10f);
        m[3] += tp / fmaxf(tp + fp + fn, 1.f);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) m[k] = vv_wave_sum(m[k]);
    if (lane == 0) { red[wave][0] = m[0]; red[wave][1] = m[1]; red[wave][2] = m[2]; red[wave][3] = m[3]; }
    __syncthreads();
    if (tid < 4) out4[tid] = (red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid]) / (float)batch;
}

__global__ __launch_bounds__(64) void final_metrics_kernel(const float *__restrict__ stats, float *__restrict__ out4, int batch) {
    const int lane = threadIdx.x;
    float m[4] = {0.f, 0.f, 0.f, 0.f};
    for (int b = lane; b < batch; b += 64) {
        const f32x4 s = *reinterpret_cast<const f32x4 *>(stats + (size_t)b * 4);
        const float tp = s[1], fp = s[2], fn = s[3];
        m[0] += s[0];
        m[1] += tp / (tp + fp + 1e-10f);
        m[2] += tp / (tp + fn + 1e-10f);
        m[3] += tp / fmaxf(tp + fp + fn, 1.f);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) m[k] = vv_wave_sum(m[k]);
    if (lane < 4) out4[lane] = (lane == 0 ? m[0] : lane == 1 ? m[1] : lane == 2 ? m[2] : m[3]) / (float)batch;
}

// per-sample sums from the partial blocks, and the batch metrics when the caller wants them
void finish_stats(const float *partials, float *stats, float *metrics4, int nblk, int batch, hipStream_t st) {
    if (metrics4 && nblk <= 64) {
        VV_LAUNCH(final_reduce_metrics_kernel, dim3(1), dim3(256), 0, st, partials, stats, metrics4, nblk, batch);
        return;
    }
    VV_LAUNCH(final_reduce_kernel, dim3(batch), dim3(64), 0, st, partials, stats, nblk);
    if (metrics4) VV_LAUNCH(final_metrics_kernel, dim3(1), dim3(64), 0, st, stats, metrics4, batch);
}

int final_bce_impl(const void *x, const float *w_keras, const float *target, float *probs, float *logits, float *stats, float *metrics4,
                   int batch, int side, int cin, float gamma, float epsilon, int dtype, void *workspace, size_t workspace_bytes, void *stream);

}  // namespace

VV_EXPORT size_t vv_convT3d_final_bce_workspace_bytes(int batch, int side) {
    const size_t nb = side / 4;
    return (size_t)batch * nb * nb * nb * 4 * sizeof(float);
}

VV_EXPORT int vv_convT3d_final_bce_fwd(const void *x, const float *w_keras, const float *target, float *probs,
                                       float *logits, float *stats, int batch, int side, int cin, float gamma,
                                       float epsilon, int dtype, void *workspace, size_t workspace_bytes, void *stream) {
    return final_bce_impl(x, w_keras, target, probs, logits, stats, nullptr, batch, side, cin, gamma, epsilon, dtype, workspace,
                          workspace_bytes, stream);
}

VV_EXPORT int vv_convT3d_final_bce_metrics_fwd(const void *x, const float *w_keras, const float *target, float *probs,
                                               float *logits, float *stats, float *metrics4, int batch, int side, int cin, float gamma,
                                               float epsilon, int dtype, void *workspace, size_t workspace_bytes, void *stream) {
    if (!metrics4) return VV_ERR_NULL;
    return final_bce_impl(x, w_keras, target, probs, logits, stats, metrics4, batch, side, cin, gamma, epsilon, dtype, workspace,
                          workspace_bytes, stream);
}

namespace {
int final_bce_impl(const void *x, const float *w_keras, const float *target, float *probs, float *logits, float *stats, float *metrics4,
                   int batch, int side, int cin, float gamma, float epsilon, int dtype, void *workspace, size_t workspace_bytes, void *stream) {
    if (!x || !w_keras || !target || !stats) return VV_ERR_NULL;
    if (dtype != VV_F32 && dtype != VV_BF16 && dtype != VV_FP8) return VV_ERR_DTYPE;
    if (batch <= 0 || batch > 65535 || side < 4 || !vv_is_pow2(side) || cin != FB_CIN) return VV_ERR_SHAPE;
    if (dtype == VV_FP8 && side < 8) return VV_ERR_SHAPE;                  // the e4m3fn input exists in sweep form only
    if (!vv_aligned16(x) || !vv_aligned16(target) || (probs && !vv_aligned16(probs)) || (logits && !vv_aligned16(logits)))
        return VV_ERR_ALIGN;
    if (!workspace || workspace_bytes < vv_convT3d_final_bce_workspace_bytes(batch, side) || !vv_aligned16(workspace))
        return VV_ERR_WORKSPACE;
    const int nb = side / 4, nblk = nb * nb * nb;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    float *partials = reinterpret_cast<float *>(workspace);
    const int ntile = (side / 8) * (side / 8);
    if (dtype == VV_FP8) {
        const int nt8 = vv_final_bce_sweep_fp8_launch(x, w_keras, target, probs, logits, partials, batch, side, gamma, epsilon, st);
        finish_stats(partials, stats, metrics4, nt8, batch, st);
        return vv_launch_status();
    }
    const char *force = vv_hook("VV_FINAL_BCE");                  // "sweep" / "sweepp" / "box": override the batch heuristic (tests)
    const bool sweep = dtype == VV_BF16 && side >= 8 &&
                       (force ? force[0] == 's' : (long)batch * ntile >= 128);   // enough workgroups to fill the chip
    // "sweep" = the form with the w direction summed inside the MFMA (round 3); "sweepp" = the form that publishes P[halo cell][64 taps]
    const bool form_p = force && !strcmp(force, "sweepp");
    if (sweep) {
        static const bool attr = [] {
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&final_bce_sweep_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, SW_LDS);
            return true;
        }();
        (void)attr;
        // 32-bit buffer offsets: <= 2 GiB of input per launch; every per-sample tensor moves on by the same sample range
        const size_t in_per = (size_t)side * side * side * FB_CIN * 2, vox = (size_t)8 * side * side * side;
        const int per = vv_chunk_samples(in_per, batch);
        if (per < 1) return VV_ERR_SHAPE;
        for (int b0 = 0; b0 < batch; b0 += per) {
            const int nbt = batch - b0 < per ? batch - b0 : per;
            if (form_p)
                VV_LAUNCH(final_bce_sweep_kernel, dim3(ntile * nbt), dim3(256), SW_LDS, st,
                          reinterpret_cast<const __bf16 *>(reinterpret_cast<const char *>(x) + (size_t)b0 * in_per), w_keras, target + (size_t)b0 * vox,
                          probs ? probs + (size_t)b0 * vox : nullptr, logits ? logits + (size_t)b0 * vox : nullptr, partials + (size_t)b0 * ntile * 4,
                          vv_log2(side), (unsigned)((size_t)nbt * in_per), gamma, epsilon);
            else
                VV_LAUNCH(final_bce_sweepw_kernel, dim3(ntile * nbt), dim3(256), SWW_LDS, st,
                          reinterpret_cast<const __bf16 *>(reinterpret_cast<const char *>(x) + (size_t)b0 * in_per), w_keras, target + (size_t)b0 * vox,
                          probs ? probs + (size_t)b0 * vox : nullptr, logits ? logits + (size_t)b0 * vox : nullptr, partials + (size_t)b0 * ntile * 4,
                          vv_log2(side), (unsigned)((size_t)nbt * in_per), gamma, epsilon);
        }
        finish_stats(partials, stats, metrics4, ntile, batch, st);
        return vv_launch_status();
    }
    if (dtype == VV_BF16)
        VV_LAUNCH(final_bce_mfma_kernel, dim3(nblk * batch), dim3(256), (size_t)FM_ROWS * 128 + 64 * 128, st,
                  reinterpret_cast<const __bf16 *>(x), w_keras, target, probs, logits, partials, vv_log2(side),
                  (unsigned)((size_t)batch * side * side * side * FB_CIN * 2), gamma, epsilon);
    else
        VV_LAUNCH((final_bce_kernel<float>), dim3(nblk, batch), dim3(256), 0, st, reinterpret_cast<const float *>(x),
                           w_keras, target, probs, logits, partials, vv_log2(side), gamma, epsilon);
    finish_stats(partials, stats, metrics4, nblk, batch, st);
    return vv_launch_status();
}
}  // namespace

// bf16 fast path of vv_conv3d_first_fwd (igemm.hip dispatches here): w_packed = vv_pack_conv_k4(cin = 1) = [64][64] bf16.
int vv_first_conv_bf16_launch(const float *x, const void *w_packed, const float *scale, const float *shift, void *y, int batch,
                              int side, int act, void *stream, int out_fp8) {
    const int li = vv_log2(side);
    if (out_fp8 && !(side >= 32 && side <= 256)) return VV_ERR_DTYPE;      // only the plane-form kernel stores e4m3fn
    if (side >= 32 && side <= 256 && (out_fp8 || !vv_hook("VV_FIRSTCONV_GATHER"))) {
        const int ow = side / 2, oh = 256 / ow, r = 2 * oh + 2, pd = side / 2 + 2;
        const long nitems = (long)batch * ow * (ow / oh);
        const size_t tile_b = ((size_t)4 * r * pd * 4 + 15) & ~(size_t)15;
        const size_t lds = tile_b + (out_fp8 ? (size_t)256 * (64 * 2) : tile_b) + 128 * sizeof(float) + 8 * 64 * 16;
        const int nslots = 4 * r * (side / 4), ni = (nslots + 255) / 256;
        static const long envwg = vv_hook("VV_FIRSTCONV_WGS") ? atol(vv_hook("VV_FIRSTCONV_WGS")) : 0;
        // D = 32 / 64 and batches that give every workgroup a chain of >= 2 consecutive output planes: the chained kernel, FOUR persistent
        // workgroups per CU (D = 32, batch 256: 4,096 items = 1,024 x 4); its input offsets are 32-bit (< 2 GiB of input per launch)
        const bool nochain = vv_hook("VV_FIRSTCONV_NOCHAIN") != nullptr;      // test hook: the plane form at every batch
        if ((side == 32 || side == 64) && !nochain && (size_t)batch * side * side * side * sizeof(float) < 0x7FFFFFFFull) {
            const long maxwg4 = envwg > 0 ? envwg : 256 * 4;
            const int ipw4 = (int)((nitems + maxwg4 - 1) / maxwg4);
            if (ipw4 >= 2) {
                const size_t lds4 = (size_t)4 * 2 * r * pd * 4 + 128 * sizeof(float) + 8 * 64 * 16;
                const dim3 g4((unsigned)((nitems + ipw4 - 1) / ipw4));
                hipStream_t st4 = reinterpret_cast<hipStream_t>(stream);
                const __bf16 *wb4 = reinterpret_cast<const __bf16 *>(w_packed);
                if (side == 32) {
                    if (out_fp8) VV_LAUNCH((first_conv_chain_kernel<5, true>), g4, dim3(256), lds4, st4, x, wb4, scale, shift, y, batch, act, ipw4);
                    else VV_LAUNCH((first_conv_chain_kernel<5, false>), g4, dim3(256), lds4, st4, x, wb4, scale, shift, y, batch, act, ipw4);
                } else {
                    if (out_fp8) VV_LAUNCH((first_conv_chain_kernel<6, true>), g4, dim3(256), lds4, st4, x, wb4, scale, shift, y, batch, act, ipw4);
                    else VV_LAUNCH((first_conv_chain_kernel<6, false>), g4, dim3(256), lds4, st4, x, wb4, scale, shift, y, batch, act, ipw4);
                }
                return vv_launch_status();
            }
        }
        // persistent workgroups of the plane form: what fits a CU at once = 3 (134-168 VGPRs; the bf16-output form at D = 32 once fitted 4)
        const long maxwg = envwg > 0 ? envwg : 256 * 3;
        const int ipw = (int)((nitems + maxwg - 1) / maxwg);
        const int grid = (int)((nitems + ipw - 1) / ipw);
        hipStream_t st = reinterpret_cast<hipStream_t>(stream);
        const __bf16 *wb = reinterpret_cast<const __bf16 *>(w_packed);
        if (ni <= 5) {
            if (out_fp8) VV_LAUNCH((first_conv_plane_kernel<5, true>), dim3(grid), dim3(256), lds, st, x, wb, scale, shift, y, batch, li, act, ipw);
            else VV_LAUNCH((first_conv_plane_kernel<5, false>), dim3(grid), dim3(256), lds, st, x, wb, scale, shift, y, batch, li, act, ipw);
        } else {
            if (out_fp8) VV_LAUNCH((first_conv_plane_kernel<6, true>), dim3(grid), dim3(256), lds, st, x, wb, scale, shift, y, batch, li, act, ipw);
            else VV_LAUNCH((first_conv_plane_kernel<6, false>), dim3(grid), dim3(256), lds, st, x, wb, scale, shift, y, batch, li, act, ipw);
        }
        return vv_launch_status();
    }
    const long M = (long)batch << (3 * (li - 1));
    const int ntiles = (int)((M + 127) / 128);
    const int grid = ntiles < 2048 ? ntiles : 2048;
    VV_LAUNCH(first_conv_bf16_kernel, dim3(grid), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x,
              reinterpret_cast<const __bf16 *>(w_packed), scale, shift, reinterpret_cast<__bf16 *>(y), batch, li, act);
    return vv_launch_status();
}

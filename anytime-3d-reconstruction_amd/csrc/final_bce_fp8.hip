// Last transposed conv (64 -> 1) + sigmoid / BCE / TP / FP / FN, sweep form, with the INPUT in fp8 (OCP e4m3fn): the fp8
// twin of final_bce_sweep_kernel (first_last.hip) for the 'fp8' inference mode, where it halves the largest HBM stream of
// the decoder (the 64-channel activation of the widest layer).  Same decomposition -- one workgroup per 8 x 8 (h, w) cell
// tile swept through the depth, P_d = X_d W^T on MFMA, scatter-form gather of the 8 taps per output voxel, the four sums
// in registers -- with
//   * 64-byte voxel rows: 4 slots of 16 B, four rows share the 64 banks; consecutive rows r, and the four 4-row segments
//     of a ds_read_b128 lane group start at r0 + {0, 12, 20, 24} (or {4, 8, 16, 28}): (row >> 2) & 3 separates them;
//   * ONE v_mfma_scale_f32_32x32x64_f8f6f4 per 32-row tile and tap half (K = the 64 channels);
//   * the float32 Keras kernel quantised in the kernel prologue, per TAP (the MFMA's output row): w / s_tap -> e4m3fn,
//     s_tap = max|w[tap]| / 256, and s_tap multiplied back when P is published -- the logits stay float32 sums of
//     fp8 x fp8 products.
// Everything after P (gather, hardware exp / log / rcp, threshold on the logit) is the bf16 kernel's code.
#include "common.h"

namespace {

typedef __attribute__((ext_vector_type(8))) int i32x8;

constexpr int F8B_CIN = 64;
constexpr int SW_ROWS = 100, SW_PP = 36, SW_PSZ = 100 * SW_PP;                 // as in first_last.hip: P row pitch / buffer floats
constexpr int S8_XB = 7 * 1024, S8_NX = 2;                                     // X slot: 112 rows x 64 B
constexpr int S8_LDS = S8_NX * S8_XB + 1024 + 1024 + 3 * SW_PSZ * 4;           // slots, sink, 16 spare rows, P buffers

__global__ __launch_bounds__(256, 2) void final_bce_sweep_fp8_kernel(const unsigned char *__restrict__ x, const float *__restrict__ w,
                                                                     const float *__restrict__ target, float *__restrict__ probs,
                                                                     float *__restrict__ logits, float *__restrict__ partials,
                                                                     int din_log2, unsigned x_bytes, float gamma, float epsilon) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *Xs = smem;                                             // ring of 2 planes x [112 rows][64 B], slot-swizzled; 1 KiB sink; 16 spare rows
    float *PL = reinterpret_cast<float *>(smem + S8_NX * S8_XB + 2048);   // P_d[td 0,1]      [100][36]
    float *PH = PL + SW_PSZ;                                     // P_d / P_{d-1}[td 2,3]  [2][100][33]
    __shared__ float red[4][4];
    __shared__ float tsc[64];                                    // per-tap weight scales
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
    // plane d -> ring slot d % 2: 7 pieces of 16 rows; every wave issues 2 (the surplus one goes to the sink so that the
    // vector-memory counter advances uniformly); rows >= 100, voxels outside the grid and planes outside [0, n) arrive as
    // zeros (the virtual plane d = n closes the sweep).  The 4th MFMA row tile reads rows 96..127, i.e. 16 rows past the
    // slot: whatever it finds there only reaches accumulator rows >= 104, which are never published.
    auto stage = [&](int d) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int piece = wv * 2 + i, row = piece * 16 + (lane >> 2);
            const int zh = row / 10, zw = row - zh * 10;
            const int ih = h0 - 1 + zh, iw = w0 - 1 + zw;
            const bool ok = row < SW_ROWS && (unsigned)d < (unsigned)n && (unsigned)ih < (unsigned)n && (unsigned)iw < (unsigned)n;
            const int g = (lane & 3) ^ ((row >> 2) & 3);
            const unsigned vo = ok ? (unsigned)((((((b << li) + d) << li) + ih) << li) + iw) * F8B_CIN + g * 16 : 0xFFFFFFF0u;
            vv_dma16(rs, vo, piece < 7 ? ldsx + (d % S8_NX) * S8_XB + piece * 1024 : ldsx + S8_NX * S8_XB);
        }
    };
    stage(0);

    // weights of this wave's tap half as the MFMA's first operand: lane = tap nt*32 + fr, k = channels 32 fh .. 32 fh + 31,
    // quantised per tap (the two lanes of a tap combine their maxima)
    const int nt = wv & 1, mt0 = wv >> 1;
    i32x8 fbv;
    {
        const float *wr = w + (nt * 32 + fr) * F8B_CIN + 32 * fh;
        f32x4 wq[8];
        float mx = 0.f;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            wq[q] = *reinterpret_cast<const f32x4 *>(wr + 4 * q);
#pragma unroll
            for (int e = 0; e < 4; ++e) mx = fmaxf(mx, fabsf(wq[q][e]));
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float s = fmaxf(mx, 1e-30f) / 256.f, inv = 1.f / s;
#pragma unroll
        for (int q = 0; q < 8; ++q) fbv[q] = (int)vv_pack_fp8x4(wq[q] * inv);
        if (wv < 2 && fh == 0) tsc[nt * 32 + fr] = s;
    }
    for (int i = tid; i < SW_PSZ; i += 256) PH[i] = 0.f;         // P_{-1} = 0

    // gather role: s = od parity slot, ohh = output row inside the tile, mw = cell column (both pw per lane)
    const int mw = tid & 7, ohh = (tid >> 3) & 15, sl = tid >> 7;
    const int mh = ohh >> 1, ph = ohh & 1;
    const int lo = li + 1, n2 = 2 * n;
    const int oh = 2 * h0 + ohh, ow = 2 * (w0 + mw);
    const float hi = 1.0f - epsilon;
    float bce = 0.f, tp = 0.f, fp = 0.f, fn = 0.f;
    int oldh = 0;

    // P_d = X_d W^T for this wave's two row tiles and its tap half: D[tap][cell], weights-first; one K = 64 MFMA each
    auto mfma_plane = [&](int d, f32x16 (&acc)[2]) {
        const char *Xd = Xs + (d % S8_NX) * S8_XB;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[j][q] = 0.f;
            const int row = (mt0 + 2 * j) * 32 + fr;
            const char *pa = Xd + row * F8B_CIN + (((fh * 2) ^ ((row >> 2) & 3)) << 4);
            const uint4 lo4 = *reinterpret_cast<const uint4 *>(pa);
            const uint4 hi4 = *reinterpret_cast<const uint4 *>(Xd + ((unsigned)(pa - Xd) ^ 16u));
            const i32x8 xv = {(int)lo4.x, (int)lo4.y, (int)lo4.z, (int)lo4.w, (int)hi4.x, (int)hi4.y, (int)hi4.z, (int)hi4.w};
            acc[j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fbv, xv, acc[j], 0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
        }
    };

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // plane 0
    __syncthreads();                                             // ... the zeroed P_{-1} and the tap scales, for every wave
    // dequantisation factors of this lane's 16 taps: quad g = taps nt*32 + 8g + 4fh .. +3
    f32x4 tsv[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) tsv[g] = *reinterpret_cast<const f32x4 *>(&tsc[nt * 32 + 8 * g + 4 * fh]);

    // Software pipeline: step d publishes P_d (computed during step d-1) and then runs the MFMAs of plane d+1 in the same
    // instruction stream as the gather / loss math of plane d (matrix pipe under the VALU and LDS work).
    f32x16 acc[2];
    mfma_plane(0, acc);
    stage(1);
    __syncthreads();                                             // slot 0 may be refilled (plane 2) from the first step on

#pragma unroll 1
    for (int d = 0; d <= n; ++d) {
        // weights-first: lane = cell row, registers walk the taps of the half; quad g = taps 8g + 4fh .. +3 = the four tw
        // of one (td, th): one 16-byte store per quad
        float *Pw = nt == 0 ? PL : PH + (oldh ^ 1) * SW_PSZ;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int row = (mt0 + 2 * j) * 32 + fr;
            if (row < SW_ROWS) {
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *reinterpret_cast<f32x4 *>(Pw + row * SW_PP + 8 * g + 4 * fh) =
                        f32x4{acc[j][4 * g] * tsv[g][0], acc[j][4 * g + 1] * tsv[g][1], acc[j][4 * g + 2] * tsv[g][2], acc[j][4 * g + 3] * tsv[g][3]};
            }
        }
        const int od = 2 * d - 1 + sl;
        const bool ovalid = (unsigned)od < (unsigned)n2;
        const size_t o = ((((((size_t)b << lo) + (ovalid ? od : 0)) << lo) + oh) << lo) + ow;
        // The target pair is loaded by inline asm so that its wait can be counted: the vector-memory counter retires in
        // order, and a compiler-placed wait for this load would be vmcnt(0), i.e. it would also wait for the 2 pieces of
        // plane d+2 issued right after it -- the look-ahead.  In flight, oldest first:
        //   [plane d+1 x2][stores d-1] [y d][plane d+2 x2]
        // so "all but the newest 3" covers plane d+1 whatever the number of stores (more stores only wait for more).
        float2 y;
        asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(y) : "v"(target + o) : "memory");
        stage(d + 2);
        asm volatile("s_waitcnt vmcnt(3)" ::: "memory");         // plane d+1 (issued a step ago) has landed
        __syncthreads();                                         // ... for every wave; P_d is published

        f32x16 acc_next[2];
        mfma_plane(d + 1, acc_next);

        float l0 = 0.f, l1 = 0.f;
        const float *Pold = PH + oldh * SW_PSZ;
#pragma unroll
        for (int ah = 0; ah < 2; ++ah) {
            const int zh = mh + ph - ah + 1, th = 1 - ph + 2 * ah;
            const int off = (zh * 10 + mw) * SW_PP + (sl * 4 + th) * 4;
            const float *r0 = PL + off, *r1 = Pold + off;
            l0 += r0[SW_PP + 1] + r0[3] + r1[SW_PP + 1] + r1[3];                         // pw = 0
            l1 += r0[2 * SW_PP] + r0[SW_PP + 2] + r1[2 * SW_PP] + r1[SW_PP + 2];   // pw = 1
        }
        asm volatile("s_waitcnt vmcnt(2)" : "+v"(y) : : "memory");   // y has landed; plane d+2 may still be in flight
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
        oldh ^= 1;
        __syncthreads();      // every gather of P_d / P_{d-1} and every read of plane d+1 is done: publish d+1, refill its slot
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // the last (all-zero) look-ahead planes
    bce = vv_wave_sum(bce); tp = vv_wave_sum(tp); fp = vv_wave_sum(fp); fn = vv_wave_sum(fn);
    if (lane == 0) { red[wv][0] = bce; red[wv][1] = tp; red[wv][2] = fp; red[wv][3] = fn; }
    __syncthreads();
    if (tid < 4) partials[((size_t)b * ntile + tile) * 4 + tid] = red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
}


}  // namespace

// x [B][side^3][64] e4m3fn; everything else as vv_convT3d_final_bce_fwd's sweep form (same workspace: per-tile partial sums,
// reduced by the caller's final_reduce pass).  Returns the number of partial blocks per sample.
int vv_final_bce_sweep_fp8_launch(const void *x, const float *w_keras, const float *target, float *probs, float *logits, float *partials,
                                  int batch, int side, float gamma, float epsilon, hipStream_t st) {
    static const bool attr = [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&final_bce_sweep_fp8_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, S8_LDS);
        return true;
    }();
    (void)attr;
    const int ntile = (side / 8) * (side / 8);
    VV_LAUNCH(final_bce_sweep_fp8_kernel, dim3(ntile * batch), dim3(256), S8_LDS, st, reinterpret_cast<const unsigned char *>(x), w_keras, target,
              probs, logits, partials, vv_log2(side), (unsigned)((size_t)batch * side * side * side * F8B_CIN), gamma, epsilon);
    return ntile;
}

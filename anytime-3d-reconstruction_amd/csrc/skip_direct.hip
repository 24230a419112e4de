// Stride-2 convolution / transposed convolution on a 4^3 cell grid with whole samples resident in LDS and the taps that
// fall into the SAME padding skipped at MFMA granularity (bf16, gfx950).
//
//   conv  : Conv3D          k4 s2 SAME,  8^3 x Cin -> 4^3 x Cout   (autoencoder3D.py:26-39;  32^3 model: 128 -> 256)
//   convT : Conv3DTranspose k4 s2 SAME,  4^3 x Cin -> 8^3 x Cout   (autoencoder3D.py:41-54;  32^3 model: 256 -> 128)
//
// Both have 64 "cells" per sample (conv: the 4^3 outputs, convT: the 4^3 inputs) and a batch of samples.  The implicit
// GEMM runs them position-major (rows = samples at one position) so that padded taps can be dropped, but then no input row
// is ever reused inside a workgroup: every (row, tap) is a fresh 128-byte segment through L2 -> LDS, and the kernel sits on
// its per-chunk DMA latency (rocprofv3: 20-23 % MFMA busy, 44-54 % of wave cycles parked on s_waitcnt / s_barrier).
// Here a workgroup owns 4 whole samples:
//
//   rows      : MFMA row tile = 16 rows = (4 w positions) x (4 samples) at ONE (d, h) -> a tap is valid or padding for the
//               whole tile in d and h, so v_mfma_f32_16x16x32_bf16 tiles are skipped exactly where the d / h index leaves the
//               grid (23 % of the dense work); along w the out-of-grid lanes read a zero row instead.
//   A tile    : conv: the phase sub-grid X_q[j] = x[2j + 1 - q] (j = 0..3 per axis, all real data: 8 phases x Cin/64 chunks);
//               convT: the input cells themselves (Cin/64 chunks).  [4 samples][4][4][4] rows of 128 B = 32 KiB, double
//               buffered, staged once by LDS-DMA: every input byte crosses L2 -> LDS once per workgroup, not once per tap.
//   weights   : [tap][Cin/64][Cout][64] (conv) / [parity][tap][Cin/64][Cout][64] (convT) panels, a 32 KiB stage per unit
//               (conv: 4 taps x 64 channels; convT: 1 tap x 2 parities x 128 channels) in a 2-deep ring, LDS-DMA, fully
//               coalesced 8 / 16 KiB pieces.
//   waves     : 8 = 4 row groups x 2; a row group holds the 4 row tiles (d, (d + g) & 3), d = 0..3, so every wave skips the
//               same share of tiles.  conv: the two waves of a row group split K (k-step 0 / 1 of each 64-channel chunk, 64
//               channels, summed through LDS at the end); convT: they take the two w parities (128 channels each).
//   sync      : one barrier per unit (= 2048 matrix-pipe cycles per SIMD): vmcnt(0) + barrier publishes the stage that flew
//               during the previous unit; fragment reads are inline asm one step ahead of their MFMAs with counted lgkmcnt.
//   workgroups: (batch / 4) x (Cout / 64) for conv, (batch / 4) x 4 parity pairs x (Cout / 128) for convT; ordered so that
//               the groups of one sample quad run on one XCD.
#include <stdlib.h>

#include <type_traits>

#include "common.h"

namespace {

typedef __attribute__((ext_vector_type(4))) unsigned sd_u4;

constexpr int SD_ATILE = 256 * 128;        // [4 s][4][4][4] rows x 128 B
constexpr int SD_WST = 32768;              // one weight stage
constexpr int SD_A0 = 4096;                // slack in front of the A buffers: a skipped tile's fragment read may reach 2.5 KiB below its tile
constexpr int SD_W0 = SD_A0 + 2 * SD_ATILE;   // weight ring behind the two A buffers
constexpr int SD_ZERO = SD_W0 + 2 * SD_WST;   // 4 KiB of zeros: what every out-of-grid w lane reads (base + the tap's immediate offset)
constexpr int SD_LDS = SD_ZERO + 4096;     // 139,264 B
constexpr int SD_XCH = 0;                  // conv epilogue: K-half exchange [8 waves][8][64 lanes] f32x4 = 64 KiB
constexpr int SD_OST = 65536;              // conv epilogue: [256 rows][144 B] output staging
constexpr int SD_OPITCH = 144;
constexpr int SD_TPITCH = 272;             // convT epilogue: per-wave [16 rows][256 B + 16]
constexpr int SD_TWAVE = 16 * SD_TPITCH;

struct SdArgs {
    const void *x;
    const void *w;
    const float *scale;
    const float *shift;
    void *y;
    int batch, cin, cout, act;
    unsigned x_bytes, w_bytes;
    int groups;                            // workgroups per sample quad
};

// out[((t*NC + c)*cout + n)*64 + k] = w[(t*cin + c*64 + k)*cout + n]          (Keras Conv3D [kd,kh,kw,Cin,Cout])
__global__ void sd_pack_conv_kernel(const float *__restrict__ w, __bf16 *__restrict__ out, int cin, int cout) {
    const int NC = cin / 64;
    const long total = (long)64 * cin * cout;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int k = (int)(i & 63);
        long r = i >> 6;
        const int n = (int)(r % cout); r /= cout;
        const int c = (int)(r % NC);
        const int t = (int)(r / NC);
        out[i] = static_cast<__bf16>(w[((size_t)t * cin + c * 64 + k) * cout + n]);
    }
}

// The same image through a 64 x 64 LDS tile (cout % 64 == 0): block (t * NC + c, n block) reads 64 k-rows x 64 n with 16-byte loads
// along n and writes 64 n-rows of 64 consecutive k (128 B) -- both sides coalesced.  The element-wise form above reads with a
// stride of cout floats (8x the bytes past L2) and took 30 us on the 8 M-element layer; the training step packs per use.
__device__ __forceinline__ void sd_pack_conv_tiled_body(const float *__restrict__ w, __bf16 *__restrict__ out, int cin, int cout, int kb, int nb) {
    __shared__ float tile[64][65];
    const int tid = threadIdx.x;
    const int n0 = nb * 64;                                      // kb = t * NC + c: Keras rows kb * 64 .. + 63 (row = t * cin + c * 64 + k)
    const int c4 = tid & 15, r = tid >> 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int kl = r + 16 * j;
        const f32x4 v = *reinterpret_cast<const f32x4 *>(w + ((size_t)kb * 64 + kl) * cout + n0 + 4 * c4);
#pragma unroll
        for (int e = 0; e < 4; ++e) tile[kl][4 * c4 + e] = v[e];
    }
    __syncthreads();
    const int col = tid >> 2, kq = (tid & 3) * 16;
    __bf16 *dst = out + ((size_t)kb * cout + n0 + col) * 64 + kq;
#pragma unroll
    for (int e = 0; e < 16; e += 8) {
        bf16x8 o;
#pragma unroll
        for (int u = 0; u < 8; ++u) o[u] = static_cast<__bf16>(tile[kq + e + u][col]);
        *reinterpret_cast<bf16x8 *>(dst + e) = o;
    }
}

__global__ __launch_bounds__(256) void sd_pack_conv_tiled_kernel(const float *__restrict__ w, __bf16 *__restrict__ out, int cin, int cout) {
    sd_pack_conv_tiled_body(w, out, cin, cout, blockIdx.x, blockIdx.y);
}

// Several weight tensors in ONE launch (the training step packs nine such images per step, each a 5-8 us launch of its own): the job table
// travels in the kernel arguments, a block finds its job by the running block count.
constexpr int SD_MAXJOBS = 8;
struct SdPackJobs {
    const float *w[SD_MAXJOBS];
    __bf16 *out[SD_MAXJOBS];
    int cin[SD_MAXJOBS], cout[SD_MAXJOBS];
    int first[SD_MAXJOBS + 1];             // first block of job j; first[njobs] = grid size
    int njobs;
};

__global__ __launch_bounds__(256) void sd_pack_conv_tiled_multi_kernel(const SdPackJobs t) {
    int j = 0;
    while (j + 1 < t.njobs && (int)blockIdx.x >= t.first[j + 1]) ++j;
    const int lb = (int)blockIdx.x - t.first[j];
    const int nkb = 64 * (t.cin[j] / 64);
    sd_pack_conv_tiled_body(t.w[j], t.out[j], t.cin[j], t.cout[j], lb % nkb, lb / nkb);
}

// out[((((p*8 + a)*NC + c)*cout + n)*64 + k] = w[(t(p,a)*cout + n)*cin + c*64 + k],  t = 1 - p + 2a per axis
// (Keras Conv3DTranspose [kd,kh,kw,Cout,Cin]); 8 consecutive k per thread: two 16-byte reads, one 16-byte write
__device__ __forceinline__ void sd_pack_convT_body(const float *__restrict__ w, __bf16 *__restrict__ out, int cin, int cout, long bid, long nblk) {
    const int NC = cin / 64;
    const long total8 = (long)8 * cin * cout;
    for (long i8 = bid * blockDim.x + threadIdx.x; i8 < total8; i8 += nblk * blockDim.x) {
        const int k = (int)(i8 & 7) * 8;
        long r = i8 >> 3;
        const int n = (int)(r % cout); r /= cout;
        const int c = (int)(r % NC); r /= NC;
        const int a = (int)(r & 7), p = (int)(r >> 3);
        const int td = 1 - ((p >> 2) & 1) + 2 * ((a >> 2) & 1);
        const int th = 1 - ((p >> 1) & 1) + 2 * ((a >> 1) & 1);
        const int tw = 1 - (p & 1) + 2 * (a & 1);
        const int t = (td * 4 + th) * 4 + tw;
        const float *src = w + ((size_t)t * cout + n) * cin + c * 64 + k;
        const f32x4 v0 = *reinterpret_cast<const f32x4 *>(src), v1 = *reinterpret_cast<const f32x4 *>(src + 4);
        bf16x8 o;
#pragma unroll
        for (int u = 0; u < 4; ++u) { o[u] = static_cast<__bf16>(v0[u]); o[4 + u] = static_cast<__bf16>(v1[u]); }
        *reinterpret_cast<bf16x8 *>(out + i8 * 8) = o;
    }
}

__global__ void sd_pack_convT_kernel(const float *__restrict__ w, __bf16 *__restrict__ out, int cin, int cout) {
    sd_pack_convT_body(w, out, cin, cout, blockIdx.x, gridDim.x);
}

__global__ void sd_pack_convT_multi_kernel(const SdPackJobs t) {
    int j = 0;
    while (j + 1 < t.njobs && (int)blockIdx.x >= t.first[j + 1]) ++j;
    sd_pack_convT_body(t.w[j], t.out[j], t.cin[j], t.cout[j], (int)blockIdx.x - t.first[j], t.first[j + 1] - t.first[j]);
}

#define SD_RD(DST, ADDR, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(DST) : "v"(ADDR), "n"(OFF) : "memory")
// LDS returns in order: lgkmcnt(N) = everything but the N youngest reads has landed.  Naming the fragments as "+v" ties the
// MFMAs that consume them below this statement.
#define SD_WAIT8(N, F, G)                                                                                                   \
    asm volatile("s_waitcnt lgkmcnt(%8)"                                                                                    \
                 : "+v"(F[0]), "+v"(F[1]), "+v"(F[2]), "+v"(F[3]), "+v"(G[0]), "+v"(G[1]), "+v"(G[2]), "+v"(G[3])          \
                 : "n"(N)                                                                                                   \
                 : "memory")

// MODE 0 = conv (8^3 -> 4^3), MODE 1 = convT (4^3 -> 8^3)
template <int MODE>
__global__ __launch_bounds__(512, 1) void sd_kernel(const SdArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int mq = wave & 3, hv = wave >> 2;          // row group; conv: K half (k-step), convT: w parity
    const int NC = a.cin >> 6;

    // XCD-aware order: the `groups` workgroups of a sample quad, then the next quads, walk one XCD
    const int nwg = gridDim.x;
    const int wi = (nwg & 7) ? (int)blockIdx.x : (int)(blockIdx.x & 7) * (nwg >> 3) + (int)(blockIdx.x >> 3);
    const int sg = wi / a.groups, g = wi - sg * a.groups;
    const int b0 = sg * 4;
    // conv: g = channel group of 64; convT: g = parity pair (pd, ph) + 4 * channel group of 128
    const int ng = MODE == 0 ? g : (g >> 2);
    const int pd = (g >> 1) & 1, ph = g & 1;          // convT only

    const u32x4 rsx = vv_make_rsrc(a.x, a.x_bytes), rsw = vv_make_rsrc(a.w, a.w_bytes);
    const unsigned lds0 = (unsigned)(unsigned long long)(lptr_t)smem;

    for (int i = tid; i < 256; i += 512) reinterpret_cast<uint4 *>(smem + SD_ZERO)[i] = make_uint4(0, 0, 0, 0);
    __syncthreads();

    // ---- producers.  A tile row R = ((s*4 + jd)*4 + jh)*4 + jw, slot swizzle f(R) = ((s & 1) << 2) | (jw & 2): with
    // R & 1 = jw & 1 the 16 lanes of every ds_read_b128 lane group (rows {0-3,12-15} of k-quarter kq, rows {4-11} of kq ^ 1)
    // land on 16 different 16-byte bank positions.  The swizzle is applied to the SOURCE slot; LDS is written linearly.
    // Per piece the lane part of the source offset is prepared once; what changes per tile / unit rides in soffset.
    unsigned a_lane[4], w_lane[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int it = wave * 4 + k;
        const int R = it * 8 + (lane >> 3);
        const int s = R >> 6, jd = (R >> 4) & 3, jh = (R >> 2) & 3, jw = R & 3;
        const int slot = (lane & 7) ^ (((s & 1) << 2) | (jw & 2));
        if (MODE == 0) a_lane[k] = (unsigned)(((((b0 + s) * 8 + 2 * jd) * 8 + 2 * jh) * 8 + 2 * jw) * (a.cin * 2) + slot * 16);
        else a_lane[k] = (unsigned)((((b0 + s) * 64) + (R & 63)) * (a.cin * 2) + slot * 16);
        // weight rows n of a stage: slot swizzle (n >> 1) & 7 (the implicit GEMM's, conflict-free for consecutive rows)
        const int blk = MODE == 0 ? (it & 7) : (it & 15);
        const int nrow = blk * 8 + (lane >> 3);
        w_lane[k] = (unsigned)(nrow * 128 + (((lane & 7) ^ ((nrow >> 1) & 7)) << 4));
    }
    // conv tiles are numbered T = q * NC + c (phase, 64-channel chunk); the callers carry (q, c) along so that no integer
    // division sits in the loop (tq = q, tc = c; unused for convT, whose tile IS the chunk)
    auto issue_a = [&](int T, int tq, int tc) {   // the 4 pieces of this wave for tile T into buffer T & 1
        unsigned soff;
        if (MODE == 0) {
            const int q = tq, c = tc;
            soff = (unsigned)(((1 - ((q >> 2) & 1)) * 64 + (1 - ((q >> 1) & 1)) * 8 + (1 - (q & 1))) * (a.cin * 2) + c * 128);
        } else {
            soff = (unsigned)(T * 128);
        }
        soff = __builtin_amdgcn_readfirstlane(soff);       // wave-uniform by construction (the division by NC goes through the vector unit)
        const unsigned dst = lds0 + SD_A0 + (T & 1) * SD_ATILE + wave * 4096;
#pragma unroll
        for (int k = 0; k < 4; ++k) vv_dma16(rsx, a_lane[k], soff, dst + k * 1024);
    };
    auto issue_w = [&](int u, int tq, int tc) {   // the 4 pieces of this wave for unit u into ring[u & 1]
        unsigned soff;
        if (MODE == 0) {
            const int ad = u & 1;
            const int q = tq, c = tc;
            const int j = wave >> 1;                                       // this wave's pieces belong to tap j of the unit
            const int td = 2 * ad + ((q >> 2) & 1), th = 2 * (j >> 1) + ((q >> 1) & 1), tw = 2 * (j & 1) + (q & 1);
            const int t = (td * 4 + th) * 4 + tw;
            soff = (unsigned)(((t * NC + c) * a.cout + ng * 64) * 128);
        } else {
            const int c = u >> 3, ta = u & 7;
            const int p = pd * 4 + ph * 2 + hv;                            // this wave's pieces belong to its own parity
            soff = (unsigned)((((p * 8 + ta) * NC + c) * a.cout + ng * 128) * 128);
        }
        soff = __builtin_amdgcn_readfirstlane(soff);
        const unsigned dst = lds0 + SD_W0 + (u & 1) * SD_WST + wave * 4096;
#pragma unroll
        for (int k = 0; k < 4; ++k) vv_dma16(rsw, w_lane[k], soff, dst + k * 1024);
    };

    // ---- consumer addressing.  A fragment read = per-lane base (tile, w variant, k-step) + an immediate (the tap's d / h
    // offset): no address arithmetic in the loop.
    const int r = lane & 15, kq = lane >> 4, ls = r >> 2, lw = r & 3;
    auto arow = [&](int jw, int ks) -> unsigned {       // (s, jd = 0, jh = 0, jw), k-step ks: byte offset inside an A tile
        const int f = ((ls & 1) << 2) | (jw & 2);
        return (unsigned)((ls * 64 + jw) * 128 + ((((ks << 2) | kq) ^ f) << 4));
    };
    // weight fragment: row n = lane & 15 of a 16-channel tile, k-quarter kq; conv waves read one k-step (their K half)
    unsigned wb[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) wb[ks] = lds0 + SD_W0 + (unsigned)(r * 128 + ((((ks << 2) | kq) ^ ((r >> 1) & 7)) << 4));
    const unsigned wbc = hv ? wb[1] : wb[0];                 // conv
    const unsigned wbt0 = wb[0] + (unsigned)(hv * 16384), wbt1 = wb[1] + (unsigned)(hv * 16384);   // convT: this wave's parity

    // row tiles of this wave: (d, h) = (i, (i + mq) & 3)
    int td_[4], th_[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { td_[i] = i; th_[i] = (i + mq) & 3; }

    constexpr int NT = MODE == 0 ? 4 : 8;               // 16-channel tiles per wave
    f32x4 acc[4][NT];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto mma4 = [&](const sd_u4 *xf, const sd_u4 *wf, int nt0, int vmask, auto i_c) {
        constexpr int i = decltype(i_c)::value;
        if (vmask & (1 << i)) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][nt0 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8 *>(&wf[j]),
                                                                           *reinterpret_cast<const bf16x8 *>(&xf[i]), acc[i][nt0 + j], 0, 0, 0);
        }
    };
    // A block of 16 MFMAs with the fragment reads of a later block between its cell tiles (and R0 ahead of nothing: the first tile's
    // MFMAs go first): the memory instructions of a block sit in the gaps of the previous one instead of in a burst between two
    // blocks, where both waves of a SIMD would issue them in phase with the matrix pipe idle (convt_whole.hip, in-kernel stamps).
#define SD_MMA_RD(XF, WF, NT0, VM, R1, R2)                                                                                  \
    do {                                                                                                                    \
        mma4(XF, WF, NT0, VM, std::integral_constant<int, 0>{});                                                            \
        R1;                                                                                                                 \
        mma4(XF, WF, NT0, VM, std::integral_constant<int, 1>{});                                                            \
        R2;                                                                                                                 \
        mma4(XF, WF, NT0, VM, std::integral_constant<int, 2>{});                                                            \
        mma4(XF, WF, NT0, VM, std::integral_constant<int, 3>{});                                                            \
    } while (0)
    // tiles whose cell index i + off stays inside 0..3, as a 4-bit mask over the wave's tiles
    auto tile_mask = [&](const int *cell, int off) -> int {
        int m = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) m |= ((unsigned)(cell[i] + off) < 4u) ? (1 << i) : 0;
        return __builtin_amdgcn_readfirstlane(m);
    };

    const int ntiles = MODE == 0 ? 8 * NC : NC;         // A tiles
    const int nunits = ntiles * (MODE == 0 ? 2 : 8);

    // Pipeline.  A unit's barrier sits in front of its LAST MFMA block: by then every fragment of the unit has returned (its
    // weight stage and, at the end of a tile, the A buffer are free for the DMA two units / tiles ahead), the stage of the next
    // unit has landed, and the first fragment reads of the next unit are issued before that last block runs -- the matrix
    // pipe never drains at a barrier.
    // (q, c) of conv tile 1: chunk 1 of phase 0, or phase 1 when a phase has one chunk
    const int q1_ = NC > 1 ? 0 : 1, c1_ = NC > 1 ? 1 : 0;
    issue_a(0, 0, 0);
    issue_w(0, 0, 0);
    if (ntiles > 1) issue_a(1, q1_, c1_); else issue_a(0, 0, 0);   // (a single-tile layer re-stages tile 0: the count below stays fixed)
    issue_w(1, 0, 0);                                       // unit 1: conv (tile 0, ad 1) / convT (chunk 0, tap 1)
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");        // A(0), W(0) landed; A(1), W(1) (4 pieces each, younger) may fly on
    __builtin_amdgcn_s_barrier();

    // zero target of an out-of-grid w lane: the bank position of the (wrapped) row it would have read -> conflict-free
    auto zrow = [&](int jw, int ks) -> unsigned { return lds0 + SD_ZERO + (arow(jw & 3, ks) & 255u); };

    sd_u4 xP[4], xQ[4], wP[4], wQ[4];
    if constexpr (MODE == 0) {
        // the three w variants of a lane: w - 1, w, w + 1 (k-step hv of the chunk)
        const unsigned lm1 = arow((lw - 1) & 3, hv), l00 = arow(lw, hv), lp1 = arow((lw + 1) & 3, hv);
        const unsigned zm1 = zrow(lw - 1, hv), zp1 = zrow(lw + 1, hv);
        const bool okm1 = lw >= 1, okp1 = lw <= 2;
        unsigned xb0[4], xb1[4];                    // aw = 0 / 1 bases of the CURRENT tile
        int dm0, dm1, hm0, hm1;
        // tap (ad, ah, aw) of phase q reads cell (d + ad - 1 + qd, h + ah - 1 + qh, w + aw - 1 + qw)
        auto setup = [&](int T, int q) {
            const int qd = (q >> 2) & 1, qh = (q >> 1) & 1, qw = q & 1;
            const unsigned abuf = lds0 + SD_A0 + (T & 1) * SD_ATILE;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const unsigned sc_ = abuf + (unsigned)((td_[i] - 1 + qd) * 2048 + (th_[i] - 1 + qh) * 512);
                xb0[i] = qw ? l00 + sc_ : (okm1 ? lm1 + sc_ : zm1);
                xb1[i] = qw ? (okp1 ? lp1 + sc_ : zp1) : l00 + sc_;
            }
            dm0 = tile_mask(td_, qd - 1); dm1 = tile_mask(td_, qd); hm0 = tile_mask(th_, qh - 1); hm1 = tile_mask(th_, qh);
        };
        // step j = (ah, aw) of unit AD: x fragments at immediate AD*2048 + ah*512, weight tiles at AD*32768 (ring stage) +
        // j*8192 (tap) + nt*2048
#define SD_CONV_RD(AD, J, XF, WF)                                                                                           \
    do {                                                                                                                    \
        constexpr int XO = (AD) * 2048 + ((J) >> 1) * 512, WO = (AD) * SD_WST + (J) * 8192;                                 \
        if ((J) & 1) { SD_RD(XF[0], xb1[0], XO); SD_RD(XF[1], xb1[1], XO); SD_RD(XF[2], xb1[2], XO); SD_RD(XF[3], xb1[3], XO); } \
        else { SD_RD(XF[0], xb0[0], XO); SD_RD(XF[1], xb0[1], XO); SD_RD(XF[2], xb0[2], XO); SD_RD(XF[3], xb0[3], XO); }     \
        SD_RD(WF[0], wbc, WO); SD_RD(WF[1], wbc, WO + 2048); SD_RD(WF[2], wbc, WO + 4096); SD_RD(WF[3], wbc, WO + 6144);    \
    } while (0)
#define SD_CONV_RDX(AD, J, XF)                                                                                              \
    do {                                                                                                                    \
        constexpr int XO = (AD) * 2048 + ((J) >> 1) * 512;                                                                  \
        if ((J) & 1) { SD_RD(XF[0], xb1[0], XO); SD_RD(XF[1], xb1[1], XO); SD_RD(XF[2], xb1[2], XO); SD_RD(XF[3], xb1[3], XO); } \
        else { SD_RD(XF[0], xb0[0], XO); SD_RD(XF[1], xb0[1], XO); SD_RD(XF[2], xb0[2], XO); SD_RD(XF[3], xb0[3], XO); }     \
    } while (0)
#define SD_CONV_RDW(AD, J, WF)                                                                                              \
    do {                                                                                                                    \
        constexpr int WO = (AD) * SD_WST + (J) * 8192;                                                                      \
        SD_RD(WF[0], wbc, WO); SD_RD(WF[1], wbc, WO + 2048); SD_RD(WF[2], wbc, WO + 4096); SD_RD(WF[3], wbc, WO + 6144);    \
    } while (0)
        setup(0, 0);
        SD_CONV_RD(0, 0, xP, wP);
        int qa = 0, ca = 0;                             // (q, c) of tile T, T + 1, T + 2
        int qb = q1_, cb = c1_;
        int qc = cb + 1 < NC ? qb : qb + 1, cc = cb + 1 < NC ? cb + 1 : 0;
#pragma unroll 1
        for (int T = 0; T < ntiles; ++T) {
            // ---- unit AD = 0
            {
                const int u = 2 * T;
                SD_WAIT8(0, xP, wP);
                SD_MMA_RD(xP, wP, 0, dm0 & hm0, SD_CONV_RDX(0, 1, xQ), SD_CONV_RDW(0, 1, wQ));
                SD_WAIT8(0, xQ, wQ);
                SD_MMA_RD(xQ, wQ, 0, dm0 & hm0, SD_CONV_RDX(0, 2, xP), SD_CONV_RDW(0, 2, wP));
                SD_WAIT8(0, xP, wP);
                SD_MMA_RD(xP, wP, 0, dm0 & hm1, SD_CONV_RDX(0, 3, xQ), SD_CONV_RDW(0, 3, wQ));
                SD_WAIT8(0, xQ, wQ);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                SD_MMA_RD(xQ, wQ, 0, dm0 & hm1, if (u + 2 < nunits) issue_w(u + 2, qb, cb); SD_CONV_RDX(1, 0, xP), SD_CONV_RDW(1, 0, wP));
            }
            // ---- unit AD = 1
            {
                const int u = 2 * T + 1;
                SD_WAIT8(0, xP, wP);
                SD_MMA_RD(xP, wP, 0, dm1 & hm0, SD_CONV_RDX(1, 1, xQ), SD_CONV_RDW(1, 1, wQ));
                SD_WAIT8(0, xQ, wQ);
                SD_MMA_RD(xQ, wQ, 0, dm1 & hm0, SD_CONV_RDX(1, 2, xP), SD_CONV_RDW(1, 2, wP));
                SD_WAIT8(0, xP, wP);
                SD_MMA_RD(xP, wP, 0, dm1 & hm1, SD_CONV_RDX(1, 3, xQ), SD_CONV_RDW(1, 3, wQ));
                SD_WAIT8(0, xQ, wQ);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                const int lastmask = dm1 & hm1;
                mma4(xQ, wQ, 0, lastmask, std::integral_constant<int, 0>{});
                if (u + 2 < nunits) issue_w(u + 2, qb, cb);
                if (T + 2 < ntiles) issue_a(T + 2, qc, cc);
                mma4(xQ, wQ, 0, lastmask, std::integral_constant<int, 1>{});
                if (T + 1 < ntiles) {
                    setup(T + 1, qb);
                    SD_CONV_RD(0, 0, xP, wP);
                }
                mma4(xQ, wQ, 0, lastmask, std::integral_constant<int, 2>{});
                mma4(xQ, wQ, 0, lastmask, std::integral_constant<int, 3>{});
            }
            qa = qb; ca = cb; qb = qc; cb = cc;
            qc = cb + 1 < NC ? qb : qb + 1; cc = cb + 1 < NC ? cb + 1 : 0;
            (void)qa; (void)ca;
        }
        // Nothing of the epilogue may be scheduled above this wait: to the compiler an asm output is a ready value, and it placed
        // the epilogue's first address / select instructions -- into registers of the look-ahead fragments -- in front of it
        // (tests/isa_lint.py walks the generated code for exactly that; the `T + 1 < ntiles` guard makes that path infeasible
        // here, which neither the scheduler nor the lint can know).  Naming the fragments as "+v" here instead made them live
        // out of the loop and the compiler COPIED in-flight registers at the loop head (also caught by the lint).
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#undef SD_CONV_RD
#undef SD_CONV_RDX
#undef SD_CONV_RDW
    } else {
        // tap (ad, ah, aw) of parity (pd, ph, hv) reads cell (d + pd - ad, h + ph - ah, w + hv - aw)
        const int jw0 = lw + hv, jw1 = lw + hv - 1;        // aw = 0 / 1
        const unsigned l0[2] = {arow(jw0 & 3, 0), arow(jw0 & 3, 1)}, l1[2] = {arow(jw1 & 3, 0), arow(jw1 & 3, 1)};
        const unsigned z0[2] = {zrow(jw0, 0), zrow(jw0, 1)}, z1[2] = {zrow(jw1, 0), zrow(jw1, 1)};
        const bool ok0 = (unsigned)jw0 < 4u, ok1 = (unsigned)jw1 < 4u;
        int dmA[2] = {tile_mask(td_, pd), tile_mask(td_, pd - 1)};                          // [ad]
        const int hmA[2] = {tile_mask(th_, ph), tile_mask(th_, ph - 1)};                    // [ah]
        unsigned xb[2][2][4];                        // [aw][k-step][tile] of the CURRENT tile; immediate (1 - ad)*2048 + (1 - ah)*512 on top
        auto setup = [&](int T) {
            const unsigned abuf = lds0 + SD_A0 + (T & 1) * SD_ATILE;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const unsigned sc_ = abuf + (unsigned)((td_[i] + pd - 1) * 2048 + (th_[i] + ph - 1) * 512);
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    xb[0][ks][i] = ok0 ? l0[ks] + sc_ : z0[ks];
                    xb[1][ks][i] = ok1 ? l1[ks] + sc_ : z1[ks];
                }
            }
        };
#define SD_T_X(A, KS, XF)                                                                                                   \
    do {                                                                                                                    \
        constexpr int XO = (1 - (((A) >> 2) & 1)) * 2048 + (1 - (((A) >> 1) & 1)) * 512;                                    \
        SD_RD(XF[0], xb[(A) & 1][KS][0], XO); SD_RD(XF[1], xb[(A) & 1][KS][1], XO);                                         \
        SD_RD(XF[2], xb[(A) & 1][KS][2], XO); SD_RD(XF[3], xb[(A) & 1][KS][3], XO);                                         \
    } while (0)
#define SD_T_W(A, WB, HI, WF)                                                                                               \
    do {                                                                                                                    \
        constexpr int WO = ((A) & 1) * SD_WST + (HI) * 8192;                                                                \
        SD_RD(WF[0], WB, WO); SD_RD(WF[1], WB, WO + 2048); SD_RD(WF[2], WB, WO + 4096); SD_RD(WF[3], WB, WO + 6144);       \
    } while (0)
        setup(0);
        SD_T_X(0, 0, xP); SD_T_W(0, wbt0, 0, wP);
#pragma unroll 1
        for (int T = 0; T < ntiles; ++T) {
            // unit = tap A of this wave's parity: blocks (k-step 0: channel tiles 0..3, 4..7), (k-step 1: 0..3, 4..7); entered
            // with xP (k-step 0) and wP (k-step 0, tiles 0..3) in flight
            auto unit = [&](auto a_c) {
                constexpr int A = decltype(a_c)::value;
                constexpr int AD = (A >> 2) & 1, AH = (A >> 1) & 1, AN = (A + 1) & 7;
                const int u = 8 * T + A;
                const int vm = dmA[AD] & hmA[AH];
                SD_WAIT8(0, xP, wP);
                SD_MMA_RD(xP, wP, 0, vm, SD_T_W(A, wbt0, 1, wQ), (void)0);
                SD_WAIT8(0, xP, wQ);
                SD_MMA_RD(xP, wQ, 4, vm, SD_T_X(A, 1, xQ), SD_T_W(A, wbt1, 0, wP));
                SD_WAIT8(0, xQ, wP);
                SD_MMA_RD(xQ, wP, 0, vm, SD_T_W(A, wbt1, 1, wQ), (void)0);
                SD_WAIT8(0, xQ, wQ);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                mma4(xQ, wQ, 4, vm, std::integral_constant<int, 0>{});
                if (u + 2 < nunits) issue_w(u + 2, 0, 0);
                if (A == 7) {
                    if (T + 2 < ntiles) issue_a(T + 2, 0, 0);
                    mma4(xQ, wQ, 4, vm, std::integral_constant<int, 1>{});
                    if (T + 1 < ntiles) {
                        setup(T + 1);
                        SD_T_X(AN, 0, xP); SD_T_W(AN, wbt0, 0, wP);
                    }
                } else {
                    SD_T_X(AN, 0, xP);
                    mma4(xQ, wQ, 4, vm, std::integral_constant<int, 1>{});
                    SD_T_W(AN, wbt0, 0, wP);
                }
                mma4(xQ, wQ, 4, vm, std::integral_constant<int, 2>{});
                mma4(xQ, wQ, 4, vm, std::integral_constant<int, 3>{});
            };
            unit(std::integral_constant<int, 0>{});
            unit(std::integral_constant<int, 1>{});
            unit(std::integral_constant<int, 2>{});
            unit(std::integral_constant<int, 3>{});
            unit(std::integral_constant<int, 4>{});
            unit(std::integral_constant<int, 5>{});
            unit(std::integral_constant<int, 6>{});
            unit(std::integral_constant<int, 7>{});
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);              // as in the conv form: the epilogue stays below the wait
#undef SD_T_X
#undef SD_T_W
    }
    __builtin_amdgcn_s_barrier();                  // every fragment read has returned: the stages are free

    const int kq4 = kq * 4;
    if constexpr (MODE == 0) {
        // ---- the two K halves of a row group exchange the channel tiles they do not finish: wave hv keeps tiles 2hv, 2hv+1
        f32x4 *xch = reinterpret_cast<f32x4 *>(smem + SD_XCH);
        {
            f32x4 *mine = xch + (size_t)wave * 8 * 64 + lane;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) mine[(i * 2 + j) * 64] = hv ? acc[i][j] : acc[i][2 + j];
        }
        __syncthreads();
        f32x4 fin[4][2];
        {
            const f32x4 *theirs = xch + (size_t)(wave ^ 4) * 8 * 64 + lane;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) fin[i][j] = (hv ? acc[i][2 + j] : acc[i][j]) + theirs[(i * 2 + j) * 64];
        }
        // folded BN + activation; lane = output row (sample ls, w lw) of tile (d, h), registers = 4 consecutive channels
        f32x4 sc[2], sh[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = ng * 64 + (2 * hv + j) * 16 + kq4;
            sc[j] = a.scale ? *reinterpret_cast<const f32x4 *>(a.scale + n) : f32x4{1.f, 1.f, 1.f, 1.f};
            sh[j] = a.shift ? *reinterpret_cast<const f32x4 *>(a.shift + n) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
        auto fill = [&](auto act_c) {
            constexpr int ACT = decltype(act_c)::value;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = ls * 64 + td_[i] * 16 + th_[i] * 4 + lw;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    bf16x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float t = fin[i][j][e] * sc[j][e] + sh[j][e];
                        if (ACT == VV_ACT_ELU) { const float em = __expf(fminf(t, 0.f)) - 1.f; t = t > 0.f ? t : em; }
                        else if (ACT == VV_ACT_RELU) t = fmaxf(t, 0.f);
                        else if (ACT == VV_ACT_LRELU) t = t > 0.f ? t : 0.3f * t;
                        o[e] = static_cast<__bf16>(t);
                    }
                    *reinterpret_cast<bf16x4 *>(smem + SD_OST + row * SD_OPITCH + ((2 * hv + j) * 16 + kq4) * 2) = o;
                }
            }
        };
        switch (a.act) {
            case VV_ACT_ELU: fill(std::integral_constant<int, VV_ACT_ELU>{}); break;
            case VV_ACT_RELU: fill(std::integral_constant<int, VV_ACT_RELU>{}); break;
            case VV_ACT_LRELU: fill(std::integral_constant<int, VV_ACT_LRELU>{}); break;
            default: fill(std::integral_constant<int, VV_ACT_NONE>{}); break;
        }
        __syncthreads();
        // 256 rows x 128 B leave as 16-byte pieces: 8 lanes per output row
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int id = tid + 512 * k, row = id >> 3, pc = id & 7;
            const int b = b0 + (row >> 6);
            if (b < a.batch)
                *reinterpret_cast<uint4 *>(reinterpret_cast<char *>(a.y) + ((size_t)b * 64 + (row & 63)) * (a.cout * 2) + ng * 128 + pc * 16) =
                    *reinterpret_cast<const uint4 *>(smem + SD_OST + row * SD_OPITCH + pc * 16);
        }
    } else {
        // ---- convT: lane = cell (sample ls, mw = lw) of tile (md, mh); output voxel (2 md + pd, 2 mh + ph, 2 mw + hv)
        char *mystage = smem + wave * SD_TWAVE;
        f32x4 sc[NT], sh[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int n = ng * 128 + j * 16 + kq4;
            sc[j] = a.scale ? *reinterpret_cast<const f32x4 *>(a.scale + n) : f32x4{1.f, 1.f, 1.f, 1.f};
            sh[j] = a.shift ? *reinterpret_cast<const f32x4 *>(a.shift + n) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
        auto fill = [&](auto act_c) {
            constexpr int ACT = decltype(act_c)::value;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    bf16x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float t = acc[i][j][e] * sc[j][e] + sh[j][e];
                        if (ACT == VV_ACT_ELU) { const float em = __expf(fminf(t, 0.f)) - 1.f; t = t > 0.f ? t : em; }
                        else if (ACT == VV_ACT_RELU) t = fmaxf(t, 0.f);
                        else if (ACT == VV_ACT_LRELU) t = t > 0.f ? t : 0.3f * t;
                        o[e] = static_cast<__bf16>(t);
                    }
                    *reinterpret_cast<bf16x4 *>(mystage + r * SD_TPITCH + (j * 16 + kq4) * 2) = o;
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_wave_barrier();
                const int od = 2 * td_[i] + pd, oh = 2 * th_[i] + ph;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int id = lane + 64 * k, row = id >> 4, pc = id & 15;
                    const int b = b0 + (row >> 2), ow = 2 * (row & 3) + hv;
                    if (b < a.batch)
                        *reinterpret_cast<uint4 *>(reinterpret_cast<char *>(a.y) + ((((size_t)b * 8 + od) * 8 + oh) * 8 + ow) * (a.cout * 2) + ng * 256 + pc * 16) =
                            *reinterpret_cast<const uint4 *>(mystage + row * SD_TPITCH + pc * 16);
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_wave_barrier();
            }
        };
        switch (a.act) {
            case VV_ACT_ELU: fill(std::integral_constant<int, VV_ACT_ELU>{}); break;
            case VV_ACT_RELU: fill(std::integral_constant<int, VV_ACT_RELU>{}); break;
            case VV_ACT_LRELU: fill(std::integral_constant<int, VV_ACT_LRELU>{}); break;
            default: fill(std::integral_constant<int, VV_ACT_NONE>{}); break;
        }
    }
}

inline int sd_grid_1d(long n) {
    long g = (n + 255) / 256;
    return (int)(g < 1 ? 1 : (g > 8192 ? 8192 : g));
}

template <int MODE>
int sd_launch(const void *x, const void *w, const float *scale, const float *shift, void *y, int batch, int cin, int cout, int act,
              hipStream_t st) {
    static const bool attr = [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&sd_kernel<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, SD_LDS);
        return true;
    }();
    (void)attr;
    const size_t in_vox = MODE == 0 ? 512 : 64, out_vox = MODE == 0 ? 64 : 512;
    const size_t sample_in = in_vox * cin * 2, sample_out = out_vox * cout * 2;
    // 32-bit buffer offsets: a launch covers at most 2 GiB of input; larger batches go out as several launches (sample quads
    // are independent)
    int per = (int)((0x7FFFFFFFull / sample_in) & ~3ull);
    if (per < 4) return VV_ERR_SHAPE;
    for (int s0 = 0; s0 < batch; s0 += per) {
        const int nb = batch - s0 < per ? batch - s0 : per;
        SdArgs a;
        a.x = reinterpret_cast<const char *>(x) + (size_t)s0 * sample_in;
        a.y = reinterpret_cast<char *>(y) + (size_t)s0 * sample_out;
        a.w = w; a.scale = scale; a.shift = shift;
        a.batch = nb; a.cin = cin; a.cout = cout; a.act = act;
        a.x_bytes = (unsigned)((size_t)nb * sample_in);
        a.w_bytes = (unsigned)((size_t)64 * cin * cout * 2);
        a.groups = MODE == 0 ? cout / 64 : 4 * (cout / 128);
        const int nsg = (nb + 3) / 4;
        VV_LAUNCH(sd_kernel<MODE>, dim3(nsg * a.groups), dim3(512), SD_LDS, st, a);
        const int rc = vv_launch_status();
        if (rc != VV_OK) return rc;
    }
    return VV_OK;
}

}  // namespace

VV_EXPORT int vv_conv3d_k4s2_skip_supported(int side, int cin, int cout, int dtype) {
    return dtype == VV_BF16 && side == 8 && cin >= 64 && cin % 64 == 0 && cout >= 64 && cout % 64 == 0 && (size_t)64 * cin * cout * 2 < 0xFFFFFFF0ull;
}

VV_EXPORT int vv_convT3d_k4s2_skip_supported(int side, int cin, int cout, int dtype) {
    return dtype == VV_BF16 && side == 4 && cin >= 64 && cin % 64 == 0 && cout >= 128 && cout % 128 == 0 && (size_t)64 * cin * cout * 2 < 0xFFFFFFF0ull;
}

VV_EXPORT int vv_pack_conv_k4_skip(const float *w_keras, void *packed, int cin, int cout, void *stream) {
    if (!w_keras || !packed) return VV_ERR_NULL;
    if (cin <= 0 || cout <= 0 || cin % 64) return VV_ERR_SHAPE;
    if (cout % 64 == 0 && vv_aligned16(w_keras) && vv_aligned16(packed))
        VV_LAUNCH(sd_pack_conv_tiled_kernel, dim3(64 * (cin / 64), cout / 64), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), w_keras,
                  reinterpret_cast<__bf16 *>(packed), cin, cout);
    else
        VV_LAUNCH(sd_pack_conv_kernel, dim3(sd_grid_1d((long)64 * cin * cout)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), w_keras,
                  reinterpret_cast<__bf16 *>(packed), cin, cout);
    return vv_launch_status();
}

VV_EXPORT int vv_pack_convT_k4s2_skip(const float *w_keras, void *packed, int cin, int cout, void *stream) {
    if (!w_keras || !packed) return VV_ERR_NULL;
    if (cin <= 0 || cout <= 0 || cin % 64) return VV_ERR_SHAPE;
    if (!vv_aligned16(w_keras) || !vv_aligned16(packed)) return VV_ERR_ALIGN;
    VV_LAUNCH(sd_pack_convT_kernel, dim3(sd_grid_1d((long)8 * cin * cout)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), w_keras,
              reinterpret_cast<__bf16 *>(packed), cin, cout);
    return vv_launch_status();
}

// kinds[j]: 0 = vv_pack_conv_k4_skip, 1 = vv_pack_convT_k4s2_skip of (w_keras[j], cin[j], cout[j]) into packed[j]; one launch per kind and
// per 8 jobs.  Same images, bit for bit, as the single calls.
VV_EXPORT int vv_pack_skip_images(const int *kinds, const float *const *w_keras, void *const *packed, const int *cin, const int *cout, int njobs,
                                  void *stream) {
    if (!kinds || !w_keras || !packed || !cin || !cout) return VV_ERR_NULL;
    if (njobs <= 0) return VV_ERR_SHAPE;
    for (int j = 0; j < njobs; ++j) {
        if (!w_keras[j] || !packed[j]) return VV_ERR_NULL;
        if ((kinds[j] != 0 && kinds[j] != 1) || cin[j] <= 0 || cout[j] <= 0 || cin[j] % 64 || (kinds[j] == 0 && cout[j] % 64)) return VV_ERR_SHAPE;
        if (!vv_aligned16(w_keras[j]) || !vv_aligned16(packed[j])) return VV_ERR_ALIGN;
    }
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    for (int kind = 0; kind < 2; ++kind) {
        SdPackJobs t;
        t.njobs = 0;
        t.first[0] = 0;
        int rc_all = VV_OK;
        auto flush = [&]() {
            if (t.njobs == 0) return;
            if (kind == 0) VV_LAUNCH(sd_pack_conv_tiled_multi_kernel, dim3(t.first[t.njobs]), dim3(256), 0, st, t);
            else VV_LAUNCH(sd_pack_convT_multi_kernel, dim3(t.first[t.njobs]), dim3(256), 0, st, t);
            const int rc = vv_launch_status();
            if (rc != VV_OK) rc_all = rc;
            t.njobs = 0;
        };
        for (int j = 0; j < njobs; ++j) {
            if (kinds[j] != kind) continue;
            const int n = t.njobs;
            t.w[n] = w_keras[j]; t.out[n] = reinterpret_cast<__bf16 *>(packed[j]); t.cin[n] = cin[j]; t.cout[n] = cout[j];
            const int blocks = kind == 0 ? 64 * (cin[j] / 64) * (cout[j] / 64) : sd_grid_1d((long)8 * cin[j] * cout[j]);
            t.first[n + 1] = t.first[n] + blocks;
            t.njobs = n + 1;
            if (t.njobs == SD_MAXJOBS) flush();
        }
        flush();
        if (rc_all != VV_OK) return rc_all;
    }
    return VV_OK;
}

VV_EXPORT int vv_conv3d_k4s2_skip_fwd(const void *x, const void *w_skip, const float *scale, const float *shift, void *y, int batch,
                                      int side, int cin, int cout, int act, int dtype, void *stream) {
    if (!x || !w_skip || !y) return VV_ERR_NULL;
    if (!vv_conv3d_k4s2_skip_supported(side, cin, cout, dtype) || batch <= 0) return VV_ERR_SHAPE;
    if (!vv_aligned16(x) || !vv_aligned16(w_skip) || !vv_aligned16(y) || (scale && !vv_aligned16(scale)) || (shift && !vv_aligned16(shift)))
        return VV_ERR_ALIGN;
    return sd_launch<0>(x, w_skip, scale, shift, y, batch, cin, cout, act, reinterpret_cast<hipStream_t>(stream));
}

VV_EXPORT int vv_convT3d_k4s2_skip_fwd(const void *x, const void *w_skip, const float *scale, const float *shift, void *y, int batch,
                                       int side, int cin, int cout, int act, int dtype, void *stream) {
    if (!x || !w_skip || !y) return VV_ERR_NULL;
    if (!vv_convT3d_k4s2_skip_supported(side, cin, cout, dtype) || batch <= 0) return VV_ERR_SHAPE;
    if (!vv_aligned16(x) || !vv_aligned16(w_skip) || !vv_aligned16(y) || (scale && !vv_aligned16(scale)) || (shift && !vv_aligned16(shift)))
        return VV_ERR_ALIGN;
    return sd_launch<1>(x, w_skip, scale, shift, y, batch, cin, cout, act, reinterpret_cast<hipStream_t>(stream));
}

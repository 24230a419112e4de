// Implicit-GEMM convolution kernels on MFMA for gfx950 (CDNA4).
//
//   C[M,N] = gather(A)[M,K] * W[N,K]^T ;  y = act(C*scale[n] + shift[n])
//
// MODE_DENSE : A row m is K contiguous elements               (linearTransform, pooled E5, dense D1)
// MODE_CONV  : Conv3D k4 s2 SAME, rows = output voxels, k = (tap 4x4x4, ci)      -- autoencoder3D.py:26-39
// MODE_CONVT : Conv3DTranspose k4 s2 SAME as 8 output-parity sub-convolutions, rows = INPUT-grid cells,
//              k = (tap 2x2x2, ci), blockIdx.z = parity                          -- autoencoder3D.py:41-54
//
// MODE_FIRST: Conv3D k4 s2 SAME with Cin = 1 on the float32 occupancy grid: K = the 64 taps of the 4x4x4 window,
//              gathered element-wise from x (two 4-voxel runs per 16-byte slot) and converted on the fly
//
// Channels-last makes every K chunk of one row a contiguous 128-byte segment (or zeros, for a tap in the SAME
// padding), so both operands are staged as [rows][128 B] LDS images by LDS-DMA (buffer_load_dwordx4 ... lds issued from
// inline asm; out-of-range descriptors offsets deposit zeros), XOR-swizzled per 16-byte slot on the SOURCE side so the
// ds_read_b128 fragment reads are bank-conflict free, and consumed by v_mfma_f32_32x32x16_bf16 (bf16) or
// v_mfma_f32_32x32x2_f32 (exact-f32 parity mode).  One 256-thread workgroup (4 waves, 2x2) owns a 128 x BN tile; the
// DMA of chunk k+1 flies while chunk k is multiplied (2-stage LDS ring; a deeper ring with counted vmcnt is available
// through the STAGES parameter but measured slower than 2 stages x 2 workgroups per CU).  Small grids run position-major
// with per-tile tap lists (padded taps skipped); work is ordered XCD-aware.  Split-K writes f32 slabs that
// igemm_splitk_epilogue sums in a fixed order (deterministic).
#include <stdlib.h>

#include <type_traits>

#include "common.h"

int vv_first_conv_bf16_launch(const float *x, const void *w_packed, const float *scale, const float *shift, void *y, int batch,
                              int side, int act, void *stream, int out_fp8);   // first_last.hip

namespace {

enum { MODE_DENSE = 0, MODE_CONV = 1, MODE_CONVT = 2, MODE_FIRST = 3 };

struct IgemmArgs {
    const void *A;
    const void *W;
    void *Out;
    float *partial;  // split-K slabs [split][parity][M][N], or nullptr
    const float *scale;
    const float *shift;
    int M, N, K;       // K per parity panel
    int din_log2;      // log2(input side) for the conv modes
    int cin;           // input channels for the conv modes
    int cpt_log2;      // log2(chunks per tap) = log2(cin / BK)
    int nchunks;       // K / BK
    int chunks_per_split;
    int act;
    int out_kind;      // output element type: 0 bf16, 1 f32, 3 fp8 (e4m3fn)
    int batch;         // samples (conv modes)
    unsigned a_bytes, w_bytes;  // operand sizes for the buffer descriptors (out-of-range lanes read zeros)
    int nsplit, nparity;  // split-K shares and output-parity panels (both folded into the 1-D grid)
    int pos_major;     // conv modes: GEMM row m = position * batch + sample (tiles share a position -> padded taps skipped)
    int pair;          // MODE_CONV with 64-byte voxel rows (fp8, Cin 64): a staged 128-byte row = the taps (tw, tw+1), tw even --
                       // two voxels adjacent in memory; the tap list holds pair indices, validity is per 64-byte half
};

typedef __attribute__((ext_vector_type(8))) int i32x8;

constexpr int ROWB = 128;  // bytes per staged row = one K chunk

__device__ __forceinline__ int lds_off(int row, int chunk) { return row * ROWB + ((chunk ^ ((row >> 1) & 7)) << 4); }

template <typename T> struct Elt;
template <> struct Elt<float> { static constexpr int BK = 32; };
template <> struct Elt<__bf16> { static constexpr int BK = 64; };
template <> struct Elt<vv_fp8> { static constexpr int BK = 128; };

// Per-thread description of one A row it stages.
struct RowCtx {
    int off0;  // element offset of tap (0,0,0), channel 0 (may be "negative" for padded taps: never dereferenced)
    int d0, h0, w0;
    bool ok;
};

// GEMM row -> (sample, position-in-grid).  Sample-major: rows of one sample are contiguous (L1 reuse across taps);
// position-major: rows of one output position are contiguous, so a tile sees ONE position and every tap that falls
// into the SAME padding for it is skipped for the whole tile (58 % of the taps of a 4^3 -> 2^3 layer).
__device__ __forceinline__ void split_row(const IgemmArgs &a, int m, int pos_log2, int &b, int &pos) {
    if (a.pos_major) { pos = m / a.batch; b = m - pos * a.batch; }
    else { b = m >> pos_log2; pos = m & ((1 << pos_log2) - 1); }
}

template <int MODE>
__device__ __forceinline__ RowCtx make_row(const IgemmArgs &a, int m, int parity) {
    RowCtx r;
    r.ok = m < a.M;
    if (MODE == MODE_DENSE) {
        r.off0 = m * a.K;
        r.d0 = r.h0 = r.w0 = 0;
        return r;
    }
    const int li = a.din_log2, n = 1 << li;
    int b;
    int pos;
    if (MODE == MODE_CONV || MODE == MODE_FIRST) {
        const int lo = li - 1, msk = (1 << lo) - 1;
        split_row(a, m, 3 * lo, b, pos);
        const int ow = pos & msk, oh = (pos >> lo) & msk, od = (pos >> (2 * lo)) & msk;
        r.d0 = 2 * od - 1; r.h0 = 2 * oh - 1; r.w0 = 2 * ow - 1;  // SAME, k4 s2: pad_before = 1
    } else {
        const int msk = n - 1;
        split_row(a, m, 3 * li, b, pos);
        const int mw = pos & msk, mh = (pos >> li) & msk, md = (pos >> (2 * li)) & msk;
        r.d0 = md + ((parity >> 2) & 1); r.h0 = mh + ((parity >> 1) & 1); r.w0 = mw + (parity & 1);
    }
    r.off0 = (((b * n + r.d0) * n + r.h0) * n + r.w0) * a.cin;  // may be out of range for padded taps: never dereferenced
    return r;
}

// element offset of (row, chunk kc) and its validity
template <int MODE, int BK>
__device__ __forceinline__ bool row_chunk(const IgemmArgs &a, const RowCtx &r, int kc, int chunk, int &off) {
    if (MODE == MODE_DENSE || MODE == MODE_FIRST) {  // K tail (MODE_FIRST gathers in gload itself): 16-byte slots past K read as zero (K % slot == 0)
        off = r.off0 + kc * BK;
        return r.ok && kc * BK + chunk * (BK / 8) < a.K;
    }
    const int li = a.din_log2, n = 1 << li;
    const int tap = kc >> a.cpt_log2, ci0 = (kc & ((1 << a.cpt_log2) - 1)) * BK;
    if (MODE == MODE_CONV) {
        const int td = tap >> 4, th = (tap >> 2) & 3, tw = tap & 3;
        off = r.off0 + ((((td << li) + th) << li) + tw) * a.cin + ci0;
        return r.ok && (unsigned)(r.d0 + td) < (unsigned)n && (unsigned)(r.h0 + th) < (unsigned)n &&
               (unsigned)(r.w0 + tw) < (unsigned)n;
    } else {
        const int ad = tap >> 2, ah = (tap >> 1) & 1, aw = tap & 1;
        off = r.off0 - ((((ad << li) + ah) << li) + aw) * a.cin + ci0;
        return r.ok && (unsigned)(r.d0 - ad) < (unsigned)n && (unsigned)(r.h0 - ah) < (unsigned)n &&
               (unsigned)(r.w0 - aw) < (unsigned)n;
    }
}

// output element offset of row m (channel 0)
template <int MODE>
__device__ __forceinline__ size_t out_row(const IgemmArgs &a, int m, int parity) {
    if (MODE == MODE_DENSE) return (size_t)m * a.N;
    const int li = a.din_log2;
    int b, pos;
    if (MODE != MODE_CONVT) {
        const int lo3 = 3 * (li - 1);
        split_row(a, m, lo3, b, pos);
        return (((size_t)b << lo3) + pos) * a.N;
    }
    const int msk = (1 << li) - 1, lo = li + 1;
    split_row(a, m, 3 * li, b, pos);
    const int mw = pos & msk, mh = (pos >> li) & msk, md = (pos >> (2 * li)) & msk;
    const size_t od = 2 * md + ((parity >> 2) & 1), oh = 2 * mh + ((parity >> 1) & 1), ow = 2 * mw + (parity & 1);
    return (((((((size_t)b << lo) + od) << lo) + oh) << lo) + ow) * (size_t)a.N;
}

// bit t set <=> tap t touches real data for this row (conv: t = (td*4+th)*4+tw; transposed: t = (ad*2+ah)*2+aw)
template <int MODE>
__device__ __forceinline__ unsigned long long row_tapmask(const IgemmArgs &a, const RowCtx &r) {
    if (!r.ok) return 0ull;
    const int n = 1 << a.din_log2;
    unsigned long long m = 0ull;
    if (MODE == MODE_CONV) {
        unsigned vw = 0;
        for (int t = 0; t < 4; ++t) vw |= ((unsigned)(r.w0 + t) < (unsigned)n) << t;
        for (int td = 0; td < 4; ++td)
            for (int th = 0; th < 4; ++th)
                if ((unsigned)(r.d0 + td) < (unsigned)n && (unsigned)(r.h0 + th) < (unsigned)n)
                    m |= (unsigned long long)vw << ((td * 4 + th) * 4);
    } else {
        for (int t = 0; t < 8; ++t)
            if ((unsigned)(r.d0 - (t >> 2)) < (unsigned)n && (unsigned)(r.h0 - ((t >> 1) & 1)) < (unsigned)n &&
                (unsigned)(r.w0 - (t & 1)) < (unsigned)n)
                m |= 1ull << t;
    }
    return m;
}

template <typename T>
__device__ __forceinline__ void mma_step(const uint4 &a, const uint4 &b, f32x16 &acc);
template <>
__device__ __forceinline__ void mma_step<__bf16>(const uint4 &a, const uint4 &b, f32x16 &acc) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8 *>(&a),
                                                  *reinterpret_cast<const bf16x8 *>(&b), acc, 0, 0, 0);
}
// fp8, non-scaled form (kept for reference / VV-independent checks; the kernel's compute step uses the K = 64 block-scaled
// MFMA below): the 16-byte fragment is two 8-byte k-groups; v_mfma_f32_32x32x16_fp8_fp8 takes 8 bytes per lane (lane half h =
// k 8h..8h+7 of its 16), so the low and the high words of both operands go through one MFMA each -- the same fixed
// permutation of k on both sides.  Same MFMA count per MAC as bf16 (the non-scaled fp8 MFMA runs at the bf16 rate),
// half the staged bytes.
template <>
__device__ __forceinline__ void mma_step<vv_fp8>(const uint4 &a, const uint4 &b, f32x16 &acc) {
    const long alo = (long)(((unsigned long long)a.y << 32) | a.x), ahi = (long)(((unsigned long long)a.w << 32) | a.z);
    const long blo = (long)(((unsigned long long)b.y << 32) | b.x), bhi = (long)(((unsigned long long)b.w << 32) | b.z);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(alo, blo, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(ahi, bhi, acc, 0, 0, 0);
}
template <>
__device__ __forceinline__ void mma_step<float>(const uint4 &a, const uint4 &b, f32x16 &acc) {
    // lane (r,h) holds k = 4h..4h+3 of this 8-deep step; MFMA q pairs k = q (h=0) with k = 4+q (h=1): a fixed
    // permutation of k applied to both operands.
    const float *af = reinterpret_cast<const float *>(&a);
    const float *bf = reinterpret_cast<const float *>(&b);
#pragma unroll
    for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[q], bf[q], acc, 0, 0, 0);
}

template <typename T, int MODE, int BM, int BN, int STAGES, int KH>
__global__ __launch_bounds__(256 * KH) void igemm_kernel(const IgemmArgs a) {
    // KH = 2: two 4-wave halves of one 512-thread workgroup share the output tile and take alternate K chunks, each
    // through its own staging buffers; at the end the second half hands its accumulators over through LDS.  Same waves
    // per CU as two split-K workgroups, but no slabs in HBM and no separate reduce pass.
    static_assert(KH == 1 || (KH == 2 && STAGES == 2 && MODE != MODE_FIRST), "the two-half form uses the 2-stage LDS-DMA loop");
    constexpr int BK = Elt<T>::BK;
    constexpr int TM = BM / 64, TN = BN / 64;  // 32x32 tiles per wave (waves laid out 2 x 2)
    constexpr int RA = BM / 32, RB = BN / 32;  // rows staged per thread
    constexpr bool DMA = MODE != MODE_FIRST;   // global -> LDS directly (global_load_lds_dwordx4), no VGPR round trip
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int HB = STAGES * (BM + BN) * ROWB;   // staging bytes of one half
    const int half = KH == 2 ? __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 8) : 0;
    char *As = smem + half * HB;                // [STAGES][BM][128]
    char *Bs = As + STAGES * BM * ROWB;         // [STAGES][BN][128]

    const int tid = threadIdx.x & 255, lane = tid & 63;   // role inside the half
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    // XCD-aware work order.  Workgroups are dealt round-robin over the 8 XCDs (private 4 MiB L2 each), so block b
    // takes item (b % 8) * (T / 8) + b / 8 of a list ordered parity -> split -> n tile -> m tile (fastest first):
    // every XCD walks a contiguous stretch of the list, and the 8 parity panels (or the split-K shares) of one tile,
    // then the neighbouring tiles of the same sample, run on ONE XCD back to back and share their input rows in its L2.
    const int ntn = (a.N + BN - 1) / BN;
    const int nwg = gridDim.x;
    const int wi = (nwg & 7) ? (int)blockIdx.x : (int)(blockIdx.x & 7) * (nwg >> 3) + (int)(blockIdx.x >> 3);
    const int parity = wi % a.nparity;
    const int split = (wi / a.nparity) % a.nsplit;
    const int tl = wi / (a.nparity * a.nsplit);
    const int tile_n = tl % ntn, tile_m = tl / ntn;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int kc_begin = split * a.chunks_per_split;
    const int kc_end = min(a.nchunks, kc_begin + a.chunks_per_split);

    // Lane -> (row r0 + 32 i, 16-byte slot `pos`) of the staged images.  The LDS image is written linearly (slot pos
    // of row r at r*128 + pos*16, as LDS-DMA requires); the XOR swizzle is applied on the SOURCE side: the lane
    // fetches global slot gchunk = pos ^ f(row), f(row) = (row >> 1) & 7, and fragment reads look a slot c up at
    // position c ^ f(row).  f is the same for rows r0 + 32 i.
    const int pos = tid & 7, r0 = tid >> 3;
    const int gchunk = pos ^ ((r0 >> 1) & 7);

    RowCtx rows[RA];
#pragma unroll
    for (int i = 0; i < RA; ++i) rows[i] = make_row<MODE>(a, m0 + r0 + 32 * i, parity);

    constexpr bool TAPS = MODE == MODE_CONV || MODE == MODE_CONVT;
    // Everything per-row that the K loop needs is hoisted here: a 64-bit source base (tap 0, channel 0, this lane's
    // slot), a per-tap validity bitmask, and for the weights a base + validity.  The loop then adds one wave-uniform
    // delta per chunk and selects the zero page for padded taps: ~7 VALU per staged row instead of a coordinate decode.
    // LDS-DMA goes through buffer descriptors: a lane whose offset is out of range deposits ZEROS in its LDS slot, which
    // is exactly what a tap in the SAME padding (or a row past M / N / K) must contribute -- no zero page, no branches.
    constexpr unsigned OOB = 0xFFFFFFF0u;
    const u32x4 rsA = vv_make_rsrc(a.A, a.a_bytes), rsW = vv_make_rsrc(a.W, a.w_bytes);
    const unsigned lds0 = (unsigned)(unsigned long long)(lptr_t)smem + half * HB;   // LDS byte address of this half's staging area
    int aoff[RA];                     // byte offset of (row, tap 0, channel 0, this lane's slot); garbage where never valid
    unsigned long long rmask[RA];     // per-tap validity of the row
#pragma unroll
    for (int i = 0; i < RA; ++i) {
        aoff[i] = rows[i].off0 * (int)sizeof(T) + gchunk * 16;
        rmask[i] = TAPS ? row_tapmask<MODE>(a, rows[i]) : (rows[i].ok ? ~0ull : 0ull);
    }
    unsigned woff[RB];                // byte offset of (weight row, k 0, this lane's slot); OOB past N
#pragma unroll
    for (int i = 0; i < RB; ++i)
        woff[i] = (n0 + r0 + 32 * i < a.N) ? (unsigned)(((size_t)parity * a.N + n0 + r0 + 32 * i) * a.K * sizeof(T)) + gchunk * 16 : OOB;

    // Tile-uniform list of taps that touch real data for at least one row (smem tail, lives through the K loop).
    int *taplist = reinterpret_cast<int *>(smem + KH * HB);   // [64] + 2 mask words
    int nvalid = kc_end - kc_begin, vb = kc_begin;                         // dense / first: chunks are the list
    if constexpr (TAPS) {
        unsigned *mw = reinterpret_cast<unsigned *>(taplist + 64);
        if (tid < 2) mw[tid] = 0u;
        __syncthreads();
        unsigned long long mine = 0ull;
#pragma unroll
        for (int i = 0; i < RA; ++i) mine |= rmask[i];
        if ((unsigned)mine) atomicOr(&mw[0], (unsigned)mine);
        if ((unsigned)(mine >> 32)) atomicOr(&mw[1], (unsigned)(mine >> 32));
        __syncthreads();
        unsigned long long tapmask = ((unsigned long long)mw[1] << 32) | mw[0];
        if (MODE == MODE_CONV && a.pair) {
            // pair p = taps (2p, 2p+1): listed if either tap is live; entries are pair indices
            const unsigned long long e = (tapmask | (tapmask >> 1)) & 0x5555555555555555ull;
            unsigned long long packed = 0ull;
            for (int q = 0; q < 32; ++q) packed |= ((e >> (2 * q)) & 1ull) << q;
            tapmask = packed;
        }
        if (tid < 64 && ((tapmask >> tid) & 1ull)) taplist[__builtin_popcountll(tapmask & ((1ull << tid) - 1ull))] = tid;
        __syncthreads();
        const int ntap = __builtin_amdgcn_readfirstlane(__builtin_popcountll(tapmask));
        const int total = ntap << a.cpt_log2;                 // valid chunks of this tile
        const int per = (total + a.nsplit - 1) / a.nsplit;   // split-K shares cut from the VALID list
        vb = split * per;
        nvalid = max(0, min(total, vb + per) - vb);
    }

    // ---- DMA path: one buffer_load_dwordx4 ... lds per staged row per thread; a wave instruction fills 8 rows (1 KiB).
    // Per chunk and row: test one bit, add the wave-uniform tap delta, select OOB.  Weights: the delta rides in soffset.
    auto issue = [&](int vi, int buf, bool live = true) {     // live == false: deposit zeros (a half without a chunk left)
        int tap = 0, kc = vi, delta;
        if constexpr (TAPS) {
            const int sub = vi & ((1 << a.cpt_log2) - 1);
            tap = live ? __builtin_amdgcn_readfirstlane(taplist[vi >> a.cpt_log2]) : 0;
            kc = (tap << a.cpt_log2) + sub;
            const int li = a.din_log2;
            int toff;
            if (MODE == MODE_CONV) {
                if (a.pair) tap *= 2;                      // first tap of the pair; the lane's own tap is tap + (slot >> 2)
                toff = ((((tap >> 4) << li) + ((tap >> 2) & 3)) << li) + (tap & 3);
            }
            else toff = -(((((tap >> 2) << li) + ((tap >> 1) & 1)) << li) + (tap & 1));
            delta = (toff * a.cin + sub * BK) * (int)sizeof(T);
        } else {
            delta = kc * ROWB;
        }
        const bool kin = live && ((MODE != MODE_DENSE) || (kc * BK + gchunk * (BK / 8) < a.K));
#pragma unroll
        for (int i = 0; i < RA; ++i) {
            bool v;
            if constexpr (MODE == MODE_CONV) v = (rmask[i] >> (a.pair ? tap + (gchunk >> 2) : tap)) & 1ull;
            else v = ((unsigned)rmask[i] >> tap) & 1u;
            const unsigned vo = (v && kin) ? (unsigned)(aoff[i] + delta) : OOB;
            vv_dma16(rsA, vo, 0u, lds0 + buf * BM * ROWB + (wave * 8 + 32 * i) * ROWB);
        }
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            const unsigned vo = kin ? woff[i] : OOB;
            vv_dma16(rsW, vo, (unsigned)(kc * ROWB), lds0 + STAGES * BM * ROWB + buf * BN * ROWB + (wave * 8 + 32 * i) * ROWB);
        }
    };

    // ---- register path (MODE_FIRST): slot gchunk of K chunk kc holds 16/sizeof(T) consecutive taps
    // t = (td*4 + th)*4 + tw of the 4x4x4 window, gathered from the float32 grid and converted
    uint4 ra[DMA ? 1 : RA], rb[DMA ? 1 : RB];
    auto gload = [&](int kc) {
        if constexpr (!DMA) {
            constexpr int EPS = 16 / (int)sizeof(T);
            const int n = 1 << a.din_log2;
            const float *xf = reinterpret_cast<const float *>(a.A);
#pragma unroll
            for (int i = 0; i < RA; ++i) {
                T vals[EPS];
                // EPS/4 runs of 4 consecutive taps (tw = 0..3): one (td, th) row check and base address per run; only the
                // two outer voxels of a run can fall off the grid along w (iw = 2 ow - 1 .. 2 ow + 2)
#pragma unroll
                for (int r = 0; r < EPS / 4; ++r) {
                    const int t0 = kc * BK + gchunk * EPS + 4 * r;
                    const int td = t0 >> 4, th = (t0 >> 2) & 3;
                    const bool rowok = rows[i].ok && (unsigned)(rows[i].d0 + td) < (unsigned)n && (unsigned)(rows[i].h0 + th) < (unsigned)n;
                    const float *xr = xf + rows[i].off0 + (((td << a.din_log2) + th) << a.din_log2);
                    const float v0 = (rowok && rows[i].w0 >= 0) ? xr[0] : 0.f;
                    const float v1 = rowok ? xr[1] : 0.f, v2 = rowok ? xr[2] : 0.f;
                    const float v3 = (rowok && rows[i].w0 + 3 < n) ? xr[3] : 0.f;
                    vals[4 * r + 0] = static_cast<T>(v0); vals[4 * r + 1] = static_cast<T>(v1);
                    vals[4 * r + 2] = static_cast<T>(v2); vals[4 * r + 3] = static_cast<T>(v3);
                }
                ra[i] = *reinterpret_cast<const uint4 *>(vals);
            }
#pragma unroll
            for (int i = 0; i < RB; ++i)
                rb[i] = woff[i] != OOB ? *reinterpret_cast<const uint4 *>(reinterpret_cast<const char *>(a.W) + woff[i] + (size_t)kc * ROWB)
                                       : make_uint4(0, 0, 0, 0);
        }
    };
    auto lstore = [&](int buf) {
        if constexpr (!DMA) {
#pragma unroll
            for (int i = 0; i < RA; ++i) *reinterpret_cast<uint4 *>(As + buf * BM * ROWB + (r0 + 32 * i) * ROWB + pos * 16) = ra[i];
#pragma unroll
            for (int i = 0; i < RB; ++i) *reinterpret_cast<uint4 *>(Bs + buf * BN * ROWB + (r0 + 32 * i) * ROWB + pos * 16) = rb[i];
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;

    const int fr = lane & 31, fh = lane >> 5;
    int vi = vb;
    const int ve = vb + nvalid;
    auto compute = [&](int buf) {
        const char *Ac = As + buf * BM * ROWB, *Bc = Bs + buf * BN * ROWB;
        // fragments of k-step ks+1 are read while the MFMAs of k-step ks issue (ping-pong registers): a wave does not
        // sit on its own LDS latency between k-steps
        uint4 fa[2][TM], fb[2][TN];
        auto read_frags = [&](int ks, uint4 *pa, uint4 *pb) {
#pragma unroll
            for (int i = 0; i < TM; ++i) pa[i] = *reinterpret_cast<const uint4 *>(Ac + lds_off(wm * (BM / 2) + i * 32 + fr, ks * 2 + fh));
#pragma unroll
            for (int j = 0; j < TN; ++j) pb[j] = *reinterpret_cast<const uint4 *>(Bc + lds_off(wn * (BN / 2) + j * 32 + fr, ks * 2 + fh));
        };
        if constexpr (sizeof(T) == 1) {
            // fp8: two 16-byte slots (k-steps 2 kp, 2 kp + 1) make the 32-byte operand of the block-scaled K = 64 MFMA,
            // which retires 4x the K of the bf16 instruction in 2x its cycles (unit scales: E8M0 127 in every byte)
            uint4 ga[2][2][TM], gb[2][2][TN];
            read_frags(0, ga[0][0], gb[0][0]);
            read_frags(1, ga[0][1], gb[0][1]);
#pragma unroll
            for (int kp = 0; kp < 2; ++kp) {
                if (kp == 0) { read_frags(2, ga[1][0], gb[1][0]); read_frags(3, ga[1][1], gb[1][1]); }
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        const uint4 &w0 = gb[kp][0][j], &w1 = gb[kp][1][j], &x0 = ga[kp][0][i], &x1 = ga[kp][1][i];
                        const i32x8 wv = {(int)w0.x, (int)w0.y, (int)w0.z, (int)w0.w, (int)w1.x, (int)w1.y, (int)w1.z, (int)w1.w};
                        const i32x8 xv = {(int)x0.x, (int)x0.y, (int)x0.z, (int)x0.w, (int)x1.x, (int)x1.y, (int)x1.z, (int)x1.w};
                        acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(wv, xv, acc[i][j], 0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
                    }
            }
            return;
        }
        read_frags(0, fa[0], fb[0]);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            if (ks + 1 < 4) read_frags(ks + 1, fa[(ks + 1) & 1], fb[(ks + 1) & 1]);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) mma_step<T>(fb[ks & 1][j], fa[ks & 1][i], acc[i][j]);   // D[n][m]: lane = row m, registers walk n
        }
    };
    if constexpr (DMA && STAGES > 2) {
        // STAGES-deep LDS ring, ONE barrier per chunk: chunk i+STAGES-1 is issued into the stage chunk i-1 just left, and
        // the wait before the barrier is a COUNTED vmcnt that leaves the STAGES-2 younger chunks in flight.
        constexpr int LPC = RA + RB;              // LDS-DMA instructions per thread per chunk
        const int nv = ve - vb;
#pragma unroll
        for (int s = 0; s < STAGES - 1; ++s)
            if (s < nv) issue(vb + s, s);
        int st = 0;
        for (int i = 0; i < nv; ++i) {
            const int younger = min(STAGES - 2, nv - 1 - i);
            if (younger >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LPC) : "memory");
            else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPC) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (i + STAGES - 1 < nv) issue(vb + i + STAGES - 1, st == 0 ? STAGES - 1 : st - 1);
            compute(st);
            st = st + 1 == STAGES ? 0 : st + 1;
        }
        __syncthreads();
    } else if constexpr (KH == 2) {
        // both halves run the same number of steps (one barrier each); half h takes chunks vb + 2 i + h
        const int nsteps = (ve - vb + 1) >> 1;
        auto issue_step = [&](int i, int buf) { const int v = vb + 2 * i + half; issue(v < ve ? v : vb, buf, v < ve); };
        if (nsteps > 0) issue_step(0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        int buf = 0;
        for (int i = 0; i < nsteps; ++i) {
            if (i + 1 < nsteps) issue_step(i + 1, buf ^ 1);
            compute(buf);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            buf ^= 1;
        }
        // hand-over: thread t of the second half parks its accumulators where thread t of the first half picks them up
        // (16 bytes per lane per step: conflict-free), over the staging area nobody reads any more
        f32x4 *xch = reinterpret_cast<f32x4 *>(smem);
        if (half == 1) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        xch[((i * TN + j) * 4 + g) * 256 + tid] = f32x4{acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
        }
        __syncthreads();
        if (half == 1) return;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 v = xch[((i * TN + j) * 4 + g) * 256 + tid];
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[i][j][4 * g + e] += v[e];
                }
        __syncthreads();          // (first half only from here on) the epilogue tile reuses the same bytes
    } else {
        if (vi < ve) {
            if constexpr (DMA) issue(vi, 0);
            else { gload(vi); lstore(0); }
        }
        if constexpr (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        int buf = 0;
        for (; vi < ve; ++vi) {
            const bool more = vi + 1 < ve;
            if (more) {
                if constexpr (DMA) issue(vi + 1, buf ^ 1);
                else gload(vi + 1);
            }
            compute(buf);
            if (more) lstore(buf ^ 1);
            if constexpr (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // chunk vi+1 has landed (it flew during the MFMAs)
            __syncthreads();
            buf ^= 1;
        }
    }

    // ---- epilogue.  The MFMAs were issued weights-first, so a lane owns ONE output row m = lane & 31 and its 16
    // registers walk the channels n = (q & 3) + 8 (q >> 2) + 4 (lane >> 5): four consecutive channels per register
    // quad.  Each quad gets the folded BN + activation as a float4, is packed (8 B of bf16 / 16 B of f32) into a
    // [BM][BN] row-major LDS tile, and the tile leaves as 16-byte stores: whole channel rows, coalesced.  The
    // activation / output kind are hoisted into template parameters so the hot code has no per-element branches.
    const int okind = a.partial ? 2 : a.out_kind;
    const int es = okind == 0 ? 2 : (okind == 3 ? 1 : 4);
    const int pitch = BN * es + 16;
    // Folded BN quads of this lane's channels, fetched as one batch (two uniform branches, one wait) instead of a
    // dependent load + wait per quad.
    f32x4 scv[TN][4], shv[TN][4];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) { scv[j][g] = f32x4{1.f, 1.f, 1.f, 1.f}; shv[j][g] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    if (okind != 2 && a.scale) {
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int cl = wn * (BN / 2) + j * 32 + 8 * g + 4 * fh;
                if (n0 + cl < a.N) scv[j][g] = *reinterpret_cast<const f32x4 *>(a.scale + n0 + cl);
            }
    }
    if (okind != 2 && a.shift) {
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int cl = wn * (BN / 2) + j * 32 + 8 * g + 4 * fh;
                if (n0 + cl < a.N) shv[j][g] = *reinterpret_cast<const f32x4 *>(a.shift + n0 + cl);
            }
    }
    auto fill = [&](auto act_c, auto kind_c) {
        constexpr int ACT = decltype(act_c)::value, KIND = decltype(kind_c)::value;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int cl = wn * (BN / 2) + j * 32 + 8 * g + 4 * fh;   // tile-local channel of the quad
                const f32x4 sc = scv[j][g], sh = shv[j][g];
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const int rl = wm * (BM / 2) + i * 32 + fr;
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float t = acc[i][j][4 * g + e];
                        if (KIND != 2) {
                            t = t * sc[e] + sh[e];
                            if (ACT == VV_ACT_ELU) { const float tn = fminf(t, 0.f), em = (KIND == 0 || KIND == 3) ? __expf(tn) - 1.f : expm1f(tn); t = t > 0.f ? t : em; }
                            else if (ACT == VV_ACT_RELU) t = fmaxf(t, 0.f);
                            else if (ACT == VV_ACT_LRELU) t = t > 0.f ? t : 0.3f * t;
                        }
                        v[e] = t;
                    }
                    if (KIND == 0) {
                        bf16x4 o;
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = static_cast<__bf16>(v[e]);
                        *reinterpret_cast<bf16x4 *>(smem + rl * pitch + cl * 2) = o;
                    } else if (KIND == 3) {
                        *reinterpret_cast<unsigned *>(smem + rl * pitch + cl) = vv_pack_fp8x4(v);
                    } else {
                        *reinterpret_cast<f32x4 *>(smem + rl * pitch + cl * 4) = v;
                    }
                }
            }
        }
    };
    auto with_kind = [&](auto act_c) {
        if (okind == 0) fill(act_c, std::integral_constant<int, 0>{});
        else if (okind == 1) fill(act_c, std::integral_constant<int, 1>{});
        else if (okind == 3) fill(act_c, std::integral_constant<int, 3>{});
        else fill(act_c, std::integral_constant<int, 2>{});
    };
    switch (a.partial ? VV_ACT_NONE : a.act) {
        case VV_ACT_ELU: with_kind(std::integral_constant<int, VV_ACT_ELU>{}); break;
        case VV_ACT_RELU: with_kind(std::integral_constant<int, VV_ACT_RELU>{}); break;
        case VV_ACT_LRELU: with_kind(std::integral_constant<int, VV_ACT_LRELU>{}); break;
        default: with_kind(std::integral_constant<int, VV_ACT_NONE>{}); break;
    }
    __syncthreads();
    const int cpr = BN * es / 16;  // 16-byte chunks per tile row
    char *outb = a.partial ? reinterpret_cast<char *>(a.partial) : reinterpret_cast<char *>(a.Out);
    const int epc = 16 / es;
    for (int idx = tid; idx < BM * cpr; idx += 256) {
        const int rl = idx / cpr, c = idx % cpr;
        const int m = m0 + rl, n = n0 + c * epc;
        if (m >= a.M || n >= a.N) continue;
        const size_t row = a.partial ? ((size_t)(split * a.nparity + parity) * a.M + m) * a.N : out_row<MODE>(a, m, parity);
        *reinterpret_cast<uint4 *>(outb + (row + n) * es) = *reinterpret_cast<const uint4 *>(smem + rl * pitch + c * 16);
    }
}

// Sums the split-K slabs in split order, applies the folded BN + activation, scatters to the output: one channel quad
// per thread (float4 slab reads, float4 folded-BN reads, one 8- / 16-byte store), activation hoisted out of the loop.
template <int MODE, int ACT>
__global__ __launch_bounds__(256) void igemm_splitk_epilogue(const IgemmArgs a, int nsplit, int nparity) {
    const int n4 = a.N >> 2;
    const size_t total = (size_t)nparity * a.M * n4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % n4);
        const size_t pm = i / n4;
        const int m = (int)(pm % a.M), parity = (int)(pm / a.M);
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        for (int sp = 0; sp < nsplit; ++sp)
            s += *reinterpret_cast<const f32x4 *>(a.partial + (((size_t)(sp * nparity + parity) * a.M + m) * a.N + c4 * 4));
        f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
        if (a.scale) sc = *reinterpret_cast<const f32x4 *>(a.scale + c4 * 4);
        if (a.shift) sh = *reinterpret_cast<const f32x4 *>(a.shift + c4 * 4);
        const size_t o = out_row<MODE>(a, m, parity) + c4 * 4;
        f32x4 v;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float t = s[q] * sc[q] + sh[q];
            if (ACT == VV_ACT_ELU) { const float tn = fminf(t, 0.f), em = a.out_kind != 1 ? __expf(tn) - 1.f : expm1f(tn); t = t > 0.f ? t : em; }
            else if (ACT == VV_ACT_RELU) t = fmaxf(t, 0.f);
            else if (ACT == VV_ACT_LRELU) t = t > 0.f ? t : 0.3f * t;
            v[q] = t;
        }
        if (a.out_kind == 0) {
            bf16x4 ob;
#pragma unroll
            for (int q = 0; q < 4; ++q) ob[q] = static_cast<__bf16>(v[q]);
            *reinterpret_cast<bf16x4 *>(reinterpret_cast<__bf16 *>(a.Out) + o) = ob;
        } else if (a.out_kind == 3) {
            *reinterpret_cast<unsigned *>(reinterpret_cast<unsigned char *>(a.Out) + o) = vv_pack_fp8x4(v);
        } else {
            *reinterpret_cast<f32x4 *>(reinterpret_cast<float *>(a.Out) + o) = v;
        }
    }
}

struct Plan {
    int bm, bn, split, cps, nparity, kh;
    size_t ws_bytes;
};

// Tile / split-K choice: fill >= 2 workgroups per CU (256 CUs) when the tile grid alone cannot.
Plan make_plan(int mode, int M, int N, int K, int dtype) {
    Plan p;
    p.nparity = mode == MODE_CONVT ? 8 : 1;
    p.bm = 128;
    p.bn = (N % 128 == 0) ? 128 : 64;  // N tail (N % 64 != 0): weight rows past N read as zero, stores masked
    // Dense-shaped layers with a small tile grid: 64-wide tiles double the workgroups (measured: E5 0.021 -> 0.017 ms,
    // D1 0.0105 -> 0.0081; the stride-2 layers lose with 64-wide tiles and keep 128)
    if (mode == MODE_DENSE && p.bn == 128 && (long)((M + p.bm - 1) / p.bm) * (N / 128) <= 64) p.bn = 64;
    const int bk = dtype == VV_BF16 ? 64 : (dtype == VV_FP8 ? 128 : 32);
    const int nchunks = (K + bk - 1) / bk;
    const long tiles = (long)((M + p.bm - 1) / p.bm) * ((N + p.bn - 1) / p.bn) * p.nparity;
    static const long target = vv_hook("VV_SPLIT_TARGET") ? atol(vv_hook("VV_SPLIT_TARGET")) : 512;
    int split = 1;
    // at least 8 chunks per share -- 4 when the tile grid is a handful of workgroups (the Dense-shaped layers at batch
    // 256: 2 tiles x K = 4096 ran as 8 workgroups of 16 dependent chunks, a latency chain of 18 us)
    static const int min_chunks_small = vv_hook("VV_SPLIT_MINCHUNKS") ? atoi(vv_hook("VV_SPLIT_MINCHUNKS")) : 4;
    const int min_chunks = tiles <= 8 ? min_chunks_small : 8;
    while (tiles * split < target && split * 2 <= nchunks / min_chunks && split < 64) split *= 2;
    // The first factor of two is taken INSIDE the workgroup (two 4-wave halves on alternate chunks, accumulators handed
    // over through LDS): same waves per CU, half the slabs -- or none, and then no reduce pass at all.
    static const bool no_kh = vv_hook("VV_NO_KHALVES") != nullptr;
    p.kh = 1;
    if (split >= 2 && mode != MODE_FIRST && !no_kh) { p.kh = 2; split /= 2; }
    p.cps = (nchunks + split - 1) / split;
    p.split = (nchunks + p.cps - 1) / p.cps;
    p.ws_bytes = p.split > 1 ? (size_t)p.split * p.nparity * M * N * sizeof(float) : 0;
    return p;
}

// Position-major rows pay off where a large share of the taps is padding (small grids) and the batch fills tiles.
bool use_pos_major(int mode, int din, int batch) {
    static const int conv_max = vv_hook("VV_POSMAJOR_CONV_SIDE") ? atoi(vv_hook("VV_POSMAJOR_CONV_SIDE")) : 8;
    static const int convT_max = vv_hook("VV_POSMAJOR_CONVT_SIDE") ? atoi(vv_hook("VV_POSMAJOR_CONVT_SIDE")) : 4;
    if (batch < 32) return false;
    if (mode == MODE_CONV) return din <= conv_max;
    if (mode == MODE_CONVT) return din <= convT_max;
    return false;
}

template <typename T, int MODE, int BN, int STAGES, int KH>
void launch_k(const IgemmArgs &a, dim3 grid, hipStream_t st) {
    constexpr size_t stage_bytes = (size_t)KH * STAGES * (128 + BN) * ROWB + 272;   // staging of every half + tap list
    constexpr size_t epi_bytes = (size_t)128 * (BN * 4 + 16);
    constexpr size_t lds = stage_bytes > epi_bytes ? stage_bytes : epi_bytes;
    static const bool attr_set = [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&igemm_kernel<T, MODE, 128, BN, STAGES, KH>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        return true;
    }();
    (void)attr_set;
    VV_LAUNCH((igemm_kernel<T, MODE, 128, BN, STAGES, KH>), grid, dim3(256 * KH), lds, st, a);
}

template <typename T, int MODE>
int launch_t(const IgemmArgs &a, const Plan &p, hipStream_t st) {
    const int tiles = ((a.M + p.bm - 1) / p.bm) * ((a.N + p.bn - 1) / p.bn);
    dim3 grid(tiles * p.split * p.nparity);
    static const int stages_env = vv_hook("VV_STAGES") ? atoi(vv_hook("VV_STAGES")) : 2;
    const int stages = (MODE == MODE_FIRST || sizeof(T) == 4) ? 2 : stages_env;   // deep ring: bf16 LDS-DMA modes only
    if (p.kh == 2) {
        if constexpr (MODE != MODE_FIRST) {
            if (p.bn == 128) launch_k<T, MODE, 128, 2, 2>(a, grid, st);
            else launch_k<T, MODE, 64, 2, 2>(a, grid, st);
        }
    } else if (p.bn == 128) {
        if (stages == 3) launch_k<T, MODE, 128, 3, 1>(a, grid, st);
        else if (stages == 4) launch_k<T, MODE, 128, 4, 1>(a, grid, st);
        else launch_k<T, MODE, 128, 2, 1>(a, grid, st);
    } else {
        if (stages == 3) launch_k<T, MODE, 64, 3, 1>(a, grid, st);
        else if (stages == 4) launch_k<T, MODE, 64, 4, 1>(a, grid, st);
        else launch_k<T, MODE, 64, 2, 1>(a, grid, st);
    }
    if (p.split > 1) {
        const size_t total = (size_t)p.nparity * a.M * (a.N / 4);
        const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
        switch (a.act) {
            case VV_ACT_ELU: VV_LAUNCH((igemm_splitk_epilogue<MODE, VV_ACT_ELU>), dim3(blocks), dim3(256), 0, st, a, p.split, p.nparity); break;
            case VV_ACT_RELU: VV_LAUNCH((igemm_splitk_epilogue<MODE, VV_ACT_RELU>), dim3(blocks), dim3(256), 0, st, a, p.split, p.nparity); break;
            case VV_ACT_LRELU: VV_LAUNCH((igemm_splitk_epilogue<MODE, VV_ACT_LRELU>), dim3(blocks), dim3(256), 0, st, a, p.split, p.nparity); break;
            default: VV_LAUNCH((igemm_splitk_epilogue<MODE, VV_ACT_NONE>), dim3(blocks), dim3(256), 0, st, a, p.split, p.nparity); break;
        }
    }
    return vv_launch_status();
}

int run_igemm(int mode, const void *x, const void *w, const float *scale, const float *shift, void *y, int M, int N,
              int K, int din, int cin, int act, int dtype, int out_dtype, void *ws, size_t ws_bytes, void *stream,
              int batch = 1) {
    if (!x || !w || !y) return VV_ERR_NULL;
    if (dtype != VV_F32 && dtype != VV_BF16 && dtype != VV_FP8) return VV_ERR_DTYPE;
    if (out_dtype != VV_F32 && out_dtype != VV_BF16 && out_dtype != VV_FP8) return VV_ERR_DTYPE;
    if (mode == MODE_FIRST && dtype == VV_FP8) return VV_ERR_DTYPE;
    if (dtype == VV_F32 && out_dtype == VV_FP8) return VV_ERR_DTYPE;          // fp8 activations come from the bf16 / fp8 MFMA paths
    const int bk = dtype == VV_BF16 ? 64 : (dtype == VV_FP8 ? 128 : 32);
    if (M <= 0 || N <= 0 || K <= 0 || N % (out_dtype == VV_BF16 ? 8 : (out_dtype == VV_FP8 ? 16 : 4))) return VV_ERR_SHAPE;
    if (mode == MODE_DENSE ? (K % (bk / 8)) != 0 : (K % bk) != 0) return VV_ERR_SHAPE;
    if (mode == MODE_FIRST && (!vv_is_pow2(din) || cin != 1)) return VV_ERR_SHAPE;
    // fp8 with Cin = 64 (64-byte voxel rows): a 128-byte staged row is a pair of w-adjacent taps (stride-2 conv only)
    const bool pair = mode == MODE_CONV && dtype == VV_FP8 && cin == 64;
    if ((mode == MODE_CONV || mode == MODE_CONVT) && !pair && (!vv_is_pow2(din) || cin % bk || !vv_is_pow2(cin / bk))) return VV_ERR_SHAPE;
    if (pair && !vv_is_pow2(din)) return VV_ERR_SHAPE;
    if (!vv_aligned16(x) || !vv_aligned16(w) || !vv_aligned16(y)) return VV_ERR_ALIGN;
    const Plan p = make_plan(mode, M, N, K, dtype);
    const size_t a_bytes = mode == MODE_DENSE ? (size_t)M * K * vv_dtype_size(dtype)
                           : (size_t)batch * din * din * din * cin * (mode == MODE_FIRST ? 4 : vv_dtype_size(dtype));
    if (a_bytes >= 0xFFFFFFF0ull || (size_t)p.nparity * N * K * vv_dtype_size(dtype) >= 0xFFFFFFF0ull) return VV_ERR_SHAPE;
    if (p.split > 1 && (!ws || ws_bytes < p.ws_bytes || !vv_aligned16(ws))) return VV_ERR_WORKSPACE;
    IgemmArgs a;
    a.A = x; a.W = w; a.Out = y;
    a.partial = p.split > 1 ? reinterpret_cast<float *>(ws) : nullptr;
    a.scale = scale; a.shift = shift;
    a.M = M; a.N = N; a.K = K;
    a.din_log2 = mode == MODE_DENSE ? 0 : vv_log2(din);
    a.cin = cin;
    a.cpt_log2 = (mode == MODE_DENSE || mode == MODE_FIRST || pair) ? 0 : vv_log2(cin / bk);
    a.pair = pair ? 1 : 0;
    a.nchunks = (K + bk - 1) / bk;
    a.chunks_per_split = p.cps;
    a.act = act;
    a.out_kind = out_dtype == VV_BF16 ? 0 : (out_dtype == VV_FP8 ? 3 : 1);
    a.batch = batch;
    a.a_bytes = (unsigned)a_bytes;
    a.w_bytes = (unsigned)((size_t)p.nparity * N * K * vv_dtype_size(dtype));
    a.nsplit = p.split;
    a.nparity = p.nparity;
    a.pos_major = use_pos_major(mode, din, batch);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (mode == MODE_FIRST) return dtype == VV_BF16 ? launch_t<__bf16, MODE_FIRST>(a, p, st) : launch_t<float, MODE_FIRST>(a, p, st);
    if (dtype == VV_BF16) {
        if (mode == MODE_DENSE) return launch_t<__bf16, MODE_DENSE>(a, p, st);
        if (mode == MODE_CONV) return launch_t<__bf16, MODE_CONV>(a, p, st);
        return launch_t<__bf16, MODE_CONVT>(a, p, st);
    }
    if (dtype == VV_FP8) {
        if (mode == MODE_DENSE) return launch_t<vv_fp8, MODE_DENSE>(a, p, st);
        if (mode == MODE_CONV) return launch_t<vv_fp8, MODE_CONV>(a, p, st);
        return launch_t<vv_fp8, MODE_CONVT>(a, p, st);
    }
    if (mode == MODE_DENSE) return launch_t<float, MODE_DENSE>(a, p, st);
    if (mode == MODE_CONV) return launch_t<float, MODE_CONV>(a, p, st);
    return launch_t<float, MODE_CONVT>(a, p, st);
}

}  // namespace

VV_EXPORT size_t vv_conv3d_k4s2_workspace_bytes(int batch, int side, int cin, int cout, int dtype) {
    const int o = side / 2;
    return make_plan(MODE_CONV, batch * o * o * o, cout, 64 * cin, dtype).ws_bytes;
}

VV_EXPORT int vv_conv3d_k4s2_fwd(const void *x, const void *w_packed, const float *scale, const float *shift, void *y,
                                 int batch, int side, int cin, int cout, int act, int dtype, void *workspace,
                                 size_t workspace_bytes, void *stream) {
    if (batch <= 0 || side < 2 || !vv_is_pow2(side)) return VV_ERR_SHAPE;
    const int o = side / 2;
    if ((long)batch * side * side * side * cin >= (1L << 31)) return VV_ERR_SHAPE;
    return run_igemm(MODE_CONV, x, w_packed, scale, shift, y, batch * o * o * o, cout, 64 * cin, side, cin, act, dtype,
                     dtype, workspace, workspace_bytes, stream, batch);
}

VV_EXPORT int vv_conv3d_k4s2_fwd_io(const void *x, const void *w_packed, const float *scale, const float *shift, void *y,
                                    int batch, int side, int cin, int cout, int act, int dtype, int out_dtype, void *workspace,
                                    size_t workspace_bytes, void *stream) {
    if (batch <= 0 || side < 2 || !vv_is_pow2(side)) return VV_ERR_SHAPE;
    const int o = side / 2;
    if ((long)batch * side * side * side * cin >= (1L << 31)) return VV_ERR_SHAPE;
    return run_igemm(MODE_CONV, x, w_packed, scale, shift, y, batch * o * o * o, cout, 64 * cin, side, cin, act, dtype,
                     out_dtype, workspace, workspace_bytes, stream, batch);
}

VV_EXPORT size_t vv_convT3d_k4s2_workspace_bytes(int batch, int side, int cin, int cout, int dtype) {
    return make_plan(MODE_CONVT, batch * side * side * side, cout, 8 * cin, dtype).ws_bytes;
}

VV_EXPORT int vv_convT3d_k4s2_fwd(const void *x, const void *w_packed, const float *scale, const float *shift, void *y,
                                  int batch, int side, int cin, int cout, int act, int dtype, void *workspace,
                                  size_t workspace_bytes, void *stream) {
    if (batch <= 0 || side < 1 || !vv_is_pow2(side)) return VV_ERR_SHAPE;
    if ((long)batch * side * side * side * cin >= (1L << 31)) return VV_ERR_SHAPE;
    return run_igemm(MODE_CONVT, x, w_packed, scale, shift, y, batch * side * side * side, cout, 8 * cin, side, cin, act,
                     dtype, dtype, workspace, workspace_bytes, stream, batch);
}

VV_EXPORT int vv_convT3d_k4s2_fwd_io(const void *x, const void *w_packed, const float *scale, const float *shift, void *y,
                                     int batch, int side, int cin, int cout, int act, int dtype, int out_dtype, void *workspace,
                                     size_t workspace_bytes, void *stream) {
    if (batch <= 0 || side < 1 || !vv_is_pow2(side)) return VV_ERR_SHAPE;
    if ((long)batch * side * side * side * cin >= (1L << 31)) return VV_ERR_SHAPE;
    return run_igemm(MODE_CONVT, x, w_packed, scale, shift, y, batch * side * side * side, cout, 8 * cin, side, cin, act,
                     dtype, out_dtype, workspace, workspace_bytes, stream, batch);
}

VV_EXPORT size_t vv_dense_workspace_bytes(int m, int n, int k, int dtype) { return make_plan(MODE_DENSE, m, n, k, dtype).ws_bytes; }

VV_EXPORT int vv_dense_fwd(const void *x, const void *w_packed, const float *scale, const float *shift, void *y, int m,
                           int n, int k, int act, int dtype, int out_dtype, void *workspace, size_t workspace_bytes,
                           void *stream) {
    if ((long)m * k >= (1L << 31)) return VV_ERR_SHAPE;
    return run_igemm(MODE_DENSE, x, w_packed, scale, shift, y, m, n, k, 0, 0, act, dtype, out_dtype, workspace,
                     workspace_bytes, stream);
}

// Cin = 1 first layer on the same MFMA tile engine: w_packed = vv_pack_conv_k4(cin = 1) = [Cout][64 taps].
VV_EXPORT int vv_conv3d_first_fwd_io(const float *x, const void *w_packed, const float *scale, const float *shift, void *y,
                                     int batch, int side, int cout, int act, int dtype, int out_dtype, void *stream) {
    if (batch <= 0 || side < 2 || !vv_is_pow2(side)) return VV_ERR_SHAPE;
    if ((long)batch * side * side * side >= (1L << 31)) return VV_ERR_SHAPE;
    const int o = side / 2;
    if (out_dtype == VV_FP8) {                  // e4m3fn output: the bf16 plane-form kernel only (side >= 32, Cout 64)
        if (dtype != VV_BF16 || cout != 64 || side < 32 || side > 256) return VV_ERR_DTYPE;
        if (!x || !w_packed || !y) return VV_ERR_NULL;
        if (!vv_aligned16(y) || !vv_aligned16(w_packed)) return VV_ERR_ALIGN;
        return vv_first_conv_bf16_launch(x, w_packed, scale, shift, y, batch, side, act, stream, 1);
    }
    if (out_dtype != dtype) return VV_ERR_DTYPE;
    if (dtype == VV_BF16 && cout == 64 && x && w_packed && y && vv_aligned16(y) && vv_aligned16(w_packed) && !vv_hook("VV_NO_FIRSTCONV"))
        return vv_first_conv_bf16_launch(x, w_packed, scale, shift, y, batch, side, act, stream, 0);
    return run_igemm(MODE_FIRST, x, w_packed, scale, shift, y, batch * o * o * o, cout, 64, side, 1, act, dtype, dtype, nullptr,
                     0, stream, batch);
}

VV_EXPORT int vv_conv3d_first_fwd(const float *x, const void *w_packed, const float *scale, const float *shift, void *y,
                                  int batch, int side, int cout, int act, int dtype, void *stream) {
    return vv_conv3d_first_fwd_io(x, w_packed, scale, shift, y, batch, side, cout, act, dtype, dtype, stream);
}

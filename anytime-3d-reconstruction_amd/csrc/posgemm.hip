// Position-major split-K GEMM for the two layers between the 4^3 and the 2^3 grid (bf16, gfx950):
//
//   conv  : Conv3D          k4 s2 SAME,  4^3 x Cin -> 2^3 x Cout   (autoencoder3D.py:26-39;  32^3 model: 256 -> 512)
//   convT : Conv3DTranspose k4 s2 SAME,  2^3 x Cin -> 4^3 x Cout   (autoencoder3D.py:41-54;  32^3 model: 512 -> 256)
//
// 58 % of their dense taps fall into the SAME padding, their weights (16.8 MB) outweigh their activations, and the whole
// layer is only 432 MFMA-microseconds per SIMD: it has to be cut along K to fill 256 CUs, and each weight byte should cross
// L2 -> LDS once.  Per OUTPUT POSITION the layer is a plain GEMM
//      C_p[sample, n] = sum over the (tap, 64-channel chunk) pairs that are valid for p of  x[sample, cell(p, tap), chunk] * W[tap][chunk][n]
// with M = the batch.  The implicit GEMM runs the same decomposition on 128 x 128 tiles with a 2-deep ring and a tap list it
// rebuilds with LDS atomics per workgroup (rocprofv3: 11-20 % MFMA busy, > 50 % of the wave cycles parked on s_waitcnt /
// s_barrier).  Here:
//
//   tile      : 256 samples x 128 channels per workgroup (8 waves, 64 x 64 each on v_mfma_f32_32x32x16_bf16): every staged
//               weight row feeds 256 samples, every staged activation row 128 channels (48 B/clk/CU at the full MFMA rate)
//   K shares  : the host cuts each position's chunk list into equal shares of <= ~16 chunks (table in the kernel arguments;
//               positions with one share write the finished layer output, the others float32 slabs that pg_reduce_kernel sums
//               in share order -- deterministic)
//   ring      : 3 stages of (256 + 128) x 128 B by LDS-DMA, issued two chunks ahead with a counted vmcnt; the per-chunk barrier
//               sits in front of the chunk's last k-step, whose MFMAs cover it, and the next chunk's first fragments are read
//               before them
//   weights   : the [tap][Cin/64][Cout][64] / [parity][tap][Cin/64][Cout][64] panels of skip_direct.hip (a stage = 16 KiB
//               contiguous)
//   order     : the shares and channel tiles of one position are neighbours in the XCD-aware work order (they read the same
//               activation rows)
#include <stdlib.h>

#include <type_traits>

#include "common.h"

namespace {

typedef __attribute__((ext_vector_type(4))) unsigned pg_u4;

constexpr int PG_BM = 256, PG_BN = 128;
constexpr int PG_STAGE = (PG_BM + PG_BN) * 128;       // 49,152 B
constexpr int PG_NST = 3;
constexpr int PG_LDS = PG_NST * PG_STAGE;             // 147,456 B
constexpr int PG_EPITCH_F32 = PG_BN * 4 + 16, PG_EPITCH_BF16 = PG_BN * 2 + 16;
constexpr int PG_MAXPOS = 64;

struct PgArgs {
    const void *x;
    const void *w;
    const float *scale;
    const float *shift;
    void *y;
    float *slabs;                      // [sum of shares][mtiles][256][cout] float32
    int batch, cin, cout, act;
    unsigned x_bytes, w_bytes;
    int ntn, mtiles, npos, nitems;     // channel tiles, sample tiles, output positions, sum of shares over the positions
    int force_slabs;                   // 1: one-share positions write a float32 slab too (the consumer sums / activates: vv_pg_conv_slabs)
    unsigned char nsplit[PG_MAXPOS];   // shares of position p
    unsigned short first[PG_MAXPOS];   // first share slot of position p
};

// ---- the chunk list of an output position.  Axis by axis a position has a run of `cnt` valid taps starting at `lo`.
// conv  (input side n = 4, output o in 0..1): tap t in [max(0, 1 - 2o), min(3, n - 2o)], input cell 2o - 1 + t
// convT (input side n = 2, output o in 0..3): parity q = o & 1, m = o >> 1, tap a in {0,1} with 0 <= m + q - a < n
struct PgAxis { int lo, cnt; };
template <int MODE>
__host__ __device__ inline PgAxis pg_axis(int o) {
    PgAxis r;
    if (MODE == 0) {
        const int n = 4;
        r.lo = 1 - 2 * o > 0 ? 1 - 2 * o : 0;
        const int hi = n - 2 * o < 3 ? n - 2 * o : 3;
        r.cnt = hi - r.lo + 1;
    } else {
        const int n = 2, q = o & 1, m = o >> 1;
        // a = 0 needs m + q < n; a = 1 needs m + q >= 1
        const bool a0 = m + q < n, a1 = m + q >= 1;
        r.lo = a0 ? 0 : 1;
        r.cnt = (a0 ? 1 : 0) + (a1 ? 1 : 0);
    }
    return r;
}

#define PG_RD(DST, ADDR, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(DST) : "v"(ADDR), "n"(OFF) : "memory")
#define PG_WAIT4(N, F) asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(F[0]), "+v"(F[1]), "+v"(F[2]), "+v"(F[3]) : "n"(N) : "memory")

template <int MODE>
__global__ __launch_bounds__(512, 1) void pg_kernel(const PgArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;              // 64-row block, 64-channel block
    const int NC = a.cin >> 6;
    constexpr int VIN = MODE == 0 ? 64 : 8;               // input cells per sample
    constexpr int LI = MODE == 0 ? 2 : 1;                 // log2(input side)

    // XCD-aware order: item = ((slot * ntn + channel tile) * mtiles + sample tile); the shares / channel tiles of a position
    // sit next to each other and walk one XCD
    const int nwg = gridDim.x;
    const int wi = (nwg & 7) ? (int)blockIdx.x : (int)(blockIdx.x & 7) * (nwg >> 3) + (int)(blockIdx.x >> 3);
    const int mt = wi % a.mtiles;
    const int nt = (wi / a.mtiles) % a.ntn;
    const int slot = wi / (a.mtiles * a.ntn);
    // position of this share slot: the last p with first[p] <= slot (one table entry per lane, one ballot)
    const bool le = lane < a.npos && (int)a.first[lane < PG_MAXPOS ? lane : 0] <= slot;
    const int p = __builtin_amdgcn_readfirstlane(__builtin_popcountll(__ballot(le)) - 1);
    const int share = slot - (int)a.first[p], nshare = a.nsplit[p];
    // output position -> per-axis tap runs
    int od, oh, ow;
    if (MODE == 0) { od = (p >> 2) & 1; oh = (p >> 1) & 1; ow = p & 1; }
    else { od = (p >> 4) & 3; oh = (p >> 2) & 3; ow = p & 3; }
    const PgAxis xd = pg_axis<MODE>(od), xh = pg_axis<MODE>(oh), xw = pg_axis<MODE>(ow);
    const int nchunks = xd.cnt * xh.cnt * xw.cnt * NC;
    const int c_begin = (int)((long)nchunks * share / nshare), c_end = (int)((long)nchunks * (share + 1) / nshare);
    const int nmine = c_end - c_begin;
    const int m0 = mt * PG_BM, n0 = nt * PG_BN;

    const u32x4 rsx = vv_make_rsrc(a.x, a.x_bytes), rsw = vv_make_rsrc(a.w, a.w_bytes);
    const unsigned lds0 = (unsigned)(unsigned long long)(lptr_t)smem;

    // ---- producers: per stage 32 activation pieces (8 rows = 8 samples x 128 B) + 16 weight pieces over 8 waves
    unsigned a_lane[4], w_lane[2];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int row = (wave * 4 + k) * 8 + (lane >> 3);
        const int b = m0 + row;
        const int slot16 = (lane & 7) ^ ((row >> 1) & 7);
        a_lane[k] = b < a.batch ? (unsigned)(b * VIN * (a.cin * 2) + slot16 * 16) : 0xFFFFFFF0u;   // rows past the batch: zeros
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int nrow = (wave * 2 + k) * 8 + (lane >> 3);
        const int slot16 = (lane & 7) ^ ((nrow >> 1) & 7);
        w_lane[k] = n0 + nrow < a.cout ? (unsigned)(nrow * 128 + slot16 * 16) : 0xFFFFFFF0u;
    }
    // chunks are issued in order: (tap digits id, ih, iw; channel chunk c) of the NEXT chunk to issue ride along as scalars, so
    // that no integer division sits in the loop
    int nx_c, nx_w, nx_h, nx_d;
    {
        const int tapi = c_begin / NC;
        nx_c = c_begin - tapi * NC;
        nx_w = tapi % xw.cnt;
        const int r1 = tapi / xw.cnt;
        nx_h = r1 % xh.cnt;
        nx_d = r1 / xh.cnt;
        nx_c = __builtin_amdgcn_readfirstlane(nx_c); nx_w = __builtin_amdgcn_readfirstlane(nx_w);
        nx_h = __builtin_amdgcn_readfirstlane(nx_h); nx_d = __builtin_amdgcn_readfirstlane(nx_d);
    }
    auto issue = [&](int buf) {                    // the next chunk of this share into ring buffer buf
        const int c = nx_c, iw = nx_w, ih = nx_h, id = nx_d;
        unsigned soa, sow;
        if (MODE == 0) {
            const int td = xd.lo + id, th = xh.lo + ih, tw = xw.lo + iw;
            const int cell = ((((2 * od - 1 + td) << LI) + (2 * oh - 1 + th)) << LI) + (2 * ow - 1 + tw);
            soa = (unsigned)(cell * (a.cin * 2) + c * 128);
            sow = (unsigned)(((((td * 4 + th) * 4 + tw) * NC + c) * a.cout + n0) * 128);
        } else {
            const int ad = xd.lo + id, ah = xh.lo + ih, aw = xw.lo + iw;
            const int cd = (od >> 1) + (od & 1) - ad, ch = (oh >> 1) + (oh & 1) - ah, cw = (ow >> 1) + (ow & 1) - aw;
            const int cell = (((cd << LI) + ch) << LI) + cw;
            const int par = ((od & 1) * 2 + (oh & 1)) * 2 + (ow & 1), ta = (ad * 2 + ah) * 2 + aw;
            soa = (unsigned)(cell * (a.cin * 2) + c * 128);
            sow = (unsigned)((((par * 8 + ta) * NC + c) * a.cout + n0) * 128);
        }
        const unsigned da = lds0 + buf * PG_STAGE + wave * 4096, db = lds0 + buf * PG_STAGE + PG_BM * 128 + wave * 2048;
#pragma unroll
        for (int k = 0; k < 4; ++k) vv_dma16(rsx, a_lane[k], soa, da + k * 1024);
#pragma unroll
        for (int k = 0; k < 2; ++k) vv_dma16(rsw, w_lane[k], sow, db + k * 1024);
        if (++nx_c == NC) {
            nx_c = 0;
            if (++nx_w == xw.cnt) {
                nx_w = 0;
                if (++nx_h == xh.cnt) { nx_h = 0; ++nx_d; }
            }
        }
    };
    constexpr int LPC = 6;                        // LDS-DMA instructions per wave per chunk

    // ---- consumer addressing: 32x32x16 fragments, weights first (D[n][m]: lane = sample row, registers walk channels)
    const int fr = lane & 31, fh = lane >> 5;
    unsigned ab[4], bb[4];                         // byte address of k-step ks inside stage 0; row tiles / channel tiles at + 32 rows
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const int ra = wm * 64 + fr, rb = wn * 64 + fr;
        ab[ks] = lds0 + (unsigned)(ra * 128 + (((ks * 2 + fh) ^ ((ra >> 1) & 7)) << 4));
        bb[ks] = lds0 + (unsigned)(PG_BM * 128 + rb * 128 + (((ks * 2 + fh) ^ ((rb >> 1) & 7)) << 4));
    }
    f32x16 acc[2][2];                              // [channel tile][row tile]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // fragment set = {row tile 0, row tile 1, channel tile 0, channel tile 1} of one k-step (rows 32 apart: 4096 B)
    auto rd = [&](pg_u4 *F, int ks, unsigned stoff) {
        const unsigned aa = ab[ks] + stoff, bbv = bb[ks] + stoff;
        PG_RD(F[0], aa, 0); PG_RD(F[1], aa, 4096); PG_RD(F[2], bbv, 0); PG_RD(F[3], bbv, 4096);
    };
    auto mma = [&](const pg_u4 *F) {
#pragma unroll
        for (int nt_ = 0; nt_ < 2; ++nt_)
#pragma unroll
            for (int mt_ = 0; mt_ < 2; ++mt_)
                acc[nt_][mt_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8 *>(&F[2 + nt_]),
                                                                        *reinterpret_cast<const bf16x8 *>(&F[mt_]), acc[nt_][mt_], 0, 0, 0);
    };

    // ---- pipeline: chunk i lives in ring buffer i % 3; chunks i + 1 and i + 2 are in flight while i is multiplied
    if (nmine > 0) issue(0);
    if (nmine > 1) issue(1);
    if (nmine > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPC) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (nmine > 2) issue(2);
    pg_u4 P[4], Q[4];
    int buf = 0;
    if (nmine > 0) rd(P, 0, 0u);
#pragma unroll 1
    for (int i = 0; i < nmine; ++i) {
        const unsigned st = (unsigned)buf * PG_STAGE;
        rd(Q, 1, st);
        PG_WAIT4(4, P);
        mma(P);
        rd(P, 2, st);
        PG_WAIT4(4, Q);
        mma(Q);
        rd(Q, 3, st);
        PG_WAIT4(4, P);
        mma(P);
        PG_WAIT4(0, Q);
        // chunk i + 1 has landed (chunk i + 2, issued after it, may still fly); behind the barrier buffer `buf` is free
        if (i + 2 < nmine) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPC) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (i + 3 < nmine) issue(buf);
        buf = buf == PG_NST - 1 ? 0 : buf + 1;
        if (i + 1 < nmine) rd(P, 0, (unsigned)buf * PG_STAGE);
        mma(Q);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();

    // ---- epilogue.  lane = sample row fr of row tile mt_, registers walk channels (q & 3) + 8 (q >> 2) + 4 fh of channel
    // tile nt_.  One share: folded BN + activation, bf16, the layer's output.  Several: float32 slab.
    if (a.force_slabs) {
        // Slabs for a consumer that sums them itself (latent_tail.hip, lt_e5x_kernel): the accumulators go out in FRAGMENT order, one
        // 1 KiB store per wave and register quad -- f32x4 index ((((wave * 2 + nt_) * 2 + mt_) * 4 + g) * 64 + lane) of the
        // workgroup's [256 rows][128 channels] piece, pieces laid out [slot][sample tile][channel tile] -- no LDS transpose, no barrier.
        // Rows past the batch carry zeros (their activation rows were zero-filled).
        f32x4 *piece = reinterpret_cast<f32x4 *>(a.slabs) + (((size_t)slot * a.mtiles + mt) * a.ntn + nt) * (PG_BM * PG_BN / 4);
#pragma unroll
        for (int nt_ = 0; nt_ < 2; ++nt_)
#pragma unroll
            for (int mt_ = 0; mt_ < 2; ++mt_)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    piece[(((wave * 2 + nt_) * 2 + mt_) * 4 + g) * 64 + lane] =
                        f32x4{acc[nt_][mt_][4 * g], acc[nt_][mt_][4 * g + 1], acc[nt_][mt_][4 * g + 2], acc[nt_][mt_][4 * g + 3]};
        return;
    }
    const bool final_out = nshare == 1;
    const int m_rows = a.batch - m0 < PG_BM ? a.batch - m0 : PG_BM;
    auto fill = [&](auto act_c, auto fin_c) {
        constexpr int ACT = decltype(act_c)::value;
        constexpr bool FIN = decltype(fin_c)::value;
        constexpr int PITCH = FIN ? PG_EPITCH_BF16 : PG_EPITCH_F32;
#pragma unroll
        for (int nt_ = 0; nt_ < 2; ++nt_)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int cl = wn * 64 + nt_ * 32 + 8 * g + 4 * fh;
                f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
                if (FIN && a.scale && n0 + cl < a.cout) sc = *reinterpret_cast<const f32x4 *>(a.scale + n0 + cl);
                if (FIN && a.shift && n0 + cl < a.cout) sh = *reinterpret_cast<const f32x4 *>(a.shift + n0 + cl);
#pragma unroll
                for (int mt_ = 0; mt_ < 2; ++mt_) {
                    const int rl = wm * 64 + mt_ * 32 + fr;
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float t = acc[nt_][mt_][4 * g + e];
                        if (FIN) {
                            t = t * sc[e] + sh[e];
                            if (ACT == VV_ACT_ELU) { const float em = __expf(fminf(t, 0.f)) - 1.f; t = t > 0.f ? t : em; }
                            else if (ACT == VV_ACT_RELU) t = fmaxf(t, 0.f);
                            else if (ACT == VV_ACT_LRELU) t = t > 0.f ? t : 0.3f * t;
                        }
                        v[e] = t;
                    }
                    if (FIN) {
                        bf16x4 o;
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = static_cast<__bf16>(v[e]);
                        *reinterpret_cast<bf16x4 *>(smem + rl * PITCH + cl * 2) = o;
                    } else {
                        *reinterpret_cast<f32x4 *>(smem + rl * PITCH + cl * 4) = v;
                    }
                }
            }
    };
    if (final_out) {
        switch (a.act) {
            case VV_ACT_ELU: fill(std::integral_constant<int, VV_ACT_ELU>{}, std::true_type{}); break;
            case VV_ACT_RELU: fill(std::integral_constant<int, VV_ACT_RELU>{}, std::true_type{}); break;
            case VV_ACT_LRELU: fill(std::integral_constant<int, VV_ACT_LRELU>{}, std::true_type{}); break;
            default: fill(std::integral_constant<int, VV_ACT_NONE>{}, std::true_type{}); break;
        }
    } else {
        fill(std::integral_constant<int, VV_ACT_NONE>{}, std::false_type{});
    }
    __syncthreads();
    constexpr int VOUT = MODE == 0 ? 8 : 64;              // output cells per sample; p is the output cell index
    if (final_out) {
        const int cpr = PG_BN * 2 / 16;
        for (int id = tid; id < PG_BM * cpr; id += 512) {
            const int rl = id / cpr, c = id % cpr;
            if (rl < m_rows && n0 + c * 8 < a.cout)
                *reinterpret_cast<uint4 *>(reinterpret_cast<char *>(a.y) + (((size_t)(m0 + rl) * VOUT + p) * a.cout + n0) * 2 + c * 16) =
                    *reinterpret_cast<const uint4 *>(smem + rl * PG_EPITCH_BF16 + c * 16);
        }
    } else {
        const int cpr = PG_BN * 4 / 16;
        float *slab = a.slabs + ((size_t)slot * a.mtiles + mt) * PG_BM * a.cout;
        for (int id = tid; id < PG_BM * cpr; id += 512) {
            const int rl = id / cpr, c = id % cpr;
            if (rl < m_rows && n0 + c * 4 < a.cout)
                *reinterpret_cast<uint4 *>(reinterpret_cast<char *>(slab) + ((size_t)rl * a.cout + n0) * 4 + c * 16) =
                    *reinterpret_cast<const uint4 *>(smem + rl * PG_EPITCH_F32 + c * 16);
        }
    }
}

// Sums the slabs of every position that was cut into several shares, in share order, applies the folded BN + activation and
// writes the layer output: one channel quad per thread.
template <int MODE, int ACT>
__global__ __launch_bounds__(256) void pg_reduce_kernel(const PgArgs a) {
    constexpr int VOUT = MODE == 0 ? 8 : 64;
    const int n4 = a.cout >> 2;
    const size_t per_pos = (size_t)a.batch * n4;
    const size_t total = (size_t)a.npos * per_pos;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int p = (int)(i / per_pos);
        const int ns = a.nsplit[p];
        if (ns == 1) continue;
        const size_t r = i - (size_t)p * per_pos;
        const int b = (int)(r / n4), c4 = (int)(r % n4);
        const int mt = b / PG_BM, rl = b - mt * PG_BM;
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        for (int sp = 0; sp < ns; ++sp)
            s += *reinterpret_cast<const f32x4 *>(a.slabs + (((size_t)(a.first[p] + sp) * a.mtiles + mt) * PG_BM + rl) * a.cout + c4 * 4);
        f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
        if (a.scale) sc = *reinterpret_cast<const f32x4 *>(a.scale + c4 * 4);
        if (a.shift) sh = *reinterpret_cast<const f32x4 *>(a.shift + c4 * 4);
        bf16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float t = s[e] * sc[e] + sh[e];
            if (ACT == VV_ACT_ELU) { const float em = __expf(fminf(t, 0.f)) - 1.f; t = t > 0.f ? t : em; }
            else if (ACT == VV_ACT_RELU) t = fmaxf(t, 0.f);
            else if (ACT == VV_ACT_LRELU) t = t > 0.f ? t : 0.3f * t;
            o[e] = static_cast<__bf16>(t);
        }
        *reinterpret_cast<bf16x4 *>(reinterpret_cast<__bf16 *>(a.y) + ((size_t)b * VOUT + p) * a.cout + c4 * 4) = o;
    }
}

// Host plan: shares per position so that a share holds at most `target` chunks, target = the average chunks per CU rounded up
// to whole taps.
template <int MODE>
void pg_plan(PgArgs &a, int batch, int cin, int cout) {
    const int NC = cin / 64;
    a.npos = MODE == 0 ? 8 : 64;
    a.ntn = (cout + PG_BN - 1) / PG_BN;
    a.mtiles = (batch + PG_BM - 1) / PG_BM;
    int chunks[PG_MAXPOS];
    long total = 0;
    for (int p = 0; p < a.npos; ++p) {
        int od, oh, ow;
        if (MODE == 0) { od = (p >> 2) & 1; oh = (p >> 1) & 1; ow = p & 1; }
        else { od = (p >> 4) & 3; oh = (p >> 2) & 3; ow = p & 3; }
        chunks[p] = pg_axis<MODE>(od).cnt * pg_axis<MODE>(oh).cnt * pg_axis<MODE>(ow).cnt * NC;
        total += chunks[p];
    }
    static const int target_env = vv_hook("VV_PG_TARGET") ? atoi(vv_hook("VV_PG_TARGET")) : 0;
    long avg = (total * a.ntn * a.mtiles + 255) / 256;
    int target = (int)((avg + NC - 1) / NC) * NC;
    if (target < 8) target = 8;
    if (target_env > 0) target = target_env;
    int slot = 0;
    for (int p = 0; p < a.npos; ++p) {
        int ns = (chunks[p] + target - 1) / target;
        if (ns < 1) ns = 1;
        if (ns > 255) ns = 255;
        a.nsplit[p] = (unsigned char)ns;
        a.first[p] = (unsigned short)slot;
        slot += ns;
    }
    a.nitems = slot;
}

template <int MODE>
size_t pg_ws_bytes(int batch, int cin, int cout) {
    PgArgs a;
    pg_plan<MODE>(a, batch, cin, cout);
    return (size_t)a.nitems * a.mtiles * PG_BM * cout * sizeof(float);
}

template <int MODE>
int pg_run(const void *x, const void *w, const float *scale, const float *shift, void *y, int batch, int cin, int cout, int act, void *ws,
           size_t ws_bytes, hipStream_t st) {
    static const bool attr = [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&pg_kernel<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, PG_LDS);
        return true;
    }();
    (void)attr;
    constexpr int VIN = MODE == 0 ? 64 : 8, VOUT = MODE == 0 ? 8 : 64;
    const size_t sample_in = (size_t)VIN * cin * 2, sample_out = (size_t)VOUT * cout * 2;
    int per = (int)((0x7FFFFFFFull / sample_in) / PG_BM) * PG_BM;        // 32-bit buffer offsets: <= 2 GiB of input per launch
    if (per < PG_BM) return VV_ERR_SHAPE;
    for (int s0 = 0; s0 < batch; s0 += per) {
        const int nb = batch - s0 < per ? batch - s0 : per;
        PgArgs a;
        pg_plan<MODE>(a, nb, cin, cout);
        const size_t need = (size_t)a.nitems * a.mtiles * PG_BM * cout * sizeof(float);
        if (!ws || ws_bytes < need || !vv_aligned16(ws)) return VV_ERR_WORKSPACE;
        a.x = reinterpret_cast<const char *>(x) + (size_t)s0 * sample_in;
        a.y = reinterpret_cast<char *>(y) + (size_t)s0 * sample_out;
        a.w = w; a.scale = scale; a.shift = shift;
        a.slabs = reinterpret_cast<float *>(ws);
        a.batch = nb; a.cin = cin; a.cout = cout; a.act = act;
        a.force_slabs = 0;
        a.x_bytes = (unsigned)((size_t)nb * sample_in);
        a.w_bytes = (unsigned)((size_t)64 * cin * cout * 2);
        VV_LAUNCH(pg_kernel<MODE>, dim3(a.nitems * a.ntn * a.mtiles), dim3(512), PG_LDS, st, a);
        int rc = vv_launch_status();
        if (rc != VV_OK) return rc;
        bool any = false;
        for (int p = 0; p < a.npos; ++p) any = any || a.nsplit[p] > 1;
        if (any) {
            const size_t total = (size_t)a.npos * nb * (cout / 4);
            const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
            switch (act) {
                case VV_ACT_ELU: VV_LAUNCH((pg_reduce_kernel<MODE, VV_ACT_ELU>), dim3(blocks), dim3(256), 0, st, a); break;
                case VV_ACT_RELU: VV_LAUNCH((pg_reduce_kernel<MODE, VV_ACT_RELU>), dim3(blocks), dim3(256), 0, st, a); break;
                case VV_ACT_LRELU: VV_LAUNCH((pg_reduce_kernel<MODE, VV_ACT_LRELU>), dim3(blocks), dim3(256), 0, st, a); break;
                default: VV_LAUNCH((pg_reduce_kernel<MODE, VV_ACT_NONE>), dim3(blocks), dim3(256), 0, st, a); break;
            }
            rc = vv_launch_status();
            if (rc != VV_OK) return rc;
        }
    }
    return VV_OK;
}

bool pg_shape_ok(int cin, int cout) {
    return cin >= 64 && cin % 64 == 0 && cout >= 8 && cout % 8 == 0 && (size_t)64 * cin * cout * 2 < 0xFFFFFFF0ull;
}

}  // namespace

// ---- internal (common.h): the 4^3 -> 2^3 convolution as float32 slabs ONLY -- no reduce launch, no folded BN: the consumer
// (latent_tail.hip: lt_e5x_kernel) sums the shares of a position in share order, exactly as pg_reduce_kernel does, while it builds
// its own MFMA operand.  One launch; the whole batch must fit 32-bit buffer offsets.
size_t vv_pg_conv_slab_bytes(int batch, int cin, int cout) {
    if (batch <= 0 || !pg_shape_ok(cin, cout) || (size_t)batch * 64 * cin * 2 > 0x7FFFFFFFull) return 0;
    PgArgs a;
    pg_plan<0>(a, batch, cin, cout);
    return (size_t)a.nitems * a.mtiles * a.ntn * PG_BM * PG_BN * sizeof(float);      // whole [256][128] pieces in fragment order
}

int vv_pg_conv_slabs(const void *x, const void *w, int batch, int cin, int cout, void *ws, size_t ws_bytes, hipStream_t st, VvPgSlabPlan *plan) {
    static const bool attr = [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&pg_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, PG_LDS);
        return true;
    }();
    (void)attr;
    const size_t need = vv_pg_conv_slab_bytes(batch, cin, cout);
    if (!need) return VV_ERR_SHAPE;
    if (!ws || ws_bytes < need || !vv_aligned16(ws)) return VV_ERR_WORKSPACE;
    PgArgs a;
    pg_plan<0>(a, batch, cin, cout);
    a.x = x; a.y = nullptr; a.w = w; a.scale = nullptr; a.shift = nullptr;
    a.slabs = reinterpret_cast<float *>(ws);
    a.batch = batch; a.cin = cin; a.cout = cout; a.act = VV_ACT_NONE;
    a.force_slabs = 1;
    a.x_bytes = (unsigned)((size_t)batch * 64 * cin * 2);
    a.w_bytes = (unsigned)((size_t)64 * cin * cout * 2);
    VV_LAUNCH(pg_kernel<0>, dim3(a.nitems * a.ntn * a.mtiles), dim3(512), PG_LDS, st, a);
    plan->npos = a.npos; plan->mtiles = a.mtiles; plan->nitems = a.nitems; plan->rows_per_tile = PG_BM; plan->ntn = a.ntn;
    for (int p = 0; p < 8; ++p) { plan->nsplit[p] = a.nsplit[p]; plan->first[p] = a.first[p]; }
    return vv_launch_status();
}

VV_EXPORT int vv_conv3d_k4s2_pos_supported(int side, int cin, int cout, int dtype) {
    return dtype == VV_BF16 && side == 4 && pg_shape_ok(cin, cout);
}

VV_EXPORT int vv_convT3d_k4s2_pos_supported(int side, int cin, int cout, int dtype) {
    return dtype == VV_BF16 && side == 2 && pg_shape_ok(cin, cout);
}

VV_EXPORT size_t vv_conv3d_k4s2_pos_workspace_bytes(int batch, int cin, int cout) {
    if (batch <= 0 || !pg_shape_ok(cin, cout)) return 0;
    return pg_ws_bytes<0>(batch < (1 << 20) ? batch : (1 << 20), cin, cout);
}

VV_EXPORT size_t vv_convT3d_k4s2_pos_workspace_bytes(int batch, int cin, int cout) {
    if (batch <= 0 || !pg_shape_ok(cin, cout)) return 0;
    return pg_ws_bytes<1>(batch < (1 << 20) ? batch : (1 << 20), cin, cout);
}

VV_EXPORT int vv_conv3d_k4s2_pos_fwd(const void *x, const void *w_skip, const float *scale, const float *shift, void *y, int batch,
                                     int side, int cin, int cout, int act, int dtype, void *workspace, size_t workspace_bytes, void *stream) {
    if (!x || !w_skip || !y) return VV_ERR_NULL;
    if (!vv_conv3d_k4s2_pos_supported(side, cin, cout, dtype) || batch <= 0) return VV_ERR_SHAPE;
    if (!vv_aligned16(x) || !vv_aligned16(w_skip) || !vv_aligned16(y) || (scale && !vv_aligned16(scale)) || (shift && !vv_aligned16(shift)))
        return VV_ERR_ALIGN;
    return pg_run<0>(x, w_skip, scale, shift, y, batch, cin, cout, act, workspace, workspace_bytes, reinterpret_cast<hipStream_t>(stream));
}

VV_EXPORT int vv_convT3d_k4s2_pos_fwd(const void *x, const void *w_skip, const float *scale, const float *shift, void *y, int batch,
                                      int side, int cin, int cout, int act, int dtype, void *workspace, size_t workspace_bytes, void *stream) {
    if (!x || !w_skip || !y) return VV_ERR_NULL;
    if (!vv_convT3d_k4s2_pos_supported(side, cin, cout, dtype) || batch <= 0) return VV_ERR_SHAPE;
    if (!vv_aligned16(x) || !vv_aligned16(w_skip) || !vv_aligned16(y) || (scale && !vv_aligned16(scale)) || (shift && !vv_aligned16(shift)))
        return VV_ERR_ALIGN;
    return pg_run<1>(x, w_skip, scale, shift, y, batch, cin, cout, act, workspace, workspace_bytes, reinterpret_cast<hipStream_t>(stream));
}

// Direct strided convolution on MFMA with the input phase tile resident in LDS (bf16, gfx950).
//
// Conv3D k4 s2 SAME (autoencoder3D.py:26-39): out[o] = sum_t x[2o - 1 + t] W[t].  Writing t = 2a + q per axis
// (a, q in {0,1}) gives x index 2(o + a) + q - 1: for a fixed input phase q = (qd,qh,qw) the layer is a k2 s1
// convolution over the phase sub-grid X_q[j] = x[2j + q - 1], j = o + a.  A 4 x 8 x 8 box of outputs therefore needs,
// per phase, only the 5 x 9 x 9 sub-grid tile, and its 8 taps a read shifted windows of that one tile.  The implicit
// GEMM re-fetches every input row from L2 once per tap (64 x 128 B per output row, 2.2x its input from HBM/MALL by
// FETCH_SIZE); here each workgroup stages 8 phase tiles of 405 rows instead of 64 im2col tiles of 256 rows.
//
//   workgroup  : 512 threads = 8 waves, 256 outputs x 128 channels; wave (wm, wn) owns output plane od0 + wm
//                (64 outputs = 2 row tiles) x 64 channels (2 channel tiles); one workgroup per CU, 2 waves per SIMD
//   LDS        : phase tile [405(+3)][128 B], slot XOR jw & 7 (conflict-free for every tap and both ds_read_b128 lane
//                groups at row pitch 9/81, checked exhaustively offline), double buffered: tile q+1 arrives by LDS-DMA
//                one 1 KiB piece per wave per tap while phase q computes; weights [128 co][128 B] per (phase, tap) in
//                a 3-deep ring (slot ^ (co>>1)&7), two pieces per wave per tap, issued three taps ahead
//   sync       : one barrier per tap, placed before the tap's LAST k-step: it publishes chunk c+1 (s_waitcnt vmcnt(N),
//                N counted over the pieces issued since) and frees chunk c's stage, and the first fragment reads of
//                chunk c+1 then fly under the last 4 MFMAs of chunk c; fragment reads are inline asm, one k-step ahead
//   epilogue   : folded BN + activation, LDS transpose, 16-byte stores of whole channel rows (the 256 outputs of a
//                workgroup are contiguous in y when the output side is 8)
#include <stdlib.h>

#include <type_traits>

#include "common.h"

namespace {

template <int N>
__device__ __forceinline__ void wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" : : "n"(N) : "memory");
}

constexpr int CD_CIN = 64, CD_COUT = 128;
constexpr int CD_RB = CD_CIN * 2;                 // bytes per voxel row
constexpr int CD_PIECES = 51;                     // 405 rows in 1 KiB pieces of 8 rows
constexpr int CD_TILE = CD_PIECES * 1024;         // 52,224 B
constexpr int CD_WST = CD_COUT * 128;             // one weight stage: 128 rows x 64 k
constexpr int CD_NST = 3;
constexpr int CD_RING = 2 * CD_TILE;
constexpr int CD_DUMMY = CD_RING + CD_NST * CD_WST;
constexpr int CD_LDS = CD_DUMMY + 1024;           // 154,624 B
constexpr int CD_SP = CD_COUT * 2 + 16;           // epilogue row pitch

__global__ __launch_bounds__(512, 1) void conv_direct_kernel(const __bf16 *__restrict__ x, const __bf16 *__restrict__ w,
                                                             const float *__restrict__ scale, const float *__restrict__ shift,
                                                             void *__restrict__ y, int dout_log2, unsigned x_bytes, unsigned w_bytes,
                                                             int act, int out_fp8) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int lo = dout_log2, no = 1 << lo, li = lo + 1, n = 1 << li;

    // XCD-aware order: consecutive boxes (same sample) run on one XCD and share its L2
    const int nwg = gridDim.x;
    int blk = (nwg & 7) == 0 ? (int)(blockIdx.x & 7) * (nwg >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    const int bxw = no >> 3, bxh = no >> 3, bxd = no >> 2;
    const int bw = blk % bxw; blk /= bxw;
    const int bh = blk % bxh; blk /= bxh;
    const int bd = blk % bxd; const int b = blk / bxd;
    const int od0 = bd * 4, oh0 = bh * 8, ow0 = bw * 8;

    const u32x4 rsx = vv_make_rsrc(x, x_bytes), rsw = vv_make_rsrc(w, w_bytes);
    const unsigned lds0 = (unsigned)(unsigned long long)(lptr_t)smem;

    // ---- producers
    auto issue_x = [&](int q, int slot) {            // slot = 0..6: piece slot*8 + wave of phase q into tile[q & 1]
        const int piece = slot * 8 + wave;
        const int rl = piece * 8 + (lane >> 3);
        const int zd = rl / 81, rem = rl - zd * 81, jh = rem / 9, jw = rem - jh * 9;
        const int g = (lane & 7) ^ (jw & 7);
        const int id = 2 * (od0 + zd) + ((q >> 2) & 1) - 1, ih = 2 * (oh0 + jh) + ((q >> 1) & 1) - 1, iw = 2 * (ow0 + jw) + (q & 1) - 1;
        const bool ok = q < 8 && rl < 405 && (unsigned)id < (unsigned)n && (unsigned)ih < (unsigned)n && (unsigned)iw < (unsigned)n;
        const unsigned vo = ok ? (unsigned)((((((b << li) + id) << li) + ih) << li) + iw) * CD_RB + g * 16 : 0xFFFFFFF0u;
        const unsigned dst = piece < CD_PIECES ? lds0 + (q & 1) * CD_TILE + piece * 1024 : lds0 + CD_DUMMY;
        vv_dma16(rsx, vo, dst);
    };
    auto issue_w = [&](int c) {                      // chunk c = q*8 + a into ring[c % 3]: rows 16*wave .. 16*wave+15
        const int q = c >> 3, a = c & 7;
        const int td = 2 * ((a >> 2) & 1) + ((q >> 2) & 1), th = 2 * ((a >> 1) & 1) + ((q >> 1) & 1), tw = 2 * (a & 1) + (q & 1);
        const int t = (td * 4 + th) * 4 + tw;
        const unsigned st = lds0 + CD_RING + (c % CD_NST) * CD_WST;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = (wave * 2 + i) * 8 + (lane >> 3);
            const int g = (lane & 7) ^ ((row >> 1) & 7);
            const unsigned vo = c < 64 ? (unsigned)row * (64 * CD_RB) + t * CD_RB + g * 16 : 0xFFFFFFF0u;
            vv_dma16(rsw, vo, st + (wave * 2 + i) * 1024);
        }
    };

    // ---- prologue: phase 0 tile, weight chunks 0..2
#pragma unroll 1
    for (int s = 0; s < 7; ++s) issue_x(0, s);
    issue_w(0);
    issue_w(1);
    issue_w(2);
    wait_vm<0>();
    __syncthreads();

    // ---- consumer addressing (LDS byte addresses)
    const int fr = lane & 31, fh = lane >> 5;
    const int rl0 = wm * 81 + (fr >> 3) * 9 + (fr & 7);
    unsigned xo[2][4], wo[4];                      // inside tile 0 / weight stage 0
#pragma unroll
    for (int aw = 0; aw < 2; ++aw)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) xo[aw][ks] = lds0 + rl0 * CD_RB + (((ks * 2 + fh) ^ (((fr & 7) + aw) & 7)) << 4);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) wo[ks] = lds0 + CD_RING + (wn * 64 + fr) * 128 + (((ks * 2 + fh) ^ ((fr >> 1) & 7)) << 4);

    f32x16 acc[2][2];                              // [nt][mt]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // Fragment sets P and Q = {x row tile 0, x row tile 1, w channel tile 0, w channel tile 1}.  The reads are inline
    // asm so that the set for k-step s+1 is in flight while the MFMAs of k-step s run (left to the scheduler, the
    // reads sink to just before their use and every k-step exposes the LDS latency); LDS returns in order, so
    // lgkmcnt(4) = "everything but the 4 reads just issued".
#define CD_LDFRAG(F, XA, XOFF, WA)                                                                                              \
    asm volatile("ds_read_b128 %0, %4 offset:%6\n\tds_read_b128 %1, %4 offset:%7\n\tds_read_b128 %2, %5\n\tds_read_b128 %3, %5 offset:4096" \
                 : "=&v"(F[0]), "=&v"(F[1]), "=&v"(F[2]), "=&v"(F[3])                                                           \
                 : "v"(XA), "v"(WA), "n"(XOFF), "n"((XOFF) + 4 * 9 * CD_RB)                                                    \
                 : "memory")
#define CD_WAITFRAG(F, N) asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(F[0]), "+v"(F[1]), "+v"(F[2]), "+v"(F[3]) : "n"(N) : "memory")
#define CD_MFMA4(F)                                                                                                             \
    do {                                                                                                                        \
        _Pragma("unroll") for (int nt = 0; nt < 2; ++nt) _Pragma("unroll") for (int mt = 0; mt < 2; ++mt)                       \
            acc[nt][mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8 *>(&F[2 + nt]),                \
                                                                  *reinterpret_cast<const bf16x8 *>(&F[mt]), acc[nt][mt], 0, 0, 0); \
        __builtin_amdgcn_sched_barrier(0);                                                                                      \
    } while (0)

    u32x4 P[4], Q[4];
    CD_LDFRAG(P, xo[0][0], 0, wo[0]);              // chunk 0, k-step 0

#pragma unroll 1
    for (int q = 0; q < 8; ++q) {
        const unsigned tsel = (q & 1) * CD_TILE, tnext = ((q + 1) & 1) * CD_TILE;
        auto tap = [&](auto a_c) {
            constexpr int A = decltype(a_c)::value;
            constexpr int AW = A & 1, TO = (((A >> 2) & 1) * 81 + ((A >> 1) & 1) * 9 + AW) * CD_RB;
            constexpr int AN = (A + 1) & 7, AWN = AN & 1, TON = (((AN >> 2) & 1) * 81 + ((AN >> 1) & 1) * 9 + AWN) * CD_RB;
            const int c = q * 8 + A;
            const unsigned wsel = (c % CD_NST) * CD_WST, wnext = ((c + 1) % CD_NST) * CD_WST;
            CD_LDFRAG(Q, xo[AW][1] + tsel, TO, wo[1] + wsel);
            CD_WAITFRAG(P, 4);
            CD_MFMA4(P);
            CD_LDFRAG(P, xo[AW][2] + tsel, TO, wo[2] + wsel);
            CD_WAITFRAG(Q, 4);
            CD_MFMA4(Q);
            CD_LDFRAG(Q, xo[AW][3] + tsel, TO, wo[3] + wsel);
            CD_WAITFRAG(P, 4);
            CD_MFMA4(P);
            CD_WAITFRAG(Q, 0);                     // every LDS read of chunk c has returned: its stage may be refilled
            // chunk c+1's weights (and, before tap 0 of the next phase, the whole next tile) have landed.  Pieces issued
            // after w(c+1): [x piece of tap A-1] w(c+2) x2; tap 7 needs the x piece of tap 6, which precedes w(c+2).
            wait_vm<(A == 0 || A == 7) ? 2 : 3>();
            __syncthreads();
            if (A < 7) issue_x(q + 1, A);
            issue_w(c + 3);
            CD_LDFRAG(P, xo[AWN][0] + (A == 7 ? tnext : tsel), TON, wo[0] + wnext);
            CD_MFMA4(Q);
        };
        tap(std::integral_constant<int, 0>{});
        tap(std::integral_constant<int, 1>{});
        tap(std::integral_constant<int, 2>{});
        tap(std::integral_constant<int, 3>{});
        tap(std::integral_constant<int, 4>{});
        tap(std::integral_constant<int, 5>{});
        tap(std::integral_constant<int, 6>{});
        tap(std::integral_constant<int, 7>{});
    }
    CD_WAITFRAG(P, 0);                              // the look-ahead reads of the non-existent chunk 64
    wait_vm<0>();                                   // trailing zero-fill pieces still target LDS
    __syncthreads();

    // ---- epilogue: lane = output (wm, mt, fr); registers walk channels
    char *stage = smem;
    // folded BN quads of this lane's 32 channels, fetched as one batch (two uniform branches, one wait)
    f32x4 scv[2][4], shv[2][4];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int g = 0; g < 4; ++g) { scv[nt][g] = f32x4{1.f, 1.f, 1.f, 1.f}; shv[nt][g] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    if (scale) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int g = 0; g < 4; ++g) scv[nt][g] = *reinterpret_cast<const f32x4 *>(scale + wn * 64 + nt * 32 + 8 * g + 4 * fh);
    }
    if (shift) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int g = 0; g < 4; ++g) shv[nt][g] = *reinterpret_cast<const f32x4 *>(shift + wn * 64 + nt * 32 + 8 * g + 4 * fh);
    }
    auto fill = [&](auto act_c, auto fp8_c) {
        constexpr int ACT = decltype(act_c)::value;
        constexpr bool FP8 = decltype(fp8_c)::value;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int c = wn * 64 + nt * 32 + 8 * g + 4 * fh;
                    const f32x4 sc = scv[nt][g], sh = shv[nt][g];
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float t = acc[nt][mt][4 * g + e] * sc[e] + sh[e];
                        if (ACT == VV_ACT_ELU) { const float em = __expf(fminf(t, 0.f)) - 1.f; t = t > 0.f ? t : em; }
                        else if (ACT == VV_ACT_RELU) t = fmaxf(t, 0.f);
                        else if (ACT == VV_ACT_LRELU) t = t > 0.f ? t : 0.3f * t;
                        v[e] = t;
                    }
                    char *dst = stage + (wm * 64 + mt * 32 + fr) * CD_SP;
                    if (FP8) {
                        *reinterpret_cast<unsigned *>(dst + c) = vv_pack_fp8x4(v);
                    } else {
                        bf16x4 o;
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = static_cast<__bf16>(v[e]);
                        *reinterpret_cast<bf16x4 *>(dst + c * 2) = o;
                    }
                }
    };
    auto with_out = [&](auto act_c) {
        if (out_fp8) fill(act_c, std::true_type{});
        else fill(act_c, std::false_type{});
    };
    switch (act) {
        case VV_ACT_ELU: with_out(std::integral_constant<int, VV_ACT_ELU>{}); break;
        case VV_ACT_RELU: with_out(std::integral_constant<int, VV_ACT_RELU>{}); break;
        case VV_ACT_LRELU: with_out(std::integral_constant<int, VV_ACT_LRELU>{}); break;
        default: with_out(std::integral_constant<int, VV_ACT_NONE>{}); break;
    }
    __syncthreads();
    const int es = out_fp8 ? 1 : 2;                   // the fp8 form hands the layer's output to an fp8 consumer (e4m3fn)
    const int cpr = CD_COUT * es / 16;                // 16-byte chunks per output row
    for (int id = tid; id < 256 * cpr; id += 512) {
        const int r = id / cpr, cc = id % cpr;
        const int od = od0 + (r >> 6), oh = oh0 + ((r >> 3) & 7), ow = ow0 + (r & 7);
        const size_t vox = ((((((size_t)b << lo) + od) << lo) + oh) << lo) + ow;
        *reinterpret_cast<uint4 *>(reinterpret_cast<char *>(y) + vox * (CD_COUT * es) + cc * 16) =
            *reinterpret_cast<const uint4 *>(stage + r * CD_SP + cc * 16);
    }
}

__global__ __launch_bounds__(512, 1) void conv_direct16_kernel(const __bf16 *__restrict__ x, const __bf16 *__restrict__ w,
                                                             const float *__restrict__ scale, const float *__restrict__ shift,
                                                             void *__restrict__ y, int dout_log2, unsigned x_bytes, unsigned w_bytes,
                                                             int act, int out_fp8) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int lo = dout_log2, no = 1 << lo, li = lo + 1, n = 1 << li;

    // XCD-aware order: consecutive boxes (same sample) run on one XCD and share its L2
    const int nwg = gridDim.x;
    int blk = (nwg & 7) == 0 ? (int)(blockIdx.x & 7) * (nwg >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    const int bxw = no >> 3, bxh = no >> 3, bxd = no >> 2;
    const int bw = blk % bxw; blk /= bxw;
    const int bh = blk % bxh; blk /= bxh;
    const int bd = blk % bxd; const int b = blk / bxd;
    const int od0 = bd * 4, oh0 = bh * 8, ow0 = bw * 8;

    const u32x4 rsx = vv_make_rsrc(x, x_bytes), rsw = vv_make_rsrc(w, w_bytes);
    const unsigned lds0 = (unsigned)(unsigned long long)(lptr_t)smem;

    // ---- producers
    auto issue_x = [&](int q, int slot) {            // slot = 0..6: piece slot*8 + wave of phase q into tile[q & 1]
        const int piece = slot * 8 + wave;
        const int rl = piece * 8 + (lane >> 3);
        const int zd = rl / 81, rem = rl - zd * 81, jh = rem / 9, jw = rem - jh * 9;
        const int g = (lane & 7) ^ ((2 * jw) & 7);
        const int id = 2 * (od0 + zd) + ((q >> 2) & 1) - 1, ih = 2 * (oh0 + jh) + ((q >> 1) & 1) - 1, iw = 2 * (ow0 + jw) + (q & 1) - 1;
        const bool ok = q < 8 && rl < 405 && (unsigned)id < (unsigned)n && (unsigned)ih < (unsigned)n && (unsigned)iw < (unsigned)n;
        const unsigned vo = ok ? (unsigned)((((((b << li) + id) << li) + ih) << li) + iw) * CD_RB + g * 16 : 0xFFFFFFF0u;
        const unsigned dst = piece < CD_PIECES ? lds0 + (q & 1) * CD_TILE + piece * 1024 : lds0 + CD_DUMMY;
        vv_dma16(rsx, vo, dst);
    };
    auto issue_w = [&](int c) {                      // chunk c = q*8 + a into ring[c % 3]: rows 16*wave .. 16*wave+15
        const int q = c >> 3, a = c & 7;
        const int td = 2 * ((a >> 2) & 1) + ((q >> 2) & 1), th = 2 * ((a >> 1) & 1) + ((q >> 1) & 1), tw = 2 * (a & 1) + (q & 1);
        const int t = (td * 4 + th) * 4 + tw;
        const unsigned st = lds0 + CD_RING + (c % CD_NST) * CD_WST;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = (wave * 2 + i) * 8 + (lane >> 3);
            const int g = (lane & 7) ^ ((row >> 1) & 7);
            const unsigned vo = c < 64 ? (unsigned)row * (64 * CD_RB) + t * CD_RB + g * 16 : 0xFFFFFFF0u;
            vv_dma16(rsw, vo, st + (wave * 2 + i) * 1024);
        }
    };

    // ---- prologue: phase 0 tile, weight chunks 0..2
#pragma unroll 1
    for (int s = 0; s < 7; ++s) issue_x(0, s);
    issue_w(0);
    issue_w(1);
    issue_w(2);
    wait_vm<0>();
    __syncthreads();

    // ---- consumer addressing (LDS byte addresses).  lane = (r, kq): row r of a 16-row fragment, k quarter kq of a 32-deep
    // k-step; a wave's 64 outputs are 4 cell tiles of two output h-rows (rows 2 ct, 2 ct + 1: +18 tile rows = +2304 B each),
    // its 64 channels 4 tiles of 16 weight rows (+2048 B each).  Slot key of a tile row: (2 jw) & 7 -- conflict-free for
    // every tap and every ds_read_b128 lane group in this fragment shape (profiles/microbench/e2_swz.py: search over all keys a jw + b jh + c zd).
    const int r = lane & 15, kq = lane >> 4;
    const int rl0 = wm * 81 + (r >> 3) * 9 + (r & 7);
    unsigned xo[2][2], wo[2];                      // [aw][k-step] inside tile 0 / [k-step] inside weight stage 0
#pragma unroll
    for (int aw = 0; aw < 2; ++aw)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) xo[aw][ks] = lds0 + rl0 * CD_RB + (((ks * 4 + kq) ^ ((2 * ((r & 7) + aw)) & 7)) << 4);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) wo[ks] = lds0 + CD_RING + (wn * 64 + r) * 128 + (((ks * 4 + kq) ^ ((r >> 1) & 7)) << 4);

    f32x4 acc[4][4];                               // [cot][ct]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

#define CD16_LD(F, XA, XOFF, WA)                                                                                                    \
    asm volatile("ds_read_b128 %0, %8 offset:%10\n\tds_read_b128 %1, %8 offset:%11\n\tds_read_b128 %2, %8 offset:%12\n\t"         \
                 "ds_read_b128 %3, %8 offset:%13\n\tds_read_b128 %4, %9\n\tds_read_b128 %5, %9 offset:2048\n\t"                   \
                 "ds_read_b128 %6, %9 offset:4096\n\tds_read_b128 %7, %9 offset:6144"                                               \
                 : "=&v"(F[0]), "=&v"(F[1]), "=&v"(F[2]), "=&v"(F[3]), "=&v"(F[4]), "=&v"(F[5]), "=&v"(F[6]), "=&v"(F[7])            \
                 : "v"(XA), "v"(WA), "n"(XOFF), "n"((XOFF) + 2304), "n"((XOFF) + 4608), "n"((XOFF) + 6912)                          \
                 : "memory")
#define CD16_WAIT(F, N)                                                                                                             \
    asm volatile("s_waitcnt lgkmcnt(%8)"                                                                                            \
                 : "+v"(F[0]), "+v"(F[1]), "+v"(F[2]), "+v"(F[3]), "+v"(F[4]), "+v"(F[5]), "+v"(F[6]), "+v"(F[7])                    \
                 : "n"(N)                                                                                                           \
                 : "memory")
#define CD16_SB __builtin_amdgcn_sched_barrier(0)
#define CD16_RD(D, A, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(D) : "v"(A), "n"(OFF) : "memory")
#define CD16_MF(F, COT, CT)                                                                                                         \
    acc[COT][CT] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8 *>(&F[4 + COT]),                          \
                                                           *reinterpret_cast<const bf16x8 *>(&F[CT]), acc[COT][CT], 0, 0, 0)

    u32x4 P[8], Q[8];
    CD16_LD(P, xo[0][0], 0, wo[0]);                // chunk 0, k-step 0

#pragma unroll 1
    for (int q = 0; q < 8; ++q) {
        const unsigned tsel = (q & 1) * CD_TILE, tnext = ((q + 1) & 1) * CD_TILE;
        auto tap = [&](auto a_c) {
            constexpr int A = decltype(a_c)::value;
            constexpr int AW = A & 1, TO = (((A >> 2) & 1) * 81 + ((A >> 1) & 1) * 9 + AW) * CD_RB;
            constexpr int AN = (A + 1) & 7, AWN = AN & 1, TON = (((AN >> 2) & 1) * 81 + ((AN >> 1) & 1) * 9 + AWN) * CD_RB;
            const int c = q * 8 + A;
            const unsigned wsel = (c % CD_NST) * CD_WST, wnext = ((c + 1) % CD_NST) * CD_WST;
            // The fragment reads of the next k-step (two per gap), the LDS-DMA pieces and their address arithmetic sit BETWEEN this
            // k-step's MFMAs, not in a burst ahead of them (convt_whole.hip: both waves of a SIMD leave the barrier in phase, and
            // a burst of memory instructions there leaves the matrix pipe idle); every LDS wait is lgkmcnt(0) at a point where
            // the reads have had 6-8 MFMAs to land.
            const unsigned xq = xo[AW][1] + tsel, wq = wo[1] + wsel;
            CD16_WAIT(P, 0);
            CD16_SB;
            CD16_MF(P, 0, 0); CD16_MF(P, 0, 1); CD16_SB; CD16_RD(Q[0], xq, TO); CD16_RD(Q[4], wq, 0); CD16_SB;
            CD16_MF(P, 0, 2); CD16_MF(P, 0, 3); CD16_SB; CD16_RD(Q[1], xq, TO + 2304); CD16_RD(Q[5], wq, 2048); CD16_SB;
            CD16_MF(P, 1, 0); CD16_MF(P, 1, 1); CD16_SB; CD16_RD(Q[2], xq, TO + 4608); CD16_RD(Q[6], wq, 4096); CD16_SB;
            CD16_MF(P, 1, 2); CD16_MF(P, 1, 3); CD16_SB; CD16_RD(Q[3], xq, TO + 6912); CD16_RD(Q[7], wq, 6144); CD16_SB;
            CD16_MF(P, 2, 0); CD16_MF(P, 2, 1); CD16_MF(P, 2, 2); CD16_MF(P, 2, 3);
            CD16_MF(P, 3, 0); CD16_MF(P, 3, 1); CD16_MF(P, 3, 2); CD16_MF(P, 3, 3);
            CD16_SB;
            CD16_WAIT(Q, 0);                       // every LDS read of chunk c has returned: its stage may be refilled
            // chunk c+1's weights (and, before tap 0 of the next phase, the whole next tile) have landed.  Pieces issued
            // after w(c+1): [x piece of tap A-1] w(c+2) x2; tap 7 needs the x piece of tap 6, which precedes w(c+2).
            wait_vm<(A == 0 || A == 7) ? 2 : 3>();
            __syncthreads();
            CD16_SB;
            CD16_MF(Q, 0, 0); CD16_MF(Q, 0, 1);
            if (A < 7) issue_x(q + 1, A);          // (the address arithmetic of the pieces: VALU beside the MFMAs around it)
            CD16_MF(Q, 0, 2); CD16_MF(Q, 0, 3);
            issue_w(c + 3);
            const unsigned xp = xo[AWN][0] + (A == 7 ? tnext : tsel), wp = wo[0] + wnext;
            CD16_SB;
            CD16_MF(Q, 1, 0); CD16_MF(Q, 1, 1); CD16_SB; CD16_RD(P[0], xp, TON); CD16_RD(P[4], wp, 0); CD16_SB;
            CD16_MF(Q, 1, 2); CD16_MF(Q, 1, 3); CD16_SB; CD16_RD(P[1], xp, TON + 2304); CD16_RD(P[5], wp, 2048); CD16_SB;
            CD16_MF(Q, 2, 0); CD16_MF(Q, 2, 1); CD16_SB; CD16_RD(P[2], xp, TON + 4608); CD16_RD(P[6], wp, 4096); CD16_SB;
            CD16_MF(Q, 2, 2); CD16_MF(Q, 2, 3); CD16_SB; CD16_RD(P[3], xp, TON + 6912); CD16_RD(P[7], wp, 6144); CD16_SB;
            CD16_MF(Q, 3, 0); CD16_MF(Q, 3, 1); CD16_MF(Q, 3, 2); CD16_MF(Q, 3, 3);
            CD16_SB;
        };
        tap(std::integral_constant<int, 0>{});
        tap(std::integral_constant<int, 1>{});
        tap(std::integral_constant<int, 2>{});
        tap(std::integral_constant<int, 3>{});
        tap(std::integral_constant<int, 4>{});
        tap(std::integral_constant<int, 5>{});
        tap(std::integral_constant<int, 6>{});
        tap(std::integral_constant<int, 7>{});
    }
    CD16_WAIT(P, 0);                                // the look-ahead reads of the non-existent chunk 64
    wait_vm<0>();                                   // trailing zero-fill pieces still target LDS
    __syncthreads();

    // ---- epilogue: lane = output (wm, mt, fr); registers walk channels
    char *stage = smem;
    // folded BN quads of this lane's 16 channels (wn*64 + 16 cot + 4 kq ..), fetched as one batch
    f32x4 scv[4], shv[4];
#pragma unroll
    for (int cot = 0; cot < 4; ++cot) { scv[cot] = f32x4{1.f, 1.f, 1.f, 1.f}; shv[cot] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    if (scale) {
#pragma unroll
        for (int cot = 0; cot < 4; ++cot) scv[cot] = *reinterpret_cast<const f32x4 *>(scale + wn * 64 + cot * 16 + 4 * kq);
    }
    if (shift) {
#pragma unroll
        for (int cot = 0; cot < 4; ++cot) shv[cot] = *reinterpret_cast<const f32x4 *>(shift + wn * 64 + cot * 16 + 4 * kq);
    }
    auto fill = [&](auto act_c, auto fp8_c) {
        constexpr int ACT = decltype(act_c)::value;
        constexpr bool FP8 = decltype(fp8_c)::value;
#pragma unroll
        for (int ct = 0; ct < 4; ++ct)
#pragma unroll
            for (int cot = 0; cot < 4; ++cot) {
                const int c = wn * 64 + cot * 16 + 4 * kq;
                const f32x4 sc = scv[cot], sh = shv[cot];
                const f32x4 v = vv_bn_act4<ACT>(acc[cot][ct], sc, sh);
                char *dst = stage + (wm * 64 + ct * 16 + r) * CD_SP;
                if (FP8) {
                    *reinterpret_cast<unsigned *>(dst + c) = vv_pack_fp8x4(v);
                } else {
                    bf16x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = static_cast<__bf16>(v[e]);
                    *reinterpret_cast<bf16x4 *>(dst + c * 2) = o;
                }
            }
    };
    auto with_out = [&](auto act_c) {
        if (out_fp8) fill(act_c, std::true_type{});
        else fill(act_c, std::false_type{});
    };
    switch (act) {
        case VV_ACT_ELU: with_out(std::integral_constant<int, VV_ACT_ELU>{}); break;
        case VV_ACT_RELU: with_out(std::integral_constant<int, VV_ACT_RELU>{}); break;
        case VV_ACT_LRELU: with_out(std::integral_constant<int, VV_ACT_LRELU>{}); break;
        default: with_out(std::integral_constant<int, VV_ACT_NONE>{}); break;
    }
    __syncthreads();
    const int es = out_fp8 ? 1 : 2;                   // the fp8 form hands the layer's output to an fp8 consumer (e4m3fn)
    const int cpr = CD_COUT * es / 16;                // 16-byte chunks per output row
    for (int id = tid; id < 256 * cpr; id += 512) {
        const int r = id / cpr, cc = id % cpr;
        const int od = od0 + (r >> 6), oh = oh0 + ((r >> 3) & 7), ow = ow0 + (r & 7);
        const size_t vox = ((((((size_t)b << lo) + od) << lo) + oh) << lo) + ow;
        *reinterpret_cast<uint4 *>(reinterpret_cast<char *>(y) + vox * (CD_COUT * es) + cc * 16) =
            *reinterpret_cast<const uint4 *>(stage + r * CD_SP + cc * 16);
    }
}


// ---- round 4: the same layer as TWO INDEPENDENT workgroups per CU (`conv_direct16h_kernel`).  conv_direct16_kernel is one 8-wave
// workgroup per CU: both waves of a SIMD belong to it, reach every per-tap barrier together and leave the matrix pipe idle together
// (50 % MFMA busy, 31 % of the wave cycles parked).  Here a workgroup is 4 waves (one per SIMD) on 256 outputs x 64 channels -- the wave
// tile, the fragment layouts, the slot keys and the per-accumulator summation order are conv_direct16_kernel's -- with ONE phase-tile
// buffer (52 KB) and 8 KB weight stages: 76 KB, so two workgroups share a CU and the second wave of every SIMD belongs to a workgroup
// with its own barriers.  What a single buffer costs -- the next phase's tile can only be fetched once every wave has read the last
// tap of this one, so the workgroup idles for one LDS-DMA round trip per phase -- is time the OTHER workgroup's waves have the SIMDs to
// themselves; the workgroups of a CU are put half a phase apart at the start (hardware wave slot parity: timing only, any placement is
// correct).  Every A tile is staged twice (once per channel half): 2 x 212 MB of L2 -> LDS traffic per launch instead of 1 x.
constexpr int CDH_COUT = 64;
constexpr int CDH_WST = CDH_COUT * 128;            // 8 KB weight stage
constexpr int CDH_RING = CD_TILE;
constexpr int CDH_DUMMY = CDH_RING + CD_NST * CDH_WST;
constexpr int CDH_LDS = CDH_DUMMY + 1024;          // 77,824 B
constexpr int CDH_SP = CDH_COUT * 2 + 16;          // epilogue row pitch (bf16; the fp8 form uses the same pitch)

__global__ __launch_bounds__(256, 2) void conv_direct16h_kernel(const __bf16 *__restrict__ x, const __bf16 *__restrict__ w,
                                                                const float *__restrict__ scale, const float *__restrict__ shift,
                                                                void *__restrict__ y, int dout_log2, unsigned x_bytes, unsigned w_bytes,
                                                                int act, int out_fp8, int stagger, int abl) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // = wm: output plane od0 + wave
    const int wm = wave;
    const int lo = dout_log2, no = 1 << lo, li = lo + 1, n = 1 << li;

    // XCD-aware order: the two channel halves of a box, then the boxes of a sample, run on one XCD (they read the same tiles)
    const int nwg = gridDim.x;
    int blk = (nwg & 7) == 0 ? (int)(blockIdx.x & 7) * (nwg >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    const int nh = blk & 1; blk >>= 1;
    const int bxw = no >> 3, bxh = no >> 3, bxd = no >> 2;
    const int bw = blk % bxw; blk /= bxw;
    const int bh = blk % bxh; blk /= bxh;
    const int bd = blk % bxd; const int b = blk / bxd;
    const int od0 = bd * 4, oh0 = bh * 8, ow0 = bw * 8;

    const u32x4 rsx = vv_make_rsrc(x, x_bytes), rsw = vv_make_rsrc(w, w_bytes);
    const unsigned lds0 = (unsigned)(unsigned long long)(lptr_t)smem;

    // ---- producers
    auto issue_x = [&](int q, int slot) {            // slot = 0..12: piece slot*4 + wave of phase q into THE tile
        const int piece = slot * 4 + wave;
        const int rl = piece * 8 + (lane >> 3);
        const int zd = rl / 81, rem = rl - zd * 81, jh = rem / 9, jw = rem - jh * 9;
        const int g = (lane & 7) ^ ((2 * jw) & 7);
        const int id = 2 * (od0 + zd) + ((q >> 2) & 1) - 1, ih = 2 * (oh0 + jh) + ((q >> 1) & 1) - 1, iw = 2 * (ow0 + jw) + (q & 1) - 1;
        const bool ok = rl < 405 && (unsigned)id < (unsigned)n && (unsigned)ih < (unsigned)n && (unsigned)iw < (unsigned)n;
        const unsigned vo = ok ? (unsigned)((((((b << li) + id) << li) + ih) << li) + iw) * CD_RB + g * 16 : 0xFFFFFFF0u;
        const unsigned dst = piece < CD_PIECES ? lds0 + piece * 1024 : lds0 + CDH_DUMMY;
        vv_dma16(rsx, vo, dst);
    };
    auto issue_w = [&](int c) {                      // chunk c = q*8 + a into ring[c % 3]: rows 16*wave .. 16*wave+15 of this half
        const int q = c >> 3, a = c & 7;
        const int td = 2 * ((a >> 2) & 1) + ((q >> 2) & 1), th = 2 * ((a >> 1) & 1) + ((q >> 1) & 1), tw = 2 * (a & 1) + (q & 1);
        const int t = (td * 4 + th) * 4 + tw;
        const unsigned st = lds0 + CDH_RING + (c % CD_NST) * CDH_WST;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = (wave * 2 + i) * 8 + (lane >> 3);
            const int g = (lane & 7) ^ ((row >> 1) & 7);
            const unsigned vo = c < 64 ? (unsigned)(nh * CDH_COUT + row) * (64 * CD_RB) + t * CD_RB + g * 16 : 0xFFFFFFF0u;
            vv_dma16(rsw, vo, st + (wave * 2 + i) * 1024);
        }
    };

    // ---- stagger: the workgroup in the odd wave slots of its SIMDs starts about half a phase later (timing only)
    {
        unsigned hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        if ((hwid & 1u) && stagger) {
            for (int i = 0; i < stagger; ++i) __builtin_amdgcn_s_sleep(8);      // 512 cycles per step
        }
    }
    // ---- prologue: phase 0 tile, weight chunks 0..2
#pragma unroll 1
    for (int s = 0; s < 13; ++s) issue_x(0, s);
    issue_w(0);
    issue_w(1);
    issue_w(2);
    wait_vm<0>();
    __syncthreads();

    // ---- consumer addressing: conv_direct16_kernel's, channel tile base 0 (the workgroup's half starts at stage row 0)
    const int r = lane & 15, kq = lane >> 4;
    const int rl0 = wm * 81 + (r >> 3) * 9 + (r & 7);
    unsigned xo[2][2], wo[2];
#pragma unroll
    for (int aw = 0; aw < 2; ++aw)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) xo[aw][ks] = lds0 + rl0 * CD_RB + (((ks * 4 + kq) ^ ((2 * ((r & 7) + aw)) & 7)) << 4);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) wo[ks] = lds0 + CDH_RING + r * 128 + (((ks * 4 + kq) ^ ((r >> 1) & 7)) << 4);

    f32x4 acc[4][4];                               // [cot][ct]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    u32x4 P[8], Q[8];
#pragma unroll 1
    for (int q = 0; q < 8; ++q) {
        if (q > 0) {
            // the refill of THE tile was issued behind the last barrier of phase q - 1 (tap 7 below); it has landed:
            wait_vm<0>();                           // (also drains the weight pieces issued before them: in-order counter)
            __syncthreads();
        }
        CD16_LD(P, xo[0][0], 0, wo[0] + ((q * 8) % CD_NST) * CDH_WST);      // chunk 8 q, k-step 0
        auto tap = [&](auto a_c) {
            constexpr int A = decltype(a_c)::value;
            constexpr int AW = A & 1, TO = (((A >> 2) & 1) * 81 + ((A >> 1) & 1) * 9 + AW) * CD_RB;
            constexpr int AN = (A + 1) & 7, AWN = AN & 1, TON = (((AN >> 2) & 1) * 81 + ((AN >> 1) & 1) * 9 + AWN) * CD_RB;
            const int c = q * 8 + A;
            const unsigned wsel = (c % CD_NST) * CDH_WST, wnext = ((c + 1) % CD_NST) * CDH_WST;
            const unsigned xq = xo[AW][1], wq = wo[1] + wsel;
            CD16_WAIT(P, 0);
            CD16_SB;
            CD16_MF(P, 0, 0); CD16_MF(P, 0, 1); CD16_SB; CD16_RD(Q[0], xq, TO); CD16_RD(Q[4], wq, 0); CD16_SB;
            CD16_MF(P, 0, 2); CD16_MF(P, 0, 3); CD16_SB; CD16_RD(Q[1], xq, TO + 2304); CD16_RD(Q[5], wq, 2048); CD16_SB;
            CD16_MF(P, 1, 0); CD16_MF(P, 1, 1); CD16_SB; CD16_RD(Q[2], xq, TO + 4608); CD16_RD(Q[6], wq, 4096); CD16_SB;
            CD16_MF(P, 1, 2); CD16_MF(P, 1, 3); CD16_SB; CD16_RD(Q[3], xq, TO + 6912); CD16_RD(Q[7], wq, 6144); CD16_SB;
            CD16_MF(P, 2, 0); CD16_MF(P, 2, 1); CD16_MF(P, 2, 2); CD16_MF(P, 2, 3);
            CD16_MF(P, 3, 0); CD16_MF(P, 3, 1); CD16_MF(P, 3, 2); CD16_MF(P, 3, 3);
            CD16_SB;
            CD16_WAIT(Q, 0);                       // every LDS read of chunk c has returned: its stage may be refilled
            // chunk c+1's weights have landed: the only pieces issued after w(c+1) are the two of w(c+2)
            wait_vm<2>();
            __syncthreads();
            CD16_SB;
            CD16_MF(Q, 0, 0); CD16_MF(Q, 0, 1);
            CD16_MF(Q, 0, 2); CD16_MF(Q, 0, 3);
            issue_w(c + 3);
            CD16_SB;
            if (A < 7) {
                const unsigned xp = xo[AWN][0], wp = wo[0] + wnext;
                CD16_MF(Q, 1, 0); CD16_MF(Q, 1, 1); CD16_SB; CD16_RD(P[0], xp, TON); CD16_RD(P[4], wp, 0); CD16_SB;
                CD16_MF(Q, 1, 2); CD16_MF(Q, 1, 3); CD16_SB; CD16_RD(P[1], xp, TON + 2304); CD16_RD(P[5], wp, 2048); CD16_SB;
                CD16_MF(Q, 2, 0); CD16_MF(Q, 2, 1); CD16_SB; CD16_RD(P[2], xp, TON + 4608); CD16_RD(P[6], wp, 4096); CD16_SB;
                CD16_MF(Q, 2, 2); CD16_MF(Q, 2, 3); CD16_SB; CD16_RD(P[3], xp, TON + 6912); CD16_RD(P[7], wp, 6144); CD16_SB;
            } else {
                // Every wave has passed this tap's barrier with all its reads of the tile returned: the tile is free, and the refill
                // for phase q + 1 goes out here, between the last MFMAs of the phase.  The next chunk's first fragments are read
                // after it has landed (top of the phase loop).
                const bool more = q < 7 && !(abl & 1);
                CD16_MF(Q, 1, 0); CD16_MF(Q, 1, 1); CD16_SB;
                if (more) { issue_x(q + 1, 0); issue_x(q + 1, 1); issue_x(q + 1, 2); }
                CD16_SB; CD16_MF(Q, 1, 2); CD16_MF(Q, 1, 3); CD16_SB;
                if (more) { issue_x(q + 1, 3); issue_x(q + 1, 4); issue_x(q + 1, 5); }
                CD16_SB; CD16_MF(Q, 2, 0); CD16_MF(Q, 2, 1); CD16_SB;
                if (more) { issue_x(q + 1, 6); issue_x(q + 1, 7); issue_x(q + 1, 8); }
                CD16_SB; CD16_MF(Q, 2, 2); CD16_MF(Q, 2, 3); CD16_SB;
                if (more) { issue_x(q + 1, 9); issue_x(q + 1, 10); issue_x(q + 1, 11); issue_x(q + 1, 12); }
                CD16_SB;
            }
            CD16_MF(Q, 3, 0); CD16_MF(Q, 3, 1); CD16_MF(Q, 3, 2); CD16_MF(Q, 3, 3);
            CD16_SB;
        };
        tap(std::integral_constant<int, 0>{});
        tap(std::integral_constant<int, 1>{});
        tap(std::integral_constant<int, 2>{});
        tap(std::integral_constant<int, 3>{});
        tap(std::integral_constant<int, 4>{});
        tap(std::integral_constant<int, 5>{});
        tap(std::integral_constant<int, 6>{});
        tap(std::integral_constant<int, 7>{});
    }
    wait_vm<0>();                                   // trailing zero-fill pieces still target LDS
    __syncthreads();

    // ---- epilogue: lane = output (wm, ct, r); registers walk channels nh*64 + 16 cot + 4 kq ..
    char *stage = smem;
    f32x4 scv[4], shv[4];
#pragma unroll
    for (int cot = 0; cot < 4; ++cot) { scv[cot] = f32x4{1.f, 1.f, 1.f, 1.f}; shv[cot] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    if (scale) {
#pragma unroll
        for (int cot = 0; cot < 4; ++cot) scv[cot] = *reinterpret_cast<const f32x4 *>(scale + nh * CDH_COUT + cot * 16 + 4 * kq);
    }
    if (shift) {
#pragma unroll
        for (int cot = 0; cot < 4; ++cot) shv[cot] = *reinterpret_cast<const f32x4 *>(shift + nh * CDH_COUT + cot * 16 + 4 * kq);
    }
    auto fill = [&](auto act_c, auto fp8_c) {
        constexpr int ACT = decltype(act_c)::value;
        constexpr bool FP8 = decltype(fp8_c)::value;
#pragma unroll
        for (int ct = 0; ct < 4; ++ct)
#pragma unroll
            for (int cot = 0; cot < 4; ++cot) {
                const int c = cot * 16 + 4 * kq;
                const f32x4 v = vv_bn_act4<ACT>(acc[cot][ct], scv[cot], shv[cot]);
                char *dst = stage + (wm * 64 + ct * 16 + r) * CDH_SP;
                if (FP8) {
                    *reinterpret_cast<unsigned *>(dst + c) = vv_pack_fp8x4(v);
                } else {
                    bf16x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = static_cast<__bf16>(v[e]);
                    *reinterpret_cast<bf16x4 *>(dst + c * 2) = o;
                }
            }
    };
    auto with_out = [&](auto act_c) {
        if (out_fp8) fill(act_c, std::true_type{});
        else fill(act_c, std::false_type{});
    };
    switch (act) {
        case VV_ACT_ELU: with_out(std::integral_constant<int, VV_ACT_ELU>{}); break;
        case VV_ACT_RELU: with_out(std::integral_constant<int, VV_ACT_RELU>{}); break;
        case VV_ACT_LRELU: with_out(std::integral_constant<int, VV_ACT_LRELU>{}); break;
        default: with_out(std::integral_constant<int, VV_ACT_NONE>{}); break;
    }
    __syncthreads();
    const int es = out_fp8 ? 1 : 2;
    const int cpr = CDH_COUT * es / 16;               // 16-byte chunks of this workgroup's half of an output row
    for (int id = tid; id < 256 * cpr; id += 256) {
        const int rr = id / cpr, cc = id % cpr;
        const int od = od0 + (rr >> 6), oh = oh0 + ((rr >> 3) & 7), ow = ow0 + (rr & 7);
        const size_t vox = ((((((size_t)b << lo) + od) << lo) + oh) << lo) + ow;
        *reinterpret_cast<uint4 *>(reinterpret_cast<char *>(y) + vox * (CD_COUT * es) + nh * CDH_COUT * es + cc * 16) =
            *reinterpret_cast<const uint4 *>(stage + rr * CDH_SP + cc * 16);
    }
}


}  // namespace

VV_EXPORT int vv_conv3d_k4s2_direct_supported(int side, int cin, int cout, int dtype) {
    return dtype == VV_BF16 && cin == CD_CIN && cout == CD_COUT && side >= 16 && vv_is_pow2(side);
}

VV_EXPORT int vv_conv3d_k4s2_direct_fwd_io(const void *x, const void *w_packed, const float *scale, const float *shift, void *y,
                                           int batch, int side, int cin, int cout, int act, int dtype, int out_dtype, void *stream) {
    if (!x || !w_packed || !y) return VV_ERR_NULL;
    if (out_dtype != VV_BF16 && out_dtype != VV_FP8) return VV_ERR_DTYPE;
    if (!vv_conv3d_k4s2_direct_supported(side, cin, cout, dtype) || batch <= 0) return VV_ERR_SHAPE;
    if (!vv_aligned16(x) || !vv_aligned16(w_packed) || !vv_aligned16(y)) return VV_ERR_ALIGN;
    const int so = side / 2;
    const int boxes = (so / 4) * (so / 8) * (so / 8);
    static const bool attr = [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_direct_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, CD_LDS);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_direct16_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, CD_LDS);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_direct16h_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, CDH_LDS);
        return true;
    }();
    (void)attr;
    const char *se = vv_hook("VV_CD_SHAPE");          // 16 = v_mfma_f32_16x16x32_bf16 (default), 32 = v_mfma_f32_32x32x16_bf16, 8 = two 4-wave workgroups per CU
    const bool s16 = !se || atoi(se) != 32;
    const bool half = se && atoi(se) == 8;
    const size_t in_per = (size_t)side * side * side * cin * 2, out_per = (size_t)so * so * so * cout * (out_dtype == VV_FP8 ? 1 : 2);
    const int per = vv_chunk_samples(in_per, batch);
    if (per < 1) return VV_ERR_SHAPE;
    for (int b0 = 0; b0 < batch; b0 += per) {
        const int nb = batch - b0 < per ? batch - b0 : per;
        const __bf16 *xc = reinterpret_cast<const __bf16 *>(reinterpret_cast<const char *>(x) + (size_t)b0 * in_per);
        void *yc = reinterpret_cast<char *>(y) + (size_t)b0 * out_per;
        if (half)
            VV_LAUNCH(conv_direct16h_kernel, dim3(nb * boxes * 2), dim3(256), CDH_LDS, reinterpret_cast<hipStream_t>(stream), xc,
                      reinterpret_cast<const __bf16 *>(w_packed), scale, shift, yc, vv_log2(so), (unsigned)((size_t)nb * in_per),
                      (unsigned)((size_t)64 * cin * cout * 2), act, out_dtype == VV_FP8 ? 1 : 0,
                      vv_hook("VV_CDH_STAGGER") ? atoi(vv_hook("VV_CDH_STAGGER")) : 5, vv_hook("VV_CDH_ABL") ? atoi(vv_hook("VV_CDH_ABL")) : 0);
        else if (s16)
            VV_LAUNCH(conv_direct16_kernel, dim3(nb * boxes), dim3(512), CD_LDS, reinterpret_cast<hipStream_t>(stream), xc,
                      reinterpret_cast<const __bf16 *>(w_packed), scale, shift, yc, vv_log2(so), (unsigned)((size_t)nb * in_per),
                      (unsigned)((size_t)64 * cin * cout * 2), act, out_dtype == VV_FP8 ? 1 : 0);
        else
            VV_LAUNCH(conv_direct_kernel, dim3(nb * boxes), dim3(512), CD_LDS, reinterpret_cast<hipStream_t>(stream), xc,
                      reinterpret_cast<const __bf16 *>(w_packed), scale, shift, yc, vv_log2(so), (unsigned)((size_t)nb * in_per),
                      (unsigned)((size_t)64 * cin * cout * 2), act, out_dtype == VV_FP8 ? 1 : 0);
        const int rc = vv_launch_status();
        if (rc != VV_OK) return rc;
    }
    return VV_OK;
}

VV_EXPORT int vv_conv3d_k4s2_direct_fwd(const void *x, const void *w_packed, const float *scale, const float *shift, void *y,
                                        int batch, int side, int cin, int cout, int act, int dtype, void *stream) {
    return vv_conv3d_k4s2_direct_fwd_io(x, w_packed, scale, shift, y, batch, side, cin, cout, act, dtype, dtype, stream);
}

// Weight gradient of the stride-2 k4 layers in phase form (bf16 operands, float32 accumulation), gfx950 only.
//
//   dW[t][ci][co] = sum over (b, o) of src[b, 2o - 1 + t, ci] * g[b, o, co]          (zero outside the grid)
//
// (src = layer input and g = dL/d(conv out) for a Conv3D -> Keras [4,4,4,cin,cout]; src = dL/d(out) and g = layer input
// for a Conv3DTranspose -> Keras [4,4,4,cout_T,cin_T].)  The reduction-GEMM form in train.hip streams the 64-tap im2col row
// of every output position through LDS: 1/32 byte per MAC at its 128 x 128 tile, and the L2 -> LDS path is what bounds it
// (0.19 ms on the 64 -> 128 layer at batch 256, against 0.11 ms for the forward convolution of the same size).  Here the
// taps are split by the parity of 2o - 1 + t per axis (t = 2a + r): for one parity r the sources of all output positions
// form the half-resolution sub-grid src[2c - 1 + r] and the 8 taps a in {0,1}^3 read cell o + a of it -- so ONE staged
// sub-grid tile (a box of 128 output positions plus a one-cell halo) feeds eight taps, and the g rows of the box feed all
// of them too: 1/133 byte per MAC.
//
// Workgroup = (parity r, 64 input channels, 128 output channels) x a range of boxes; 8 waves = the 8 taps of the parity,
// each accumulating its own 64 x 128 panel (2 x 4 MFMA tiles of v_mfma_f32_32x32x16_bf16, 128 accumulator registers) over
// the boxes.  A box is 128 consecutive output rows (b, od, oh, ow): 16 x 8 of one depth plane (S = 16), two planes (S = 8)
// or two whole samples (S = 4).  Both operands are k-strided (k = output position), so the LDS images stay row-major as
// staged by LDS-DMA -- [32-channel block][row][64 B] -- and the fragments come out of ds_read_b64_tr_b16 (see
// wgrad_bf16_kernel in train.hip for the lane mapping); every lane supplies its own row address, which is what lets the
// eight taps read the same tile at different cell offsets.  Two stages (<= 72 KB each): the DMA of box i+1 runs under the
// 64 MFMAs per wave of box i.  Partial panels go to per-split slabs, summed in split order by train.hip's reduce kernel.
#include "common.h"

namespace {

struct WgPhaseArgs {
    const void *src, *g;
    float *slabs;          // [splits][64 cin][cout]
    long rows;             // B * S^3 output positions
    int batch, li;         // li = log2(source side)
    int cin, cout;
    int nboxes, boxes_per_split, splits;
    unsigned src_bytes, g_bytes;
};

template <int LS>          // log2 of the output side S (source side 2 S)
struct WgGeo {
    static constexpr int S = 1 << LS;
    static constexpr int BW = S;                                   // a box spans whole rows
    static constexpr int BH = (128 / S) < S ? (128 / S) : S;
    static constexpr int BD = (128 / (S * BH)) < S ? (128 / (S * BH)) : S;
    static constexpr int NS = 128 / (BW * BH * BD);                // samples per box
    static_assert(BW * BH * BD * NS == 128 && BD >= 1 && NS >= 1, "box = 128 output positions");
    static constexpr int TW = BW + 1, TH = BH + 1, TD = BD + 1;    // tile = box + one halo cell per axis
    static constexpr int ROWS = NS * TD * TH * TW;
    static constexpr int NG = (ROWS + 15) / 16;                    // 16-row DMA groups per channel block
    static constexpr int XCB = NG * 1024;                          // bytes of one 32-channel block of the tile
    static constexpr int STAGE = 2 * XCB + 4 * 8192;               // tile (2 blocks) + g box (4 blocks of 128 rows)
    static constexpr int LBH = BH == 16 ? 4 : BH == 8 ? 3 : BH == 4 ? 2 : BH == 2 ? 1 : 0;
    static constexpr int LBD = BD == 16 ? 4 : BD == 8 ? 3 : BD == 4 ? 2 : BD == 2 ? 1 : 0;
    // tile row of box position k (tap offset excluded); additive over the bit fields of k
    __host__ __device__ static constexpr int rowof(int k) {
        return (((k >> (LS + LBH + LBD)) * TD + ((k >> (LS + LBH)) & (BD - 1))) * TH + ((k >> LS) & (BH - 1))) * TW + (k & (BW - 1));
    }
};

template <int LS>
__global__ __launch_bounds__(512, 1) void wgrad_phase_kernel(const WgPhaseArgs a) {
    using G = WgGeo<LS>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // XCD-major order: the 8 parities of one (channel block, split) land on the same XCD, so its L2 serves the g box to all
    const int total = gridDim.x;
    int lid = blockIdx.x;
    if ((total & 7) == 0) lid = (blockIdx.x & 7) * (total >> 3) + (blockIdx.x >> 3);
    const int ncob = a.cout >> 7, ncib = a.cin >> 6, kinds = 8 * ncib * ncob;
    const int split = lid / kinds, kind = lid - split * kinds;
    const int par = kind & 7, cb_idx = kind >> 3;
    const int cob = cb_idx % ncob, cib = cb_idx / ncob;
    const int rd = (par >> 2) & 1, rh = (par >> 1) & 1, rw = par & 1;
    const int li = a.li, n = 1 << li;
    const u32x4 rss = vv_make_rsrc(a.src, a.src_bytes), rsg = vv_make_rsrc(a.g, a.g_bytes);
    const unsigned lds0 = (unsigned)(unsigned long long)(lptr_t)smem;
    const int box0 = split * a.boxes_per_split;
    const int box1 = box0 + a.boxes_per_split < a.nboxes ? box0 + a.boxes_per_split : a.nboxes;

    // ---- staging: piece = 16 rows x 64 B of one channel block; pieces 0 .. 2 NG - 1 the tile, then 32 of the g box; wave w
    // issues pieces w, w + 8, ...  Everything about a lane's 16 bytes that does not depend on the box is computed once:
    // the byte offset relative to the box origin and the tile cell (for the padding test), so that a box costs a dozen
    // integer operations per piece instead of the divisions by 9 / 81 (the first version spent more VALU time on these
    // than the matrix pipe spent on the box).
    constexpr int NPX = 2 * G::NG, NITX = NPX / 8, NITG = 4;
    static_assert(NPX % 8 == 0, "tile pieces split evenly over the waves");
    const int prow = lane >> 2, pch = lane & 3;
    unsigned xrel[NITX];
    unsigned xcell[NITX];                                                  // zd | jh << 8 | jw << 16 | sample << 24, bit 31: past the tile
#pragma unroll
    for (int it = 0; it < NITX; ++it) {
        const int id = it * 8 + wave, cbx = id / G::NG, grp = id - cbx * G::NG;
        const int j = grp * 16 + prow;
        const int jw = j % G::TW, jh = (j / G::TW) % G::TH, zd = (j / (G::TW * G::TH)) % G::TD, sl = j / (G::TW * G::TH * G::TD);
        xrel[it] = (unsigned)((((((long)sl << li) + 2 * zd << li) + 2 * jh << li) + 2 * jw) * a.cin + cbx * 32 + pch * 8) * 2;
        xcell[it] = (unsigned)zd | (unsigned)jh << 8 | (unsigned)jw << 16 | (unsigned)sl << 24 | (j < G::ROWS ? 0u : 0x80000000u);
    }
    unsigned grel[NITG];
#pragma unroll
    for (int it = 0; it < NITG; ++it) {
        const int gid = it * 8 + wave, cbg = gid >> 3, grp = gid & 7;
        grel[it] = (unsigned)(((grp * 16 + prow) * a.cout + cob * 128 + cbg * 32 + pch * 8) * 2);
    }
    auto stage = [&](int box, int st) {
        const long r0 = (long)box * 128;
        const int oh0 = (int)((r0 >> LS) & (G::S - 1)), od0 = (int)((r0 >> (2 * LS)) & (G::S - 1));   // boxes span whole rows: ow0 = 0
        const int b0 = (int)(r0 >> (3 * LS));
        const int d0 = 2 * od0 - 1 + rd, h0 = 2 * oh0 - 1 + rh, w0 = rw - 1;                           // source cell of tile cell (0,0,0)
        const unsigned xbase = (unsigned)((((((long)b0 << li) + d0 << li) + h0 << li) + w0) * a.cin + cib * 64) * 2;
        const unsigned sbase = lds0 + st * G::STAGE;
#pragma unroll
        for (int it = 0; it < NITX; ++it) {
            const int id = it * 8 + wave, cbx = id / G::NG, grp = id - cbx * G::NG;
            const unsigned c = xcell[it];
            const int id_ = d0 + 2 * (int)(c & 255), ih = h0 + 2 * (int)((c >> 8) & 255), iw = w0 + 2 * (int)((c >> 16) & 255);
            const bool ok = (int)c >= 0 && b0 + (int)((c >> 24) & 127) < a.batch && (unsigned)id_ < (unsigned)n && (unsigned)ih < (unsigned)n &&
                            (unsigned)iw < (unsigned)n;
            vv_dma16(rss, ok ? xbase + xrel[it] : 0xFFFFFFF0u, sbase + cbx * G::XCB + grp * 1024);
        }
        const unsigned gbase = (unsigned)(r0 * a.cout * 2);
        const int rleft = (int)(a.rows - r0 < 128 ? a.rows - r0 : 128);                                // rows of the box that exist
#pragma unroll
        for (int it = 0; it < NITG; ++it) {
            const int gid = it * 8 + wave, cbg = gid >> 3, grp = gid & 7;
            vv_dma16(rsg, grp * 16 + prow < rleft ? gbase + grel[it] : 0xFFFFFFF0u, sbase + 2 * G::XCB + cbg * 8192 + grp * 1024);
        }
    };

    // ---- fragment addressing (ds_read_b64_tr_b16: lane 4q+p of a 16-lane group supplies row q / columns 4p..4p+3)
    const int g4 = lane >> 4, q4 = (lane & 15) >> 2, p4 = lane & 3;
    const int kk = (g4 >> 1) * 8 + q4;                                      // this lane's first k inside a 16-k step
    const int tap_rows = ((wave >> 2) & 1) * G::TH * G::TW + ((wave >> 1) & 1) * G::TW + (wave & 1);
    const unsigned colb = ((g4 & 1) * 16 + p4 * 4) * 2;
    const unsigned xa0 = lds0 + (G::rowof(kk) + tap_rows) * 64 + colb;       // stage 0, channel block 0
    const unsigned ga0 = lds0 + 2 * G::XCB + kk * 64 + colb;

    f32x16 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;

    constexpr int SEC = G::rowof(4) * 64;                                   // k + 4: the second half of a lane's 8 k
#define WP_LD(F, XA, GA, KS)                                                                                                      \
    asm volatile("ds_read_b64_tr_b16 %0, %12 offset:%14\n\tds_read_b64_tr_b16 %1, %12 offset:%15\n\t"                              \
                 "ds_read_b64_tr_b16 %2, %12 offset:%16\n\tds_read_b64_tr_b16 %3, %12 offset:%17\n\t"                              \
                 "ds_read_b64_tr_b16 %4, %13 offset:%18\n\tds_read_b64_tr_b16 %5, %13 offset:%19\n\t"                              \
                 "ds_read_b64_tr_b16 %6, %13 offset:%20\n\tds_read_b64_tr_b16 %7, %13 offset:%21\n\t"                              \
                 "ds_read_b64_tr_b16 %8, %13 offset:%22\n\tds_read_b64_tr_b16 %9, %13 offset:%23\n\t"                              \
                 "ds_read_b64_tr_b16 %10, %13 offset:%24\n\tds_read_b64_tr_b16 %11, %13 offset:%25"                                \
                 : "=&v"(F[0]), "=&v"(F[1]), "=&v"(F[2]), "=&v"(F[3]), "=&v"(F[4]), "=&v"(F[5]), "=&v"(F[6]), "=&v"(F[7]),          \
                   "=&v"(F[8]), "=&v"(F[9]), "=&v"(F[10]), "=&v"(F[11])                                                            \
                 : "v"(XA), "v"(GA), "n"(G::rowof((KS) * 16) * 64), "n"(G::rowof((KS) * 16) * 64 + SEC),                           \
                   "n"(G::XCB + G::rowof((KS) * 16) * 64), "n"(G::XCB + G::rowof((KS) * 16) * 64 + SEC), "n"((KS) * 1024),          \
                   "n"((KS) * 1024 + 256), "n"(8192 + (KS) * 1024), "n"(8192 + (KS) * 1024 + 256), "n"(16384 + (KS) * 1024),       \
                   "n"(16384 + (KS) * 1024 + 256), "n"(24576 + (KS) * 1024), "n"(24576 + (KS) * 1024 + 256)                        \
                 : "memory")
#define WP_WAIT(F, N)                                                                                                             \
    asm volatile("s_waitcnt lgkmcnt(%12)"                                                                                         \
                 : "+v"(F[0]), "+v"(F[1]), "+v"(F[2]), "+v"(F[3]), "+v"(F[4]), "+v"(F[5]), "+v"(F[6]), "+v"(F[7]), "+v"(F[8]),    \
                   "+v"(F[9]), "+v"(F[10]), "+v"(F[11])                                                                           \
                 : "n"(N)                                                                                                         \
                 : "memory")
#define WP_MFMA(F)                                                                                                                \
    do {                                                                                                                          \
        bf16x8 fa[2], fg[4];                                                                                                      \
        _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                                           \
            const u32x4 v = {F[2 * i][0], F[2 * i][1], F[2 * i + 1][0], F[2 * i + 1][1]};                                         \
            fa[i] = *reinterpret_cast<const bf16x8 *>(&v);                                                                        \
        }                                                                                                                         \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                                           \
            const u32x4 v = {F[4 + 2 * j][0], F[4 + 2 * j][1], F[5 + 2 * j][0], F[5 + 2 * j][1]};                                 \
            fg[j] = *reinterpret_cast<const bf16x8 *>(&v);                                                                        \
        }                                                                                                                         \
        _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int j = 0; j < 4; ++j)                               \
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fg[j], acc[i][j], 0, 0, 0);                                \
    } while (0)

    int st = 0;
    if (box0 < box1) stage(box0, 0);
#pragma unroll 1
    for (int box = box0; box < box1; ++box, st ^= 1) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                   // this box landed for everyone; the previous box's reads are done
        if (box + 1 < box1) stage(box + 1, st ^ 1);
        const unsigned xa = xa0 + st * G::STAGE, ga = ga0 + st * G::STAGE;
        u32x2 P[12], Q[12];
        WP_LD(P, xa, ga, 0);
        WP_LD(Q, xa, ga, 1);
        WP_WAIT(P, 12);
        WP_MFMA(P);
        WP_LD(P, xa, ga, 2);
        WP_WAIT(Q, 12);
        WP_MFMA(Q);
        WP_LD(Q, xa, ga, 3);
        WP_WAIT(P, 12);
        WP_MFMA(P);
        WP_LD(P, xa, ga, 4);
        WP_WAIT(Q, 12);
        WP_MFMA(Q);
        WP_LD(Q, xa, ga, 5);
        WP_WAIT(P, 12);
        WP_MFMA(P);
        WP_LD(P, xa, ga, 6);
        WP_WAIT(Q, 12);
        WP_MFMA(Q);
        WP_LD(Q, xa, ga, 7);
        WP_WAIT(P, 12);
        WP_MFMA(P);
        WP_WAIT(Q, 0);
        WP_MFMA(Q);
    }
#undef WP_LD
#undef WP_WAIT
#undef WP_MFMA

    // ---- this wave's tap panel -> slab [t][ci][co]
    const int td = 2 * ((wave >> 2) & 1) + rd, th = 2 * ((wave >> 1) & 1) + rh, tw = 2 * (wave & 1) + rw;
    const int t = (td * 4 + th) * 4 + tw;
    const int fr = lane & 31, fh = lane >> 5;
    float *slab = a.slabs + (size_t)split * 64 * a.cin * a.cout;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int nn = cob * 128 + j * 32 + fr;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int m = t * a.cin + cib * 64 + i * 32 + (q & 3) + 8 * (q >> 2) + 4 * fh;
                slab[(size_t)m * a.cout + nn] = acc[i][j][q];
            }
        }
}

struct WgPhasePlan { int nboxes, splits, bps; };
WgPhasePlan wg_phase_plan(long rows, int cin, int cout) {
    WgPhasePlan p;
    p.nboxes = (int)((rows + 127) / 128);
    const int kinds = 8 * (cin / 64) * (cout / 128);
    int splits = 1;
    while (kinds * splits * 2 <= 256 && splits * 2 <= p.nboxes) splits *= 2;     // one workgroup per CU (LDS-bound occupancy)
    p.bps = (p.nboxes + splits - 1) / splits;
    p.splits = (p.nboxes + p.bps - 1) / p.bps;
    return p;
}

template <int LS>
void wg_phase_launch(const WgPhaseArgs &a, int kinds, hipStream_t st) {
    using G = WgGeo<LS>;
    static const bool attr = [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&wgrad_phase_kernel<LS>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * G::STAGE);
        return true;
    }();
    (void)attr;
    VV_LAUNCH((wgrad_phase_kernel<LS>), dim3(kinds * a.splits), dim3(512), 2 * G::STAGE, st, a);
}

}  // namespace

bool vv_wgrad_phase_ok(const void *src, const void *g, int batch, int side, int cin, int cout) {
    if (vv_hook("VV_NO_WGRAD_PHASE") || vv_hook("VV_WGRAD_F32")) return false;
    const int S = side / 2;
    if (S != 4 && S != 8 && S != 16) return false;
    if (cin % 64 || cout % 128) return false;
    if ((size_t)batch * side * side * side * cin * 2 >= 0xFFFFFFF0ull || (size_t)batch * S * S * S * cout * 2 >= 0xFFFFFFF0ull) return false;
    if ((long)batch * S * S * S < 128 * 8) return false;          // tiny problems: the reduction-GEMM form splits finer
    return vv_aligned16(src) && vv_aligned16(g);
}

// Slab bytes for `rows` output positions (the plan depends on the row count and the channel blocks only).
size_t vv_wgrad_phase_ws(long rows, int cin, int cout) {
    if (cin <= 0 || cout <= 0 || cin % 64 || cout % 128) return 0;
    return (size_t)wg_phase_plan(rows, cin, cout).splits * 64 * cin * cout * sizeof(float);
}

// Launches the phase kernel; the caller sums the *splits slabs [64 cin][cout].
void vv_wgrad_phase_launch(const void *src, const void *g, float *slabs, int batch, int side, int cin, int cout, int *splits,
                           hipStream_t st) {
    const int S = side / 2;
    const WgPhasePlan p = wg_phase_plan((long)batch * S * S * S, cin, cout);
    WgPhaseArgs a{src, g, slabs, (long)batch * S * S * S, batch, vv_log2(side), cin, cout, p.nboxes, p.bps, p.splits,
                  (unsigned)((size_t)batch * side * side * side * cin * 2), (unsigned)((size_t)batch * S * S * S * cout * 2)};
    const int kinds = 8 * (cin / 64) * (cout / 128);
    if (S == 16) wg_phase_launch<4>(a, kinds, st);
    else if (S == 8) wg_phase_launch<3>(a, kinds, st);
    else wg_phase_launch<2>(a, kinds, st);
    *splits = p.splits;
}

// Direct stride-2 convolution in phase form on fp8 (OCP e4m3fn) operands, block-scaled K = 64 MFMA (gfx950).
// The fp8 twin of conv_direct.hip for the widest encoder layer (Cin 64 -> Cout 128, input side >= 16): the same 8 input
// phases x 8 taps over per-phase 5 x 9 x 9 sub-grid tiles of a 4 x 8 x 8 box of outputs, double-buffered tiles, 3-deep
// weight ring by LDS-DMA, one barrier per chunk -- with
//   * 64-byte voxel rows: 4 slots of 16 B; four rows share the 64 banks, and the four 4-row segments of a ds_read_b128
//     lane group sit in four different tile rows jh, so the slot swizzle is jh & 3;
//   * one K = 64 MFMA step per tap (the whole Cin), so a chunk is a PAIR of taps (aw = 0, 1): 8 MFMAs of 64 cycles per
//     wave between barriers -- the cadence of the bf16 kernel's 16 MFMAs of 32 cycles -- and a weight stage holds the two
//     taps of the pair side by side in 128-byte rows (swizzle (row >> 1) & 7 as in the bf16 kernel);
//   * operands of 32 bytes per lane: two adjacent slots, i.e. address and address ^ 16.
// Output e4m3fn (for an fp8 consumer) or bf16.  Per-output-channel weight scales ride in `scale`.
#include <type_traits>

#include "common.h"

namespace {

typedef __attribute__((ext_vector_type(8))) int i32x8;

template <int N>
__device__ __forceinline__ void wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" : : "n"(N) : "memory");
}

constexpr int F8_CIN = 64, F8_COUT = 128;
constexpr int F8_RB = F8_CIN;                     // bytes per voxel row
constexpr int F8_PIECES = 26;                     // 405 rows in 1 KiB pieces of 16 rows
constexpr int F8_TILE = F8_PIECES * 1024;         // 26,624 B
constexpr int F8_WST = F8_COUT * 128;             // one weight stage: 128 rows x (2 taps x 64 k)
constexpr int F8_NST = 3;
constexpr int F8_RING = 2 * F8_TILE;
constexpr int F8_DUMMY = F8_RING + F8_NST * F8_WST;
constexpr int F8_LDS = F8_DUMMY + 1024;           // 103,424 B
constexpr int F8_SP = F8_COUT * 2 + 16;           // epilogue row pitch

__global__ __launch_bounds__(512, 1) void conv_direct_fp8_kernel(const unsigned char *__restrict__ x, const unsigned char *__restrict__ w,
                                                                 const float *__restrict__ scale, const float *__restrict__ shift,
                                                                 void *__restrict__ y, int dout_log2, unsigned x_bytes, unsigned w_bytes,
                                                                 int act, int out_fp8) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int lo = dout_log2, no = 1 << lo, li = lo + 1, n = 1 << li;

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
    auto issue_x = [&](int q, int slot) {            // slot = 0..3: piece slot*8 + wave of phase q into tile[q & 1]
        const int ln = lane;
        const int piece = slot * 8 + wave;
        const int rl = piece * 16 + (ln >> 2);
        const int zd = rl / 81, rem = rl - zd * 81, jh = rem / 9, jw = rem - jh * 9;
        const int g = (ln & 3) ^ (jh & 3);
        const int id = 2 * (od0 + zd) + ((q >> 2) & 1) - 1, ih = 2 * (oh0 + jh) + ((q >> 1) & 1) - 1, iw = 2 * (ow0 + jw) + (q & 1) - 1;
        const bool ok = q < 8 && rl < 405 && (unsigned)id < (unsigned)n && (unsigned)ih < (unsigned)n && (unsigned)iw < (unsigned)n;
        const unsigned vo = ok ? (unsigned)((((((b << li) + id) << li) + ih) << li) + iw) * F8_RB + g * 16 : 0xFFFFFFF0u;
        const unsigned dst = piece < F8_PIECES ? lds0 + (q & 1) * F8_TILE + piece * 1024 : lds0 + F8_DUMMY;
        vv_dma16(rsx, vo, dst);
    };
    auto issue_w = [&](int c) {                      // chunk c = q*4 + j (taps a = 2j, 2j+1) into ring[c % 3]: rows 16*wave .. 16*wave+15
        const int q = c >> 2, j = c & 3;
        const unsigned st = lds0 + F8_RING + (c % F8_NST) * F8_WST;
        const int ln = lane;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = (wave * 2 + i) * 8 + (ln >> 3);
            const int g = (ln & 7) ^ ((row >> 1) & 7);
            const int a = 2 * j + (g >> 2);
            const int td = 2 * ((a >> 2) & 1) + ((q >> 2) & 1), th = 2 * ((a >> 1) & 1) + ((q >> 1) & 1), tw = 2 * (a & 1) + (q & 1);
            const int t = (td * 4 + th) * 4 + tw;
            const unsigned vo = c < 32 ? (unsigned)row * (64 * F8_RB) + t * F8_RB + (g & 3) * 16 : 0xFFFFFFF0u;
            vv_dma16(rsw, vo, st + (wave * 2 + i) * 1024);
        }
    };

    // ---- prologue: phase 0 tile, weight chunks 0..2
#pragma unroll 1
    for (int s = 0; s < 4; ++s) issue_x(0, s);
    issue_w(0);
    issue_w(1);
    issue_w(2);
    wait_vm<0>();
    __syncthreads();

    // ---- consumer addressing (LDS byte addresses); the second 16 bytes of a 32-byte operand sit at address ^ 16
    const int fr = lane & 31, fh = lane >> 5;
    const int rl0 = wm * 81 + (fr >> 3) * 9 + (fr & 7);
    unsigned xo[2], wo[2];                          // [ah] inside tile 0 / [tap of the pair] inside weight stage 0
#pragma unroll
    for (int ah = 0; ah < 2; ++ah) xo[ah] = lds0 + rl0 * F8_RB + (((fh * 2) ^ (((fr >> 3) + ah) & 3)) << 4);
    wo[0] = lds0 + F8_RING + (wn * 64 + fr) * 128 + (((fh * 2) ^ ((fr >> 1) & 7)) << 4);
    wo[1] = wo[0] ^ 64u;

    f32x16 acc[2][2];                              // [nt][mt]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // Fragment sets P and Q = {x row tile 0 (lo, hi), x row tile 1 (lo, hi), w channel tile 0 (lo, hi), w channel tile 1 (lo, hi)}:
    // 32-byte MFMA operands read as two 16-byte halves (address, address ^ 16) by inline asm, the set of the next tap in
    // flight while the MFMAs of the current tap run.  The MFMAs are pinned in place by an empty asm that "modifies" the
    // accumulators: they are pure, and left alone the compiler parked the fragments in scratch and ran all 32 MFMAs of a
    // phase at the top of the next loop trip (256 VGPRs, 364 B of spills in the loop, 0.142 ms instead of 0.08).
#define F8_LDFRAG(F, XA, XOFF, WA)                                                                                              \
    asm volatile("ds_read_b128 %0, %8 offset:%12\n\tds_read_b128 %1, %9 offset:%12\n\t"                                         \
                 "ds_read_b128 %2, %8 offset:%13\n\tds_read_b128 %3, %9 offset:%13\n\t"                                         \
                 "ds_read_b128 %4, %10\n\tds_read_b128 %5, %11\n\t"                                                             \
                 "ds_read_b128 %6, %10 offset:4096\n\tds_read_b128 %7, %11 offset:4096"                                          \
                 : "=&v"(F[0]), "=&v"(F[1]), "=&v"(F[2]), "=&v"(F[3]), "=&v"(F[4]), "=&v"(F[5]), "=&v"(F[6]), "=&v"(F[7])       \
                 : "v"(XA), "v"((XA) ^ 16u), "v"(WA), "v"((WA) ^ 16u), "n"(XOFF), "n"((XOFF) + 4 * 9 * F8_RB)                  \
                 : "memory")
#define F8_WAITFRAG(F, N)                                                                                                       \
    asm volatile("s_waitcnt lgkmcnt(%8)"                                                                                        \
                 : "+v"(F[0]), "+v"(F[1]), "+v"(F[2]), "+v"(F[3]), "+v"(F[4]), "+v"(F[5]), "+v"(F[6]), "+v"(F[7])               \
                 : "n"(N)                                                                                                       \
                 : "memory")

    // one MFMA (weight tile NT x cell tile MT) pinned in place, and one 32-byte operand (two ds_read_b128) of a later tap: the
    // reads of the next tap sit between the MFMAs of the current one instead of in a burst ahead of them (conv_direct.hip)
#define F8_MF(F, NT, MT)                                                                                                        \
    do {                                                                                                                        \
        const i32x8 wv = {(int)F[4 + 2 * NT][0], (int)F[4 + 2 * NT][1], (int)F[4 + 2 * NT][2], (int)F[4 + 2 * NT][3],           \
                          (int)F[5 + 2 * NT][0], (int)F[5 + 2 * NT][1], (int)F[5 + 2 * NT][2], (int)F[5 + 2 * NT][3]};          \
        const i32x8 xv = {(int)F[2 * MT][0], (int)F[2 * MT][1], (int)F[2 * MT][2], (int)F[2 * MT][3],                           \
                          (int)F[2 * MT + 1][0], (int)F[2 * MT + 1][1], (int)F[2 * MT + 1][2], (int)F[2 * MT + 1][3]};          \
        acc[NT][MT] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(wv, xv, acc[NT][MT], 0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F); \
        asm volatile("" : "+v"(acc[NT][MT]));                                                                                   \
        __builtin_amdgcn_sched_barrier(0);                                                                                      \
    } while (0)
#define F8_RD2(FA, FB, A, OFF)                                                                                                  \
    do {                                                                                                                        \
        asm volatile("ds_read_b128 %0, %2 offset:%4\n\tds_read_b128 %1, %3 offset:%4"                                          \
                     : "=&v"(FA), "=&v"(FB) : "v"(A), "v"((A) ^ 16u), "n"(OFF) : "memory");                                     \
        __builtin_amdgcn_sched_barrier(0);                                                                                      \
    } while (0)

    u32x4 P[8], Q[8];
    F8_LDFRAG(P, xo[0], 0, wo[0]);                 // chunk 0, tap 0

#pragma unroll 1
    for (int q = 0; q < 8; ++q) {
        const unsigned tsel = (q & 1) * F8_TILE, tnext = ((q + 1) & 1) * F8_TILE;
        auto chunk = [&](auto j_c) {
            constexpr int J = decltype(j_c)::value;
            constexpr int AD = (J >> 1) & 1, AH = J & 1;
            constexpr int TO1 = (AD * 81 + AH * 9 + 1) * F8_RB;                       // second tap of the pair (aw = 1)
            constexpr int JN = (J + 1) & 3, ADN = (JN >> 1) & 1, AHN = JN & 1;
            constexpr int TON = (ADN * 81 + AHN * 9) * F8_RB;                         // first tap of the next pair (aw = 0)
            const int c = q * 4 + J;
            const unsigned wsel = (c % F8_NST) * F8_WST, wnext = ((c + 1) % F8_NST) * F8_WST;
            const unsigned xcur = xo[AH] + tsel, wcur = wo[1] + wsel;
            F8_WAITFRAG(P, 0);
            __builtin_amdgcn_sched_barrier(0);
            F8_MF(P, 0, 0);                        // first tap of the pair
            F8_RD2(Q[0], Q[1], xcur, TO1); F8_RD2(Q[4], Q[5], wcur, 0);
            F8_MF(P, 0, 1);
            F8_RD2(Q[2], Q[3], xcur, TO1 + 4 * 9 * F8_RB); F8_RD2(Q[6], Q[7], wcur, 4096);
            F8_MF(P, 1, 0);
            F8_MF(P, 1, 1);
            F8_WAITFRAG(Q, 0);                     // every LDS read of chunk c has returned: its stage may be refilled
            // chunk c+1's weights (and, before the first tap of the next phase, the whole next tile) have landed.  Issued after
            // w(c+1): [the tile pieces of the previous chunk's slot: 2, 1, 1, 0 for J = 0..3] w(c+2) x2; chunk 3 also needs the
            // piece of chunk 2, which precedes w(c+2).
            wait_vm<(J == 1) ? 4 : (J == 2 ? 3 : 2)>();
            __syncthreads();
            __builtin_amdgcn_sched_barrier(0);
            F8_MF(Q, 0, 0);                        // second tap
            if (J == 0) { issue_x(q + 1, 0); issue_x(q + 1, 1); }
            else if (J == 1) issue_x(q + 1, 2);
            else if (J == 2) issue_x(q + 1, 3);
            issue_w(c + 3);
            const unsigned xnext = xo[AHN] + (J == 3 ? tnext : tsel), wnx = wo[0] + wnext;
            __builtin_amdgcn_sched_barrier(0);
            F8_MF(Q, 0, 1);
            F8_RD2(P[0], P[1], xnext, TON); F8_RD2(P[4], P[5], wnx, 0);
            F8_MF(Q, 1, 0);
            F8_RD2(P[2], P[3], xnext, TON + 4 * 9 * F8_RB); F8_RD2(P[6], P[7], wnx, 4096);
            F8_MF(Q, 1, 1);
        };
        chunk(std::integral_constant<int, 0>{});
        chunk(std::integral_constant<int, 1>{});
        chunk(std::integral_constant<int, 2>{});
        chunk(std::integral_constant<int, 3>{});
    }
    F8_WAITFRAG(P, 0);                              // the look-ahead reads of the non-existent chunk 32
    wait_vm<0>();                                   // trailing zero-fill pieces still target LDS
    __syncthreads();
#undef F8_LDFRAG
#undef F8_MF
#undef F8_RD2
#undef F8_WAITFRAG

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
                    char *dst = stage + (wm * 64 + mt * 32 + fr) * F8_SP;
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
    const int cpr = F8_COUT * es / 16;                // 16-byte chunks per output row
    for (int id = tid; id < 256 * cpr; id += 512) {
        const int r = id / cpr, cc = id % cpr;
        const int od = od0 + (r >> 6), oh = oh0 + ((r >> 3) & 7), ow = ow0 + (r & 7);
        const size_t vox = ((((((size_t)b << lo) + od) << lo) + oh) << lo) + ow;
        *reinterpret_cast<uint4 *>(reinterpret_cast<char *>(y) + vox * (F8_COUT * es) + cc * 16) =
            *reinterpret_cast<const uint4 *>(stage + r * F8_SP + cc * 16);
    }
}

}  // namespace

VV_EXPORT int vv_conv3d_k4s2_direct_fp8_supported(int side, int cin, int cout) {
    return cin == F8_CIN && cout == F8_COUT && side >= 16 && vv_is_pow2(side);
}

/* x [B][side^3][64] e4m3fn, w_packed = vv_pack_conv_k4(dtype VV_FP8) [128][64*64] e4m3fn; y e4m3fn or bf16. */
VV_EXPORT int vv_conv3d_k4s2_direct_fp8_fwd(const void *x, const void *w_packed, const float *scale, const float *shift, void *y,
                                            int batch, int side, int cin, int cout, int act, int out_dtype, void *stream) {
    if (!x || !w_packed || !y) return VV_ERR_NULL;
    if (out_dtype != VV_BF16 && out_dtype != VV_FP8) return VV_ERR_DTYPE;
    if (!vv_conv3d_k4s2_direct_fp8_supported(side, cin, cout) || batch <= 0) return VV_ERR_SHAPE;
    if (!vv_aligned16(x) || !vv_aligned16(w_packed) || !vv_aligned16(y)) return VV_ERR_ALIGN;
    const size_t xb = (size_t)batch * side * side * side * cin;
    if (xb >= 0xFFFFFFF0ull) return VV_ERR_SHAPE;
    const int so = side / 2;
    const int boxes = (so / 4) * (so / 8) * (so / 8);
    static const bool attr = [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_direct_fp8_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, F8_LDS);
        return true;
    }();
    (void)attr;
    VV_LAUNCH(conv_direct_fp8_kernel, dim3(batch * boxes), dim3(512), F8_LDS, reinterpret_cast<hipStream_t>(stream),
              reinterpret_cast<const unsigned char *>(x), reinterpret_cast<const unsigned char *>(w_packed), scale, shift, y,
              vv_log2(so), (unsigned)xb, (unsigned)((size_t)64 * cin * cout), act, out_dtype == VV_FP8 ? 1 : 0);
    return vv_launch_status();
}

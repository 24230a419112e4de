// Direct transposed convolution on MFMA with the input tile resident in LDS (bf16, gfx950).
//
// Conv3DTranspose k4 s2 SAME (autoencoder3D.py:41-54) as 8 output-parity sub-convolutions reads every input voxel
// 64 times (8 parities x 8 taps).  The implicit-GEMM kernel re-fetches those rows from L2 for every (parity, tap) and
// is bound by L2 -> CU bytes (96 B/clk/CU at full MFMA rate for a 128 x 64 tile).  Here one workgroup stages the
// (4+2) x (4+2) x (8+2) halo tile of its 4 x 4 x 8 block of input cells ONCE (LDS-DMA, zeros outside the grid), and all
// 8 parities x 8 taps read their A fragments from it; the weights are streamed straight into registers from a
// fragment-ordered panel (one fully coalesced 1 KiB load per MFMA operand, each byte used once per workgroup).
//
//   LDS tile   : 360 voxels x CIN bf16, 16-byte slots XOR-swizzled with (zw + 8 zh) & 15 -- conflict-free for every
//                tap and both ds_read_b128 lane groups (exhaustively checked offline)
//   waves      : default 8 (one output parity each) over a 4 x 4 x 8 block of cells: 4 row tiles (32 cells each) x COUT/32
//                channel tiles of 32x32 accumulators per wave, so every streamed weight fragment feeds 4 MFMAs; the 4-wave
//                forms (2 parities per wave, 2 x 4 x 8 or 4 x 4 x 8 cells) remain selectable (VV_DIRECT_MT = 2 / 4);
//                weights-first MFMA, lane = cell, registers walk channels
//   epilogue   : folded BN + activation on float4 quads, wave-private LDS transpose, 16-byte stores of whole channel rows
#include <stdlib.h>

#include <type_traits>

#include "common.h"

namespace {

// out[p][a][ks][nt][lane][j] = w[t(p,a)][co = nt*32 + (lane&31)][ci = ks*16 + 8*(lane>>5) + j]
__global__ void pack_convT_frag_kernel(const float *__restrict__ w, __bf16 *__restrict__ out, int cin, int cout) {
    const int KS = cin / 16, NT = cout / 32;
    const long total = (long)8 * 8 * KS * NT * 64 * 8;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int j = (int)(i & 7), lane = (int)((i >> 3) & 63);
        long r = i >> 9;
        const int nt = (int)(r % NT); r /= NT;
        const int ks = (int)(r % KS); r /= KS;
        const int a = (int)(r & 7), p = (int)(r >> 3);
        const int td = 1 - ((p >> 2) & 1) + 2 * ((a >> 2) & 1);
        const int th = 1 - ((p >> 1) & 1) + 2 * ((a >> 1) & 1);
        const int tw = 1 - (p & 1) + 2 * (a & 1);
        const int t = (td * 4 + th) * 4 + tw;
        const int co = nt * 32 + (lane & 31), ci = ks * 16 + 8 * (lane >> 5) + j;
        out[i] = static_cast<__bf16>(w[((size_t)t * cout + co) * cin + ci]);
    }
}

constexpr int HH = 6, HW = 10;   // halo tile of an MT x 4 x 8 block of cells: (MT+2) x 6 x 10 voxels

template <int CIN, int COUT, int MT, int NW>      // NW waves per workgroup; a wave owns 8 / NW output parities
__global__ __launch_bounds__(NW * 64, (NW == 8 || MT == 4) ? 1 : 2) void convT_direct_kernel(const __bf16 *__restrict__ x, const __bf16 *__restrict__ wf,
                                                              const float *__restrict__ scale, const float *__restrict__ shift,
                                                              __bf16 *__restrict__ y, int din_log2, unsigned x_bytes, int act) {
    constexpr int RB = CIN * 2;          // bytes per voxel row
    constexpr int SPR = RB / 16;         // 16-byte slots per row (16 for CIN = 128)
    constexpr int KS = CIN / 16;         // MFMA k-steps per tap
    constexpr int NT = COUT / 32;        // channel tiles
    constexpr int HALF = (MT == 4 && NW == 8) ? 1 : 2;   // k-steps per weight prefetch group (4 groups in the register ring)
    constexpr int GPT = KS / HALF;       // groups per tap
    constexpr int GPP = 8 * GPT;         // groups per parity
    constexpr int SPITCH = COUT * 2 + 16;
    constexpr int HV = (MT + 2) * HH * HW;
    static_assert(SPR == 16, "the slot swizzle is derived for 256-byte voxel rows");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *tile = smem;                                   // [360][RB]
    char *stage = smem + HV * RB;                        // [4 waves][32][SPITCH]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = din_log2, n = 1 << li;
    const int bxw = n >> 3, bxh = n >> 2, bxd = n / MT;  // boxes along w / h / d
    int blk = blockIdx.x;
    const int bw = blk % bxw; blk /= bxw;
    const int bh = blk % bxh; blk /= bxh;
    const int bd = blk % bxd; const int b = blk / bxd;
    const int d0 = bd * MT, h0 = bh * 4, w0 = bw * 8;

    // ---- stage the halo tile: 360 voxels x 16 slots = 90 wave instructions of 1 KiB (4 voxels each)
    {
        const u32x4 rs = vv_make_rsrc(x, x_bytes);
        const unsigned lds0 = (unsigned)(unsigned long long)(lptr_t)tile;
        const int pos = lane & 15, vsub = lane >> 4;
        for (int it = wave; it < HV / 4; it += NW) {
            const int v = it * 4 + vsub;
            const int zw = v % HW, zh = (v / HW) % HH, zd = v / (HW * HH);
            const int id = d0 - 1 + zd, ih = h0 - 1 + zh, iw = w0 - 1 + zw;
            const bool ok = (unsigned)id < (unsigned)n && (unsigned)ih < (unsigned)n && (unsigned)iw < (unsigned)n;
            const int g = pos ^ ((zw + 8 * zh) & 15);
            const unsigned vo = ok ? (unsigned)((((((b << li) + id) << li) + ih) << li) + iw) * RB + g * 16 : 0xFFFFFFF0u;
            vv_dma16(rs, vo, lds0 + it * 1024);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();

    const int fr = lane & 31, fh = lane >> 5;
    const int mh = fr >> 3, mw = fr & 7;
    const uint4 *wfl = reinterpret_cast<const uint4 *>(wf) + lane;      // + ((((p*8 + a)*KS + ks)*NT + nt) * 64)
    char *mystage = stage + wave * 32 * SPITCH;
    const int lo = li + 1;

    // Weights: this wave's 2 parities x 16 half-tap groups are 32 consecutive groups of the fragment panel.  They are
    // streamed through a 4-deep register ring (3 groups = 96 MFMAs ~ 3 k cycles of prefetch distance: with one wave per
    // SIMD nothing else hides the L2 latency), running ahead across the parity boundary and its epilogue.
    constexpr int GL = HALF * NT;                          // 16-byte loads per lane per group
    constexpr int PPW = 8 / NW;                            // parities per wave
    const uint4 *wp = wfl + (size_t)(wave * PPW * GPP) * GL * 64;
    uint4 b0[GL], b1[GL], b2[GL], b3[GL];
    auto load_group = [&](int G, uint4 *dst) {
        if (G < PPW * GPP) {
            const uint4 *src = wp + (size_t)G * GL * 64;
#pragma unroll
            for (int i = 0; i < GL; ++i) dst[i] = src[(size_t)i * 64];
        }
    };
    load_group(0, b0);
    load_group(1, b1);
    load_group(2, b2);

    auto run_parity = [&](auto act_c, int pi) {
        constexpr int ACT = decltype(act_c)::value;
        const int p = wave * PPW + pi, pd = (p >> 2) & 1, ph = (p >> 1) & 1, pw = p & 1;
        f32x16 acc[MT][NT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int q = 0; q < 16; ++q) acc[mt][nt][q] = 0.f;

        // A fragments come from the resident tile by inline-asm ds_read_b128, ONE k-step ahead of the MFMAs that use
        // them (set P/Q ping-pong): left to the scheduler the reads sink to just before their use and every k-step
        // exposes the LDS latency.  LDS returns in order, so lgkmcnt(MT) = "all but the MT reads just issued".
        const unsigned tile_lds = (unsigned)(unsigned long long)(lptr_t)tile;
        auto a_addr = [&](int a, int ks) -> unsigned {
            const int ad = (a >> 2) & 1, ah = (a >> 1) & 1, aw = a & 1;
            const int zh = mh + ph - ah + 1, zw = mw + pw - aw + 1;
            const int sw = (zw + 8 * zh) & 15;
            return tile_lds + (((pd - ad + 1) * HH + zh) * HW + zw) * RB + (((ks * 2 + fh) ^ sw) << 4);
        };
        u32x4 fP[MT], fQ[MT];
        auto ld = [&](u32x4 *F, unsigned addr) {
            if constexpr (MT == 2)
                asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:%3"
                             : "=&v"(F[0]), "=&v"(F[1]) : "v"(addr), "n"(HH * HW * RB) : "memory");
            else
                asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:%5\n\tds_read_b128 %2, %4 offset:%6\n\tds_read_b128 %3, %4 offset:%7"
                             : "=&v"(F[0]), "=&v"(F[1]), "=&v"(F[2]), "=&v"(F[3])
                             : "v"(addr), "n"(HH * HW * RB), "n"(2 * HH * HW * RB), "n"(3 * HH * HW * RB) : "memory");
        };
        auto wait = [&](u32x4 *F) {
            if constexpr (MT == 2) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(F[0]), "+v"(F[1]) : : "memory");
            else asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(F[0]), "+v"(F[1]), "+v"(F[2]), "+v"(F[3]) : : "memory");
        };
        auto mma = [&](const u32x4 *F, const uint4 *bfk) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8 *>(&bfk[nt]),
                                                                          *reinterpret_cast<const bf16x8 *>(&F[mt]), acc[mt][nt], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);     // (pinning the MFMAs with an asm on the accumulators, as the fp8 twin does, is neutral
                                                   // for the 8-wave form and 2.5x slower for the 4-wave MT = 4 form: not used here)
        };
        // one group = HALF k-steps of one tap with the weights of ring buffer bf; k-steps alternate P, Q (HALF is even)
        constexpr bool PINGPONG = !(MT == 4 && NW == 8);    // the 8-wave form has no registers left for a second set (300 B of spills when forced)
        static_assert(!PINGPONG || HALF % 2 == 0, "P/Q roles must line up across groups");
        auto compute_group = [&](int g, const uint4 *bf) {
            const int a = g / GPT, part = g % GPT;
            if constexpr (PINGPONG) {
#pragma unroll
                for (int k = 0; k < HALF; k += 2) {
                    const int ks = part * HALF + k;
                    ld(fQ, a_addr(a, ks + 1));
                    wait(fP);
                    mma(fP, bf + k * NT);
                    // the k-step after next: same tap, or the first of the next tap (wraps harmlessly after the last one)
                    const int ks2 = ks + 2;
                    ld(fP, ks2 < KS ? a_addr(a, ks2) : a_addr((a + 1) & 7, 0));
                    wait(fQ);
                    mma(fQ, bf + (k + 1) * NT);
                }
            } else {
#pragma unroll
                for (int k = 0; k < HALF; ++k) {
                    const int ks = part * HALF + k;
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fP[0]), "+v"(fP[1]), "+v"(fP[MT - 2]), "+v"(fP[MT - 1]) : : "memory");
                    mma(fP, bf + k * NT);
                    ld(fP, ks + 1 < KS ? a_addr(a, ks + 1) : a_addr((a + 1) & 7, 0));
                }
            }
        };
        ld(fP, a_addr(0, 0));
#pragma unroll 1
        for (int g = 0; g < GPP; g += 4) {
            const int G = pi * GPP + g;
            load_group(G + 3, b3); compute_group(g, b0);
            load_group(G + 4, b0); compute_group(g + 1, b1);
            load_group(G + 5, b1); compute_group(g + 2, b2);
            load_group(G + 6, b2); compute_group(g + 3, b3);
        }
        if constexpr (MT == 2) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fP[0]), "+v"(fP[1]) : : "memory");   // the wrapped look-ahead read
        else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fP[0]), "+v"(fP[1]), "+v"(fP[2]), "+v"(fP[3]) : : "memory");

        // ---- epilogue of this parity: lane = cell (mt, mh, mw), registers walk channels.  The folded BN quads are
        // fetched as one batch (two uniform branches, one wait) rather than quad by quad.
        f32x4 scv[NT][4], shv[NT][4];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int g = 0; g < 4; ++g) { scv[nt][g] = f32x4{1.f, 1.f, 1.f, 1.f}; shv[nt][g] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        if (scale) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int g = 0; g < 4; ++g) scv[nt][g] = *reinterpret_cast<const f32x4 *>(scale + nt * 32 + 8 * g + 4 * fh);
        }
        if (shift) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int g = 0; g < 4; ++g) shv[nt][g] = *reinterpret_cast<const f32x4 *>(shift + nt * 32 + 8 * g + 4 * fh);
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int c = nt * 32 + 8 * g + 4 * fh;
                    const f32x4 sc = scv[nt][g], sh = shv[nt][g];
                    bf16x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float t = acc[mt][nt][4 * g + e] * sc[e] + sh[e];
                        if (ACT == VV_ACT_ELU) { const float em = __expf(fminf(t, 0.f)) - 1.f; t = t > 0.f ? t : em; }
                        else if (ACT == VV_ACT_RELU) t = fmaxf(t, 0.f);
                        else if (ACT == VV_ACT_LRELU) t = t > 0.f ? t : 0.3f * t;
                        o[e] = static_cast<__bf16>(t);
                    }
                    *reinterpret_cast<bf16x4 *>(mystage + fr * SPITCH + c * 2) = o;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            constexpr int CPR = COUT * 2 / 16;            // 16-byte chunks per output row
#pragma unroll
            for (int i = 0; i < 32 * CPR / 64; ++i) {
                const int id = lane + 64 * i, r = id / CPR, c = id % CPR;
                const int od = 2 * (d0 + mt) + pd, oh = 2 * (h0 + (r >> 3)) + ph, ow = 2 * (w0 + (r & 7)) + pw;
                const size_t vox = (((((size_t)b << lo) + od) << lo) + oh << lo) + ow;
                *reinterpret_cast<uint4 *>(reinterpret_cast<char *>(y) + vox * (COUT * 2) + c * 16) =
                    *reinterpret_cast<const uint4 *>(mystage + r * SPITCH + c * 16);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
        }
    };
#pragma unroll 1
    for (int pi = 0; pi < PPW; ++pi) {
        switch (act) {
            case VV_ACT_ELU: run_parity(std::integral_constant<int, VV_ACT_ELU>{}, pi); break;
            case VV_ACT_RELU: run_parity(std::integral_constant<int, VV_ACT_RELU>{}, pi); break;
            case VV_ACT_LRELU: run_parity(std::integral_constant<int, VV_ACT_LRELU>{}, pi); break;
            default: run_parity(std::integral_constant<int, VV_ACT_NONE>{}, pi); break;
        }
    }
}

inline int grid_1d(long n) {
    long g = (n + 255) / 256;
    return (int)(g < 1 ? 1 : (g > 8192 ? 8192 : g));
}

}  // namespace

VV_EXPORT int vv_convT3d_k4s2_direct_supported(int side, int cin, int cout, int dtype) {
    return dtype == VV_BF16 && cin == 128 && cout == 64 && side >= 8 && vv_is_pow2(side);
}

VV_EXPORT int vv_pack_convT_k4s2_frag(const float *w_keras, void *packed, int cin, int cout, void *stream) {
    if (!w_keras || !packed) return VV_ERR_NULL;
    if (cin <= 0 || cout <= 0 || cin % 16 || cout % 32) return VV_ERR_SHAPE;
    VV_LAUNCH(pack_convT_frag_kernel, dim3(grid_1d((long)64 * cin * cout)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), w_keras,
              reinterpret_cast<__bf16 *>(packed), cin, cout);
    return vv_launch_status();
}

VV_EXPORT int vv_convT3d_k4s2_direct_fwd(const void *x, const void *w_frag, const float *scale, const float *shift, void *y,
                                         int batch, int side, int cin, int cout, int act, int dtype, void *stream) {
    if (!x || !w_frag || !y) return VV_ERR_NULL;
    if (!vv_convT3d_k4s2_direct_supported(side, cin, cout, dtype) || batch <= 0) return VV_ERR_SHAPE;
    if (!vv_aligned16(x) || !vv_aligned16(w_frag) || !vv_aligned16(y)) return VV_ERR_ALIGN;
    // variants: "2" = 2x4x8 cells, 4 waves x 2 parities, 2 workgroups / CU; "4" = 4x4x8 cells, 4 waves x 2 parities;
    // "8" = 4x4x8 cells, 8 waves x 1 parity (each weight fragment feeds 4 MFMAs, 2 waves / SIMD)
    const char *sel_env = vv_hook("VV_DIRECT_MT");                       // read per call so that tests can cover every variant
    const int sel = sel_env ? atoi(sel_env) : 8;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    auto launch = [&](auto mt_c, auto nw_c) {
        constexpr int MT = decltype(mt_c)::value, NW = decltype(nw_c)::value;
        const int boxes = (side / MT) * (side / 4) * (side / 8);
        constexpr int LDS = (MT + 2) * HH * HW * 128 * 2 + NW * 32 * (64 * 2 + 16);
        static const bool attr = [] {
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&convT_direct_kernel<128, 64, MT, NW>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
            return true;
        }();
        (void)attr;
        const size_t in_per = (size_t)side * side * side * cin * 2, out_per = (size_t)8 * side * side * side * cout * 2;
        const int per = vv_chunk_samples(in_per, batch);           // 32-bit buffer offsets: <= 2 GiB of input per launch
        for (int b0 = 0; b0 < batch && per > 0; b0 += per) {
            const int nb = batch - b0 < per ? batch - b0 : per;
            VV_LAUNCH((convT_direct_kernel<128, 64, MT, NW>), dim3(nb * boxes), dim3(NW * 64), LDS, st,
                      reinterpret_cast<const __bf16 *>(reinterpret_cast<const char *>(x) + (size_t)b0 * in_per),
                      reinterpret_cast<const __bf16 *>(w_frag), scale, shift,
                      reinterpret_cast<__bf16 *>(reinterpret_cast<char *>(y) + (size_t)b0 * out_per), vv_log2(side),
                      (unsigned)((size_t)nb * in_per), act);
        }
    };
    if (sel == 8) launch(std::integral_constant<int, 4>{}, std::integral_constant<int, 8>{});
    else if (sel == 4) launch(std::integral_constant<int, 4>{}, std::integral_constant<int, 4>{});
    else launch(std::integral_constant<int, 2>{}, std::integral_constant<int, 4>{});
    return vv_launch_status();
}

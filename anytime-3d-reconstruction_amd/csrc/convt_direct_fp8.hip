// Direct transposed convolution with the input tile resident in LDS, fp8 (OCP e4m3fn) operands on the block-scaled
// K = 64 MFMA (gfx950).  The fp8 twin of convt_direct.hip's 8-wave form for the widest decoder layer (Cin 128 -> Cout 64):
// same decomposition (8 output parities x 8 taps over a (4+2) x 6 x 10 halo tile of a 4 x 4 x 8 block of input cells,
// one parity per wave, weights streamed from a fragment-ordered panel through a 4-deep register ring, weights-first MFMA
// so that a lane owns one cell and its registers walk the channels), with
//   * 128-byte voxel rows (Cin 128 x 1 byte): 8 slots of 16 B, XOR-swizzled with ((zw >> 1) + 4 zh) & 7 -- with 128-byte
//     rows two voxels share the 64 banks, so the parity of zw picks the half and the swizzle has to separate the 8 cells
//     of a ds_read_b128 lane group that share that parity; exhaustively checked for all 9 tap offsets and both groups;
//   * v_mfma_scale_f32_32x32x64_f8f6f4 with unit E8M0 scales: one instruction takes a whole 64-channel half of the tap
//     (32 bytes per lane and operand: two adjacent slots of the voxel row, two 16-byte loads of the weight fragment), so
//     a tap is 2 k-steps instead of 8 and retires in half the matrix-pipe time of the bf16 form;
//   * the output stored as bf16 (the last layer consumes bf16).
// Per-output-channel weight scales ride in `scale` (folded into the BatchNorm scale by the caller).
#include <type_traits>

#include "common.h"

namespace {

typedef __attribute__((ext_vector_type(8))) int i32x8;

constexpr int HH = 6, HW = 10;       // halo tile of a 4 x 4 x 8 block of cells: 6 x 6 x 10 voxels
constexpr int F8_MT = 4, F8_NW = 8;

// out (16-byte units) [p][a][ks][nt][h][lane]: bytes j = 0..15 = w[t(p,a)][co = nt*32 + (lane & 31)][ci = ks*64 + 32*(lane >> 5) + 16 h + j]
__global__ void pack_convT_frag_fp8_kernel(const float *__restrict__ w, unsigned *__restrict__ out, int cin, int cout) {
    const int KS = cin / 64, NT = cout / 32;
    const long total4 = (long)16 * cin * cout;            // 64 taps * cin * cout bytes, 4 per thread
    for (long i4 = (long)blockIdx.x * blockDim.x + threadIdx.x; i4 < total4; i4 += (long)gridDim.x * blockDim.x) {
        const long i = i4 * 4;
        const int j = (int)(i & 15), lane = (int)((i >> 4) & 63);
        long r = i >> 10;
        const int h = (int)(r & 1); r >>= 1;
        const int nt = (int)(r % NT); r /= NT;
        const int ks = (int)(r % KS); r /= KS;
        const int a = (int)(r & 7), p = (int)(r >> 3);
        const int td = 1 - ((p >> 2) & 1) + 2 * ((a >> 2) & 1);
        const int th = 1 - ((p >> 1) & 1) + 2 * ((a >> 1) & 1);
        const int tw = 1 - (p & 1) + 2 * (a & 1);
        const int t = (td * 4 + th) * 4 + tw;
        const int co = nt * 32 + (lane & 31), ci = ks * 64 + 32 * (lane >> 5) + 16 * h + j;
        const f32x4 v = *reinterpret_cast<const f32x4 *>(w + ((size_t)t * cout + co) * cin + ci);
        out[i4] = vv_pack_fp8x4(v);
    }
}

template <int CIN, int COUT, bool OUT8>     // OUT8: store e4m3fn (for the fp8 last layer) instead of bf16
__global__ __launch_bounds__(F8_NW * 64, 1) void convT_direct_fp8_kernel(const unsigned char *__restrict__ x, const uint4 *__restrict__ wf,
                                                                          const float *__restrict__ scale, const float *__restrict__ shift,
                                                                          void *__restrict__ y, int din_log2, unsigned x_bytes, int act) {
    constexpr int MT = F8_MT, NW = F8_NW;
    constexpr int RB = CIN;              // bytes per voxel row
    constexpr int KS = CIN / 64;         // K = 64 MFMA steps per tap
    constexpr int NT = COUT / 32;        // channel tiles
    constexpr int GPP = 8 * KS * NT;     // weight groups (tap, k-step, channel tile) per parity; 2 x 16 bytes per lane each
    constexpr int SPITCH = COUT * 2 + 16;
    constexpr int HV = (MT + 2) * HH * HW;
    static_assert(RB == 128 && NT == 2 && KS == 2, "swizzle and group schedule are derived for Cin 128 -> Cout 64");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *tile = smem;                                   // [360][128]
    char *stage = smem + HV * RB;                        // [8 waves][32][SPITCH]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = din_log2, n = 1 << li;
    const int bxw = n >> 3, bxh = n >> 2, bxd = n / MT;
    int blk = blockIdx.x;
    const int bw = blk % bxw; blk /= bxw;
    const int bh = blk % bxh; blk /= bxh;
    const int bd = blk % bxd; const int b = blk / bxd;
    const int d0 = bd * MT, h0 = bh * 4, w0 = bw * 8;

    // ---- stage the halo tile: 360 voxels x 8 slots = 45 wave instructions of 1 KiB (8 voxels each)
    {
        const u32x4 rs = vv_make_rsrc(x, x_bytes);
        const unsigned lds0 = (unsigned)(unsigned long long)(lptr_t)tile;
        const int pos = lane & 7, vsub = lane >> 3;
        for (int it = wave; it < HV / 8; it += NW) {
            const int v = it * 8 + vsub;
            const int zw = v % HW, zh = (v / HW) % HH, zd = v / (HW * HH);
            const int id = d0 - 1 + zd, ih = h0 - 1 + zh, iw = w0 - 1 + zw;
            const bool ok = (unsigned)id < (unsigned)n && (unsigned)ih < (unsigned)n && (unsigned)iw < (unsigned)n;
            const int g = pos ^ (((zw >> 1) + 4 * zh) & 7);
            const unsigned vo = ok ? (unsigned)((((((b << li) + id) << li) + ih) << li) + iw) * RB + g * 16 : 0xFFFFFFF0u;
            vv_dma16(rs, vo, lds0 + it * 1024);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();

    const int fr = lane & 31, fh = lane >> 5;
    const int mh = fr >> 3, mw = fr & 7;
    char *mystage = stage + wave * 32 * SPITCH;
    const int lo = li + 1;

    // Weights: this wave's parity = GPP consecutive groups of the panel, two coalesced 1 KiB loads each, streamed through a
    // 4-deep register ring (3 groups = 12 MFMAs of 64 cycles of prefetch distance).
    const uint4 *wp = wf + (size_t)wave * GPP * 2 * 64 + lane;
    uint4 b0[2], b1[2], b2[2], b3[2];
    auto load_group = [&](int G, uint4 *dst) {
        if (G < GPP) {
            dst[0] = wp[(size_t)(G * 2) * 64];
            dst[1] = wp[(size_t)(G * 2 + 1) * 64];
        }
    };
    load_group(0, b0);
    load_group(1, b1);
    load_group(2, b2);

    const int p = wave, pd = (p >> 2) & 1, ph = (p >> 1) & 1, pw = p & 1;
    f32x16 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[mt][nt][q] = 0.f;

    const unsigned tile_lds = (unsigned)(unsigned long long)(lptr_t)tile;
    // first 16 bytes of this lane's 32 for tap a / k-step ks; the second 16 are the neighbouring slot: address ^ 16
    auto a_addr = [&](int a, int ks) -> unsigned {
        const int ad = (a >> 2) & 1, ah = (a >> 1) & 1, aw = a & 1;
        const int zh = mh + ph - ah + 1, zw = mw + pw - aw + 1;
        const int sw = ((zw >> 1) + 4 * zh) & 7;
        return tile_lds + (((pd - ad + 1) * HH + zh) * HW + zw) * RB + (((ks * 4 + fh * 2) ^ sw) << 4);
    };
    // A fragments: two sets, the one of the next k-step in flight while the 8 MFMAs of the current k-step run
    u32x4 FP[MT][2], FQ[MT][2];
    auto ld = [&](u32x4 (&F)[MT][2], unsigned addr) {
        const unsigned addr2 = addr ^ 16u;
        asm volatile("ds_read_b128 %0, %8\n\tds_read_b128 %1, %9\n\t"
                     "ds_read_b128 %2, %8 offset:%10\n\tds_read_b128 %3, %9 offset:%10\n\t"
                     "ds_read_b128 %4, %8 offset:%11\n\tds_read_b128 %5, %9 offset:%11\n\t"
                     "ds_read_b128 %6, %8 offset:%12\n\tds_read_b128 %7, %9 offset:%12"
                     : "=&v"(F[0][0]), "=&v"(F[0][1]), "=&v"(F[1][0]), "=&v"(F[1][1]), "=&v"(F[2][0]), "=&v"(F[2][1]), "=&v"(F[3][0]), "=&v"(F[3][1])
                     : "v"(addr), "v"(addr2), "n"(HH * HW * RB), "n"(2 * HH * HW * RB), "n"(3 * HH * HW * RB)
                     : "memory");
    };
    auto wait_a = [&](u32x4 (&F)[MT][2], auto n_c) {       // LDS returns in order: lgkmcnt(8) = all but the 8 reads just issued
        asm volatile("s_waitcnt lgkmcnt(%8)"
                     : "+v"(F[0][0]), "+v"(F[0][1]), "+v"(F[1][0]), "+v"(F[1][1]), "+v"(F[2][0]), "+v"(F[2][1]), "+v"(F[3][0]), "+v"(F[3][1])
                     : "n"(decltype(n_c)::value)
                     : "memory");
    };
    auto mma = [&](const u32x4 (&F)[MT][2], const uint4 *bw, int nt) {
        const i32x8 wv = {(int)bw[0].x, (int)bw[0].y, (int)bw[0].z, (int)bw[0].w, (int)bw[1].x, (int)bw[1].y, (int)bw[1].z, (int)bw[1].w};
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const i32x8 xv = {(int)F[mt][0][0], (int)F[mt][0][1], (int)F[mt][0][2], (int)F[mt][0][3],
                              (int)F[mt][1][0], (int)F[mt][1][1], (int)F[mt][1][2], (int)F[mt][1][3]};
            if (nt == 0) acc[mt][0] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(wv, xv, acc[mt][0], 0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
            else acc[mt][1] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(wv, xv, acc[mt][1], 0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) asm volatile("" : "+v"(acc[mt][0]), "+v"(acc[mt][1]));   // pin the (pure) MFMAs in place: without
        __builtin_amdgcn_sched_barrier(0);                                                         // this the kernel needs all 256 VGPRs
    };

    // k-step kk = (tap a = kk >> 1, ks = kk & 1) uses groups 2 kk (channel tile 0) and 2 kk + 1 (channel tile 1)
    using N8 = std::integral_constant<int, 8>;
    ld(FP, a_addr(0, 0));
#pragma unroll 1
    for (int kk = 0; kk < 8 * KS; kk += 2) {
        const int G = 2 * kk;
        load_group(G + 3, b3);
        ld(FQ, a_addr((kk + 1) >> 1, (kk + 1) & 1));
        wait_a(FP, N8{});
        mma(FP, b0, 0);
        load_group(G + 4, b0);
        mma(FP, b1, 1);
        load_group(G + 5, b1);
        ld(FP, a_addr(((kk + 2) >> 1) & 7, kk & 1));       // wraps harmlessly after the last tap
        wait_a(FQ, N8{});
        mma(FQ, b2, 0);
        load_group(G + 6, b2);
        mma(FQ, b3, 1);
    }
    wait_a(FP, std::integral_constant<int, 0>{});          // the wrapped look-ahead read

    // ---- epilogue: lane = cell (mt, mh, mw), registers walk channels; folded BN quads fetched as one batch
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
    auto finish = [&](auto act_c) {
        constexpr int ACT = decltype(act_c)::value;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int c = nt * 32 + 8 * g + 4 * fh;
                    const f32x4 sc = scv[nt][g], sh = shv[nt][g];
                    f32x4 tv;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float t = acc[mt][nt][4 * g + e] * sc[e] + sh[e];
                        if (ACT == VV_ACT_ELU) { const float em = __expf(fminf(t, 0.f)) - 1.f; t = t > 0.f ? t : em; }
                        else if (ACT == VV_ACT_RELU) t = fmaxf(t, 0.f);
                        else if (ACT == VV_ACT_LRELU) t = t > 0.f ? t : 0.3f * t;
                        tv[e] = t;
                    }
                    if constexpr (OUT8) {
                        *reinterpret_cast<unsigned *>(mystage + fr * SPITCH + c) = vv_pack_fp8x4(tv);
                    } else {
                        bf16x4 o;
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = static_cast<__bf16>(tv[e]);
                        *reinterpret_cast<bf16x4 *>(mystage + fr * SPITCH + c * 2) = o;
                    }
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            constexpr int ES = OUT8 ? 1 : 2;
            constexpr int CPR = COUT * ES / 16;           // 16-byte chunks per output row
#pragma unroll
            for (int i = 0; i < 32 * CPR / 64; ++i) {
                const int id = lane + 64 * i, r = id / CPR, c = id % CPR;
                const int od = 2 * (d0 + mt) + pd, oh = 2 * (h0 + (r >> 3)) + ph, ow = 2 * (w0 + (r & 7)) + pw;
                const size_t vox = (((((size_t)b << lo) + od) << lo) + oh << lo) + ow;
                *reinterpret_cast<uint4 *>(reinterpret_cast<char *>(y) + vox * (COUT * ES) + c * 16) =
                    *reinterpret_cast<const uint4 *>(mystage + r * SPITCH + c * 16);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
        }
    };
    switch (act) {
        case VV_ACT_ELU: finish(std::integral_constant<int, VV_ACT_ELU>{}); break;
        case VV_ACT_RELU: finish(std::integral_constant<int, VV_ACT_RELU>{}); break;
        case VV_ACT_LRELU: finish(std::integral_constant<int, VV_ACT_LRELU>{}); break;
        default: finish(std::integral_constant<int, VV_ACT_NONE>{}); break;
    }
}

inline int grid_1d(long n) {
    long g = (n + 255) / 256;
    return (int)(g < 1 ? 1 : (g > 8192 ? 8192 : g));
}

}  // namespace

VV_EXPORT int vv_convT3d_k4s2_direct_fp8_supported(int side, int cin, int cout) {
    return cin == 128 && cout == 64 && side >= 8 && vv_is_pow2(side);
}

VV_EXPORT int vv_pack_convT_k4s2_frag_fp8(const float *w_keras, void *packed, int cin, int cout, void *stream) {
    if (!w_keras || !packed) return VV_ERR_NULL;
    if (cin <= 0 || cout <= 0 || cin % 64 || cout % 32) return VV_ERR_SHAPE;
    if (!vv_aligned16(w_keras) || !vv_aligned16(packed)) return VV_ERR_ALIGN;
    VV_LAUNCH(pack_convT_frag_fp8_kernel, dim3(grid_1d((long)16 * cin * cout)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), w_keras,
              reinterpret_cast<unsigned *>(packed), cin, cout);
    return vv_launch_status();
}

VV_EXPORT int vv_convT3d_k4s2_direct_fp8_fwd(const void *x, const void *w_frag, const float *scale, const float *shift, void *y,
                                             int batch, int side, int cin, int cout, int act, int out_dtype, void *stream) {
    if (!x || !w_frag || !y) return VV_ERR_NULL;
    if (out_dtype != VV_BF16 && out_dtype != VV_FP8) return VV_ERR_DTYPE;
    if (!vv_convT3d_k4s2_direct_fp8_supported(side, cin, cout) || batch <= 0) return VV_ERR_SHAPE;
    if (!vv_aligned16(x) || !vv_aligned16(w_frag) || !vv_aligned16(y)) return VV_ERR_ALIGN;
    const size_t xb = (size_t)batch * side * side * side * cin;
    if (xb >= 0xFFFFFFF0ull) return VV_ERR_SHAPE;
    const int boxes = (side / F8_MT) * (side / 4) * (side / 8);
    constexpr int LDS = (F8_MT + 2) * HH * HW * 128 + F8_NW * 32 * (64 * 2 + 16);
    static const bool attr = [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&convT_direct_fp8_kernel<128, 64, false>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&convT_direct_fp8_kernel<128, 64, true>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        return true;
    }();
    (void)attr;
    const unsigned char *xb8 = reinterpret_cast<const unsigned char *>(x);
    const uint4 *wf4 = reinterpret_cast<const uint4 *>(w_frag);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (out_dtype == VV_FP8)
        VV_LAUNCH((convT_direct_fp8_kernel<128, 64, true>), dim3(batch * boxes), dim3(F8_NW * 64), LDS, st, xb8, wf4, scale, shift, y, vv_log2(side), (unsigned)xb, act);
    else
        VV_LAUNCH((convT_direct_fp8_kernel<128, 64, false>), dim3(batch * boxes), dim3(F8_NW * 64), LDS, st, xb8, wf4, scale, shift, y, vv_log2(side), (unsigned)xb, act);
    return vv_launch_status();
}

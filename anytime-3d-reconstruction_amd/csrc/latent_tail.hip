// The latent tail of the evaluation path as two launches instead of five (bf16, gfx950):
//
//   encoder tail (Conv3D k4 s1 + spatial mean folded into one Dense panel, autoencoder3D.py:86-91)
//     -> slice | clip | sampling | KL (nolbo.py:1417-1431; function.py:35-38, 84-98)
//     -> linearTransform + BN + act (autoencoder3D.py:56-70) -> first decoder Conv3DTranspose (k4 s1) + BN + act (:127-128)
//
// Five tiny dependent kernels (a 256 x 128 x 4096 GEMM in 16-chunk latency chains, its split-K reduce, a 24 KB elementwise
// pass, two GEMMs with K = 64) cost 45 us of a 590 us step, almost all of it launch gaps and exposed memory latency.  Here:
//
//   lt_e5_kernel : the encoder-tail GEMM cut into K slices of >= 256; a workgroup stages its WHOLE slice (4 chunks of 64) with
//                  one burst of LDS-DMA, multiplies, and stores a float32 slab [slice][B][E] -- one memory round trip per
//                  workgroup instead of a chain of them
//   lt_mid_kernel: workgroup = 16 samples x one slice of the decoder seed: sums the slabs in slice order (deterministic), clips,
//                  samples z, sums KL, runs the K = L dense layer and its slice of the K = S^3*8 dense layer as 16-row MFMA
//                  tiles (v_mfma_f32_16x16x32_bf16, weights straight from global memory: they are read once per workgroup),
//                  folded BN + activation in the epilogues.  The first three steps are recomputed by every slice's workgroup
//                  (a few thousand FLOPs) instead of being exchanged.
#include <stdlib.h>

#include <type_traits>

#include "common.h"

namespace {

typedef __attribute__((ext_vector_type(4))) unsigned lt_u4;

constexpr int LT_BM = 128, LT_BN = 64, LT_KS = 256;          // tile and the K a workgroup stages at once
constexpr int LT_E5_LDS = (LT_BM + LT_BN) * 128 * (LT_KS / 64);   // 98,304 B

struct LtArgs {
    const void *h;          // [B][K5] bf16
    const void *w5;         // [E][K5] bf16
    float *slabs;           // [nslice][B][E]
    int batch, K5, E, kslice, nslice;
    unsigned h_bytes, w_bytes;
};

__global__ __launch_bounds__(256) void lt_e5_kernel(const LtArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;                  // 64-row block, 32-channel block
    const int ntn = (a.E + LT_BN - 1) / LT_BN, ntm = (a.batch + LT_BM - 1) / LT_BM;
    int blk = blockIdx.x;
    const int tn = blk % ntn; blk /= ntn;
    const int tm = blk % ntm;
    const int sl = blk / ntm;
    const int m0 = tm * LT_BM, n0 = tn * LT_BN;
    const int k_begin = sl * a.kslice, k_end = k_begin + a.kslice < a.K5 ? k_begin + a.kslice : a.K5;
    const u32x4 rsa = vv_make_rsrc(a.h, a.h_bytes), rsw = vv_make_rsrc(a.w5, a.w_bytes);
    const unsigned lds0 = (unsigned)(unsigned long long)(lptr_t)smem;

    const int fr = lane & 31, fh = lane >> 5;
    f32x16 acc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;

    for (int kr = k_begin; kr < k_end; kr += LT_KS) {         // rounds of 4 chunks (one round for K5 <= 32 * 256)
        // stage [4 chunks][128 + 64 rows][128 B]: 96 pieces of 1 KiB, 24 per wave; slot swizzle (row >> 1) & 7 on the source side
        for (int it = wave; it < 96; it += 4) {
            const int ch = it / 24, r8 = it % 24;             // chunk, 8-row block (0..15 activations, 16..23 weights)
            const int k0 = kr + ch * 64;
            const int row = (r8 < 16 ? r8 : r8 - 16) * 8 + (lane >> 3);
            const int slot = (lane & 7) ^ ((row >> 1) & 7);
            const bool kin = k0 + slot * 8 < k_end;
            unsigned vo;
            if (r8 < 16) vo = (kin && m0 + row < a.batch) ? (unsigned)(((size_t)(m0 + row) * a.K5 + k0) * 2 + slot * 16) : 0xFFFFFFF0u;
            else vo = (kin && n0 + row < a.E) ? (unsigned)(((size_t)(n0 + row) * a.K5 + k0) * 2 + slot * 16) : 0xFFFFFFF0u;
            const unsigned dst = lds0 + ch * (LT_BM + LT_BN) * 128 + r8 * 1024;
            if (r8 < 16) vv_dma16(rsa, vo, dst);
            else vv_dma16(rsw, vo, dst);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
#pragma unroll
        for (int ch = 0; ch < 4; ++ch) {
            const char *As = smem + ch * (LT_BM + LT_BN) * 128, *Bs = As + LT_BM * 128;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int rb = wn * 32 + fr;
                const uint4 fb = *reinterpret_cast<const uint4 *>(Bs + rb * 128 + (((ks * 2 + fh) ^ ((rb >> 1) & 7)) << 4));
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int ra = wm * 64 + i * 32 + fr;
                    const uint4 fa = *reinterpret_cast<const uint4 *>(As + ra * 128 + (((ks * 2 + fh) ^ ((ra >> 1) & 7)) << 4));
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8 *>(&fb), *reinterpret_cast<const bf16x8 *>(&fa),
                                                                     acc[i], 0, 0, 0);      // D[n][m]: lane = row m, registers walk n
                }
            }
        }
        __syncthreads();
    }
    // slab store: lane = sample row, register quad g = channels 8 g + 4 fh .. + 3 of the wave's 32
    float *slab = a.slabs + (size_t)sl * a.batch * a.E;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int m = m0 + wm * 64 + i * 32 + fr;
        if (m >= a.batch) continue;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int n = n0 + wn * 32 + 8 * g + 4 * fh;
            if (n < a.E)
                *reinterpret_cast<f32x4 *>(slab + (size_t)m * a.E + n) = f32x4{acc[i][4 * g], acc[i][4 * g + 1], acc[i][4 * g + 2], acc[i][4 * g + 3]};
        }
    }
}

struct LtMidArgs {
    const float *slabs;      // [nslice][B][E]
    const float *e5_scale;   // per-channel scale of the encoder tail (fp8 weight scale) or nullptr
    const float *eps;        // [B][L] or nullptr (non-variational)
    const void *wd;          // [lin][L] bf16
    const float *scale_d, *shift_d;
    const void *w1;          // [n1][lin] bf16
    const float *scale_1, *shift_1;
    float *enc_out;          // [B][E] or nullptr
    float *z;                // [B][L]
    void *z_act;             // [B][L] bf16 or nullptr
    float *kl;               // [B] or nullptr
    void *h1;                // [B][n1] bf16
    int batch, E, L, lin, n1, nslice, nq, variational, act;
};

template <int ACT>
__device__ __forceinline__ float lt_act(float t) {
    if (ACT == VV_ACT_ELU) { const float em = __expf(fminf(t, 0.f)) - 1.f; return t > 0.f ? t : em; }
    if (ACT == VV_ACT_RELU) return fmaxf(t, 0.f);
    if (ACT == VV_ACT_LRELU) return t > 0.f ? t : 0.3f * t;
    return t;
}

// Up to 8 output tiles x 2 k-steps of weight fragments of one wave, loaded as ONE burst (16 independent 16-byte loads per lane):
// the dense layers here are a few MFMAs behind a memory round trip, so every load that can be in flight at once must be.
// Tile i of the wave is tile index t0 + 4 i (< ntiles); fragment = W[(tile*16 + lane & 15)][ks*32 + (lane >> 4)*8 ..].
__device__ __forceinline__ void lt_load_w(uint4 (&wf)[8][2], const __bf16 *w, int ld, int row0, int t0, int ntiles, int ks0, int nk, int r16,
                                          int kq) {
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int t = t0 + 4 * i, ks = ks0 + k;
            wf[i][k] = (t < ntiles && ks < nk) ? *reinterpret_cast<const uint4 *>(w + (size_t)(row0 + t * 16 + r16) * ld + ks * 32 + kq * 8)
                                               : make_uint4(0, 0, 0, 0);
        }
}

template <int ACT>
__global__ __launch_bounds__(256) void lt_mid_kernel(const LtMidArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int sb = blockIdx.x / a.nq, q = blockIdx.x - sb * a.nq;     // 16-sample block, slice of the seed
    const int b0 = sb * 16;
    float *encs = reinterpret_cast<float *>(smem);                    // [16][E] float32
    __bf16 *zs = reinterpret_cast<__bf16 *>(smem + 16 * a.E * 4);     // [16][L] bf16  (MFMA operand)
    __bf16 *ts = zs + 16 * a.L;                                       // [16][lin] bf16
    const int r16 = lane & 15, kq = lane >> 4;
    const __bf16 *wd = reinterpret_cast<const __bf16 *>(a.wd), *w1 = reinterpret_cast<const __bf16 *>(a.w1);
    const int slice = a.n1 / a.nq, nk0 = a.L / 32, nk1 = a.lin / 32, nt0 = a.lin / 16, nt1 = slice / 16;

    // the weights (and the sampling noise) do not depend on the data: the first burst of both dense layers is in flight while the
    // slabs are summed
    float epsv[4] = {0.f, 0.f, 0.f, 0.f};          // this thread's noise values: row tid >> 4, columns (tid & 15) + 16 k  (L <= 64; else re-read)
    if (a.variational && b0 + (tid >> 4) < a.batch) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if ((tid & 15) + 16 * k < a.L) epsv[k] = a.eps[(size_t)(b0 + (tid >> 4)) * a.L + (tid & 15) + 16 * k];
    }
    uint4 wA[8][2], wB[8][2];
    lt_load_w(wA, wd, a.L, 0, wave, nt0, 0, nk0, r16, kq);
    lt_load_w(wB, w1, a.lin, q * slice, wave, nt1, 0, nk1, r16, kq);

    // ---- A: encoder output = slabs summed in slice order.  A thread owns up to two (row, channel quad) items and keeps the
    // loads of BOTH in flight (8 slices each per round): the sum is a chain of memory round trips and nothing else.
    const int e4 = a.E >> 2;
    for (int i0 = tid; i0 < 16 * e4; i0 += 512) {
        f32x4 s[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        const float *src[2];
        bool on[2];
        int rr[2], cc[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int i = i0 + 256 * u;
            rr[u] = i / e4; cc[u] = i - rr[u] * e4;
            on[u] = i < 16 * e4 && b0 + rr[u] < a.batch;
            src[u] = a.slabs + (size_t)(b0 + (on[u] ? rr[u] : 0)) * a.E + (on[u] ? cc[u] : 0) * 4;
        }
        const size_t sstride = (size_t)a.batch * a.E;
        for (int sl0 = 0; sl0 < a.nslice; sl0 += 8) {
            f32x4 v[2][8];
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    v[u][k] = (on[u] && sl0 + k < a.nslice) ? *reinterpret_cast<const f32x4 *>(src[u] + (size_t)(sl0 + k) * sstride) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int k = 0; k < 8; ++k) s[u] += v[u][k];
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (i0 + 256 * u >= 16 * e4) continue;
            if (on[u]) {
                if (a.e5_scale) s[u] *= *reinterpret_cast<const f32x4 *>(a.e5_scale + cc[u] * 4);
                if (a.enc_out && q == 0) *reinterpret_cast<f32x4 *>(a.enc_out + (size_t)(b0 + rr[u]) * a.E + cc[u] * 4) = s[u];
            }
            *reinterpret_cast<f32x4 *>(encs + rr[u] * a.E + cc[u] * 4) = s[u];
        }
    }
    __syncthreads();
    // ---- B: slice | clip | sampling | KL: 16 lanes per sample row
    {
        const int r = tid >> 4, l16 = tid & 15;
        const bool live = b0 + r < a.batch;
        float s = 0.f;
        for (int j = l16; j < a.L; j += 16) {
            float zz;
            if (a.variational) {
                const float mu = encs[r * a.E + j];
                float lv = encs[r * a.E + a.L + j];
                lv = fminf(fmaxf(lv, -10.f), 10.f);                                   // nolbo.py:1420
                const float e = expf(lv);
                const int kk = j >> 4;
                const float ev = kk < 4 ? (kk == 0 ? epsv[0] : kk == 1 ? epsv[1] : kk == 2 ? epsv[2] : epsv[3])
                                        : (live ? a.eps[(size_t)(b0 + r) * a.L + j] : 0.f);
                zz = mu + sqrtf(e) * ev;                                              // function.py:37
                s += 0.5f * (0.f - lv) + (e + mu * mu) / 2.0f - 0.5f;                 // function.py:96, target N(0, I)
            } else {
                zz = encs[r * a.E + j];
            }
            zs[r * a.L + j] = static_cast<__bf16>(zz);
            if (live && q == 0) {
                a.z[(size_t)(b0 + r) * a.L + j] = zz;
                if (a.z_act) reinterpret_cast<__bf16 *>(a.z_act)[(size_t)(b0 + r) * a.L + j] = static_cast<__bf16>(zz);
            }
        }
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        if (a.kl && a.variational && live && q == 0 && l16 == 0) a.kl[b0 + r] = s;
    }
    __syncthreads();

    // One dense layer on 16 rows: out[row][n] = act((x[row][:] . W[n][:]) * scale[n] + shift[n]) for the wave's tiles t0 + 4 i.
    // Weights first (D[n][row]): lane = row (lane & 15), registers = 4 consecutive outputs.  `pre` holds the burst for tiles
    // t0 .. t0 + 28 and k-steps 0, 1; anything beyond (wider layers) is fetched in further bursts.
    auto dense16 = [&](const __bf16 *x, int K, int nk, const __bf16 *w, int row0, int ntiles, uint4 (&pre)[8][2], const float *scale,
                       const float *shift, int nbase, auto &&store) {
        for (int tb = wave; tb < ntiles; tb += 32) {
            f32x4 c[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) c[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            for (int ks0 = 0; ks0 < nk; ks0 += 2) {
                if (tb != wave || ks0 != 0) lt_load_w(pre, w, K, row0, tb, ntiles, ks0, nk, r16, kq);
                uint4 xf[2];
#pragma unroll
                for (int k = 0; k < 2; ++k)
                    xf[k] = ks0 + k < nk ? *reinterpret_cast<const uint4 *>(x + r16 * K + (ks0 + k) * 32 + kq * 8) : make_uint4(0, 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 8; ++i)
#pragma unroll
                    for (int k = 0; k < 2; ++k)
                        c[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8 *>(&pre[i][k]),
                                                                       *reinterpret_cast<const bf16x8 *>(&xf[k]), c[i], 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int t = tb + 4 * i;
                if (t >= ntiles) continue;
                const int n = t * 16 + kq * 4;            // local output index
                const f32x4 sc = scale ? *reinterpret_cast<const f32x4 *>(scale + nbase + n) : f32x4{1.f, 1.f, 1.f, 1.f};
                const f32x4 sh = shift ? *reinterpret_cast<const f32x4 *>(shift + nbase + n) : f32x4{0.f, 0.f, 0.f, 0.f};
                bf16x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = static_cast<__bf16>(lt_act<ACT>(c[i][e] * sc[e] + sh[e]));
                store(n, o);
            }
        }
    };
    // ---- C: t = act(BN(z Wd^T + b))
    dense16(zs, a.L, nk0, wd, 0, nt0, wA, a.scale_d, a.shift_d, 0, [&](int n, bf16x4 o) { *reinterpret_cast<bf16x4 *>(ts + r16 * a.lin + n) = o; });
    __syncthreads();
    // ---- D: this workgroup's slice of the seed: h1[:, q*slice .. ) = act(BN(t W1^T))
    dense16(ts, a.lin, nk1, w1, q * slice, nt1, wB, a.scale_1, a.shift_1, q * slice, [&](int n, bf16x4 o) {
        if (b0 + r16 < a.batch) *reinterpret_cast<bf16x4 *>(reinterpret_cast<__bf16 *>(a.h1) + (size_t)(b0 + r16) * a.n1 + q * slice + n) = o;
    });
}


// ---- round 4: the 4^3 -> 2^3 convolution's split-K slabs summed HERE instead of by a reduce launch (posgemm.hip writes slabs only:
// vv_pg_conv_slabs).  Workgroup = 16 samples x one output position p of that layer x 256 of its channels = one K slice of the
// encoder-tail GEMM: the threads sum the position's shares in share order (what pg_reduce_kernel does, same order, same sums), apply the
// layer's folded BN + activation, round to bf16 (the value the unfused path stores) and leave the [16][256] tile in LDS as the MFMA
// operand; the tile meets its [E][256] slice of the tail panel (fragments straight from global memory, in flight since the first
// instruction) on v_mfma_f32_16x16x32_bf16 and goes out as the float32 slab [slice][B][E] that lt_mid_kernel sums.  The convolution's
// output never exists in memory, its reduce launch and lt_e5_kernel's staging round trip are gone.
struct LtE5xArgs {
    const float *pslabs;      // posgemm slabs
    const float *scale4, *shift4;
    const void *w5;           // [E][K5] bf16, K5 = 8 * cout4
    float *slabs;             // [8 * nh][B][E]
    int batch, cout4, E, nh, mtiles, rows_per_tile, ntn, act;
    unsigned char nsplit[8];
    unsigned short first[8];
};

constexpr int LTX_PITCH = 512 + 16;      // bytes per row of the [16][256] bf16 tile (16-byte pad: rows 4 banks apart)

template <int ACT>
__global__ __launch_bounds__(256) void lt_e5x_kernel(const LtE5xArgs a) {
    __shared__ __attribute__((aligned(16))) char xs[16 * LTX_PITCH];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int per = 8 * a.nh;
    const int sb = blockIdx.x / per, rem = blockIdx.x - sb * per;
    const int p = rem / a.nh, chh = rem - p * a.nh;
    const int b0 = sb * 16;
    const int K5 = 8 * a.cout4;
    const int r16 = lane & 15, kq = lane >> 4;
    const int ntile = a.E >> 4;                          // 16-column tiles of the tail's output, tile wave + 4 i to this wave

    // weight fragments of this wave's (<= 2) output tiles, 8 k-steps of 32: W5[t*16 + r16][p*cout4 + chh*256 + ks*32 + kq*8 ..]
    uint4 wf[2][8];
    {
        const __bf16 *w5 = reinterpret_cast<const __bf16 *>(a.w5);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int t = wave + 4 * i;
#pragma unroll
            for (int ks = 0; ks < 8; ++ks)
                wf[i][ks] = t < ntile ? *reinterpret_cast<const uint4 *>(w5 + (size_t)(t * 16 + r16) * K5 + p * a.cout4 + chh * 256 + ks * 32 + kq * 8)
                                      : make_uint4(0, 0, 0, 0);
        }
    }
    // ---- the tile: thread = (row tid & 15, channel quads (tid >> 4) + 16 j, j = 0..3).  The slabs are in the producer's fragment
    // order (vv_pg_frag_index): 16 consecutive rows of one quad are 256 contiguous bytes.  Shares summed in order, BN, activation, bf16
    {
        const int r = tid & 15, qg = tid >> 4;
        const int b = b0 + r;
        const bool live = b < a.batch;
        const int mt = b0 / a.rows_per_tile, rl = b - mt * a.rows_per_tile;
        const int ns = a.nsplit[p];
        const size_t piece = (size_t)a.rows_per_tile * 128 / 4;                 // f32x4 per [256][128] piece
        const size_t sstride = (size_t)a.mtiles * a.ntn * piece;               // one share further
        const f32x4 *src[4];
        int cch[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = chh * 256 + (qg + 16 * j) * 4;                        // channel of the layer
            cch[j] = c;
            src[j] = reinterpret_cast<const f32x4 *>(a.pslabs) + (((size_t)a.first[p] * a.mtiles + mt) * a.ntn + (c >> 7)) * piece +
                     vv_pg_frag_index(live ? rl : 0, c & 127);
        }
        f32x4 s[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        for (int s0 = 0; s0 < ns; s0 += 4) {             // 16 independent 16-byte loads in flight per trip
            f32x4 v[4][4];
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    v[k][j] = (live && s0 + k < ns) ? src[j][(size_t)(s0 + k) * sstride] : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int j = 0; j < 4; ++j) s[j] += v[k][j];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
            if (a.scale4) sc = *reinterpret_cast<const f32x4 *>(a.scale4 + cch[j]);
            if (a.shift4) sh = *reinterpret_cast<const f32x4 *>(a.shift4 + cch[j]);
            bf16x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = static_cast<__bf16>(live ? lt_act<ACT>(s[j][e] * sc[e] + sh[e]) : 0.f);
            *reinterpret_cast<bf16x4 *>(xs + r * LTX_PITCH + (cch[j] - chh * 256) * 2) = o;
        }
    }
    __syncthreads();
    // ---- [16][256] x [256][E]: weights first (D[n][row]): lane (r16, kq) ends with outputs t*16 + 4 kq .. + 3 of row r16
    f32x4 c[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
        const uint4 xf = *reinterpret_cast<const uint4 *>(xs + r16 * LTX_PITCH + ks * 64 + kq * 16);
#pragma unroll
        for (int i = 0; i < 2; ++i)
            c[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8 *>(&wf[i][ks]), *reinterpret_cast<const bf16x8 *>(&xf), c[i], 0, 0, 0);
    }
    if (b0 + r16 < a.batch) {
        float *dst = a.slabs + ((size_t)(p * a.nh + chh) * a.batch + b0 + r16) * a.E;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int t = wave + 4 * i;
            if (t < ntile) *reinterpret_cast<f32x4 *>(dst + t * 16 + kq * 4) = c[i];
        }
    }
}

struct LtPlan { int kslice, nslice, nq; size_t ws; };

LtPlan lt_plan(int batch, int K5, int E, int n1) {
    LtPlan p;
    static const int ks_env = vv_hook("VV_LT_KSLICE") ? atoi(vv_hook("VV_LT_KSLICE")) : 0;
    int ks = (K5 + 31) / 32;                               // at most 32 slices
    if (ks_env > ks) ks = ks_env;
    ks = ((ks + LT_KS - 1) / LT_KS) * LT_KS;
    if (ks < LT_KS) ks = LT_KS;
    p.kslice = ks;
    p.nslice = (K5 + ks - 1) / ks;
    p.nq = 1;
    for (int c = 16; c >= 2; c >>= 1)                      // 16 slices of the seed per 16-sample block: 256 workgroups at batch 256 (8 slices = 128 workgroups: 24.4 -> 22.7 us)
        if (n1 % (16 * c) == 0) { p.nq = c; break; }
    p.ws = (size_t)p.nslice * batch * E * sizeof(float);
    return p;
}

int lt_launch_mid(const float *slabs, int nslice, int nq, const float *e5_scale, const float *eps, const void *wd, const float *scale_d,
                  const float *shift_d, const void *w1, const float *scale_1, const float *shift_1, float *enc_out, float *z, void *z_act, float *kl,
                  void *h1, int batch, int E, int L, int lin, int n1, int variational, int act, hipStream_t st) {
    LtMidArgs m;
    m.slabs = slabs; m.e5_scale = e5_scale; m.eps = eps; m.wd = wd; m.scale_d = scale_d; m.shift_d = shift_d;
    m.w1 = w1; m.scale_1 = scale_1; m.shift_1 = shift_1; m.enc_out = enc_out; m.z = z; m.z_act = z_act; m.kl = kl; m.h1 = h1;
    m.batch = batch; m.E = E; m.L = L; m.lin = lin; m.n1 = n1; m.nslice = nslice; m.nq = nq; m.variational = variational; m.act = act;
    const size_t lds = (size_t)16 * E * 4 + (size_t)16 * L * 2 + (size_t)16 * lin * 2;
    const dim3 grid(((batch + 15) / 16) * nq);
    switch (act) {
        case VV_ACT_ELU: VV_LAUNCH(lt_mid_kernel<VV_ACT_ELU>, grid, dim3(256), lds, st, m); break;
        case VV_ACT_RELU: VV_LAUNCH(lt_mid_kernel<VV_ACT_RELU>, grid, dim3(256), lds, st, m); break;
        case VV_ACT_LRELU: VV_LAUNCH(lt_mid_kernel<VV_ACT_LRELU>, grid, dim3(256), lds, st, m); break;
        default: VV_LAUNCH(lt_mid_kernel<VV_ACT_NONE>, grid, dim3(256), lds, st, m); break;
    }
    return vv_launch_status();
}

}  // namespace

VV_EXPORT int vv_latent_tail_supported(int K5, int E, int L, int lin, int n1, int variational, int dtype) {
    if (dtype != VV_BF16) return 0;
    if (K5 < 64 || K5 % 8 || E % 4 || E > 512 || L % 32 || L > 256 || lin % 32 || lin > 1024 || n1 % 16) return 0;
    if (variational ? E != 2 * L : E != L) return 0;
    return 1;
}

VV_EXPORT size_t vv_latent_tail_workspace_bytes(int batch, int K5, int E, int n1) {
    if (batch <= 0 || K5 <= 0 || E <= 0) return 0;
    return lt_plan(batch, K5, E, n1).ws;
}

VV_EXPORT int vv_latent_tail_fwd(const void *h, const void *w5, const float *e5_scale, const float *eps, const void *wd, const float *scale_d,
                                 const float *shift_d, const void *w1, const float *scale_1, const float *shift_1, float *enc_out, float *z,
                                 void *z_act, float *kl, void *h1, int batch, int K5, int E, int L, int lin, int n1, int variational, int act,
                                 int dtype, void *workspace, size_t workspace_bytes, void *stream) {
    if (!h || !w5 || !wd || !w1 || !z || !h1) return VV_ERR_NULL;
    if (variational && !eps) return VV_ERR_NULL;
    if (batch <= 0 || !vv_latent_tail_supported(K5, E, L, lin, n1, variational, dtype)) return VV_ERR_SHAPE;
    if ((size_t)batch * K5 * 2 >= 0xFFFFFFF0ull || (size_t)E * K5 * 2 >= 0xFFFFFFF0ull) return VV_ERR_SHAPE;
    if (!vv_aligned16(h) || !vv_aligned16(w5) || !vv_aligned16(wd) || !vv_aligned16(w1) || !vv_aligned16(h1) || !vv_aligned16(z) ||
        (enc_out && !vv_aligned16(enc_out)))
        return VV_ERR_ALIGN;
    const LtPlan p = lt_plan(batch, K5, E, n1);
    if (!workspace || workspace_bytes < p.ws || !vv_aligned16(workspace)) return VV_ERR_WORKSPACE;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    static const bool attr = [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&lt_e5_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LT_E5_LDS);
        return true;
    }();
    (void)attr;
    LtArgs a;
    a.h = h; a.w5 = w5; a.slabs = reinterpret_cast<float *>(workspace);
    a.batch = batch; a.K5 = K5; a.E = E; a.kslice = p.kslice; a.nslice = p.nslice;
    a.h_bytes = (unsigned)((size_t)batch * K5 * 2); a.w_bytes = (unsigned)((size_t)E * K5 * 2);
    const int ntn = (E + LT_BN - 1) / LT_BN, ntm = (batch + LT_BM - 1) / LT_BM;
    VV_LAUNCH(lt_e5_kernel, dim3(ntn * ntm * p.nslice), dim3(256), LT_E5_LDS, st, a);
    int rc = vv_launch_status();
    if (rc != VV_OK) return rc;
    return lt_launch_mid(a.slabs, p.nslice, p.nq, e5_scale, eps, wd, scale_d, shift_d, w1, scale_1, shift_1, enc_out, z, z_act, kl, h1, batch, E, L, lin,
                         n1, variational, act, st);
}

// ---- the 4^3 -> 2^3 convolution + the latent tail in three launches: pg_kernel<0> (slabs) -> lt_e5x_kernel -> lt_mid_kernel
VV_EXPORT int vv_conv_pos_latent_tail_supported(int cin4, int cout4, int E, int L, int lin, int n1, int variational, int dtype) {
    if (!vv_latent_tail_supported(8 * cout4, E, L, lin, n1, variational, dtype)) return 0;
    if (cout4 % 256 || E % 16 || E > 128 || 8 * (cout4 / 256) > 64) return 0;
    return vv_conv3d_k4s2_pos_supported(4, cin4, cout4, dtype);
}

VV_EXPORT size_t vv_conv_pos_latent_tail_workspace_bytes(int batch, int cin4, int cout4, int E) {
    const size_t sl = vv_pg_conv_slab_bytes(batch, cin4, cout4);
    if (!sl || E <= 0 || cout4 % 256) return 0;
    return sl + (size_t)8 * (cout4 / 256) * batch * E * sizeof(float);
}

VV_EXPORT int vv_conv_pos_latent_tail_fwd(const void *x4, const void *w4_skip, const float *scale4, const float *shift4, int cin4, int cout4,
                                          const void *w5, const float *e5_scale, const float *eps, const void *wd, const float *scale_d,
                                          const float *shift_d, const void *w1, const float *scale_1, const float *shift_1, float *enc_out,
                                          float *z, void *z_act, float *kl, void *h1, int batch, int E, int L, int lin, int n1, int variational,
                                          int act, int dtype, void *workspace, size_t workspace_bytes, void *stream) {
    if (!x4 || !w4_skip || !w5 || !wd || !w1 || !z || !h1) return VV_ERR_NULL;
    if (variational && !eps) return VV_ERR_NULL;
    if (batch <= 0 || !vv_conv_pos_latent_tail_supported(cin4, cout4, E, L, lin, n1, variational, dtype)) return VV_ERR_SHAPE;
    if ((size_t)E * 8 * cout4 * 2 >= 0xFFFFFFF0ull) return VV_ERR_SHAPE;
    if (!vv_aligned16(x4) || !vv_aligned16(w4_skip) || !vv_aligned16(w5) || !vv_aligned16(wd) || !vv_aligned16(w1) || !vv_aligned16(h1) ||
        !vv_aligned16(z) || (enc_out && !vv_aligned16(enc_out)) || (scale4 && !vv_aligned16(scale4)) || (shift4 && !vv_aligned16(shift4)))
        return VV_ERR_ALIGN;
    const size_t sl = vv_pg_conv_slab_bytes(batch, cin4, cout4);
    const int nh = cout4 / 256, nslice = 8 * nh;
    if (!sl) return VV_ERR_SHAPE;
    if (!workspace || !vv_aligned16(workspace) || workspace_bytes < sl + (size_t)nslice * batch * E * sizeof(float)) return VV_ERR_WORKSPACE;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    VvPgSlabPlan plan;
    int rc = vv_pg_conv_slabs(x4, w4_skip, batch, cin4, cout4, workspace, sl, st, &plan);
    if (rc != VV_OK) return rc;
    LtE5xArgs a;
    a.pslabs = reinterpret_cast<const float *>(workspace);
    a.scale4 = scale4; a.shift4 = shift4; a.w5 = w5;
    a.slabs = reinterpret_cast<float *>(reinterpret_cast<char *>(workspace) + sl);
    a.batch = batch; a.cout4 = cout4; a.E = E; a.nh = nh; a.mtiles = plan.mtiles; a.rows_per_tile = plan.rows_per_tile; a.ntn = plan.ntn; a.act = act;
    for (int p = 0; p < 8; ++p) { a.nsplit[p] = plan.nsplit[p]; a.first[p] = plan.first[p]; }
    const dim3 grid(((batch + 15) / 16) * nslice);
    switch (act) {
        case VV_ACT_ELU: VV_LAUNCH(lt_e5x_kernel<VV_ACT_ELU>, grid, dim3(256), 0, st, a); break;
        case VV_ACT_RELU: VV_LAUNCH(lt_e5x_kernel<VV_ACT_RELU>, grid, dim3(256), 0, st, a); break;
        case VV_ACT_LRELU: VV_LAUNCH(lt_e5x_kernel<VV_ACT_LRELU>, grid, dim3(256), 0, st, a); break;
        default: VV_LAUNCH(lt_e5x_kernel<VV_ACT_NONE>, grid, dim3(256), 0, st, a); break;
    }
    rc = vv_launch_status();
    if (rc != VV_OK) return rc;
    int nq = 1;
    for (int c = 16; c >= 2; c >>= 1)
        if (n1 % (16 * c) == 0) { nq = c; break; }
    return lt_launch_mid(a.slabs, nslice, nq, e5_scale, eps, wd, scale_d, shift_d, w1, scale_1, shift_1, enc_out, z, z_act, kl, h1, batch, E, L, lin, n1,
                         variational, act, st);
}

// Transposed convolution 8^3 x 128 -> 16^3 x 64 with one WHOLE SAMPLE resident in LDS (bf16, gfx950).
//
// Conv3DTranspose k4 s2 SAME (autoencoder3D.py:41-54) in output-parity form: voxel 2m + p of the output reads the input
// cells m + p - a, a in {0,1}^3, through tap t = 1 - p + 2a (per axis).  convt_direct.hip keeps a (4+2) x (4+2) x (8+2) halo
// tile of 128 cells in LDS and streams the parity's weights into registers, one wave per parity: every workgroup pulls the
// whole 1 MiB weight tensor through its CU for 128 cells (1 GiB of L2 -> CU traffic per launch at batch 256, 256 B per
// MFMA per wave), and that instruction stream, not the MFMA pipe, sets its pace.  The input of this layer is only
// 8^3 x 256 B = 128 KiB per sample: here the workgroup IS the sample.  No halo (out-of-grid taps read a zero row), the
// weights cross the CU once per 512 cells (4x fewer bytes), and because all eight waves now work on the same parity they
// share each weight chunk through an LDS ring exactly as conv_direct.hip does.
//
//   workgroup  : 512 threads = 8 waves; wave w owns input plane d = w: 64 cells (2 row tiles of 4 h x 8 w) x 64 channels
//                (2 channel tiles) = 4 accumulators of 32x32; parities are processed one after the other
//   LDS        : x tile [512 voxels][256 B], slot XOR (v & 15) -- conflict-free for every tap shift and every ds_read_b128
//                lane group (a tap shift adds a constant mod 16 to the slot key); weights [64 co][64 ci] per (parity, tap,
//                K half) in a 3-deep ring (slot ^ (co>>1)&7), one 1 KiB LDS-DMA piece per wave per chunk, issued three
//                chunks ahead; a 256-byte zero row that lanes of out-of-grid cells read instead (same slot -> same bank)
//   sync       : one barrier per chunk (16 MFMAs per wave), placed before the chunk's last k-step (conv_direct.hip's scheme)
//   skipping   : a wave whose whole plane is out of the grid for a tap (d = -1 or 8) issues no MFMAs for it
//   epilogue   : folded BN + activation in registers, bf16 pack, v_permlane32_swap pairs -> one 16-byte store per lane and
//                pair of channel groups (guide T21): no LDS staging, so the next parity's weight chunks keep streaming
//   grid       : batch x ps workgroups, ps = 1, 2, 4, 8 splits of the 8 parities (ps > 1 only to fill the chip at batch < 256)
#include <stdlib.h>

#include <type_traits>
#include <utility>

#include "common.h"

namespace {

constexpr int CW_X = 512 * 256;                    // 131,072 B: the sample
constexpr int CW_WST = 64 * 128;                   // one weight chunk: 64 co x 64 ci
constexpr int CW_NST = 3;
constexpr int CW_RING = CW_X;
constexpr int CW_ZERO = CW_RING + CW_NST * CW_WST; // 155,648
constexpr int CW_SS = CW_ZERO + 256;               // scale[64], shift[64]
constexpr int CW_LDS = CW_SS + 512;                // 156,416 B

template <int N>
__device__ __forceinline__ void cw_wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" : : "n"(N) : "memory");
}

#define CW_LD(F, X0, X1, WA)                                                                                                     \
    asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %5\n\tds_read_b128 %2, %6\n\tds_read_b128 %3, %6 offset:4096"          \
                 : "=&v"(F[0]), "=&v"(F[1]), "=&v"(F[2]), "=&v"(F[3])                                                            \
                 : "v"(X0), "v"(X1), "v"(WA)                                                                                     \
                 : "memory")
#define CW_WAIT(F, N) asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(F[0]), "+v"(F[1]), "+v"(F[2]), "+v"(F[3]) : "n"(N) : "memory")
#define CW_MFMA4(F)                                                                                                              \
    do {                                                                                                                         \
        _Pragma("unroll") for (int nt = 0; nt < 2; ++nt) _Pragma("unroll") for (int mt = 0; mt < 2; ++mt)                        \
            acc[nt][mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8 *>(&F[2 + nt]),                 \
                                                                  *reinterpret_cast<const bf16x8 *>(&F[mt]), acc[nt][mt], 0, 0, 0); \
        __builtin_amdgcn_sched_barrier(0);                                                                                       \
    } while (0)

// w: vv_pack_convT_k4s2_skip's image [parity][tap][K half][64 co][64 ci]
template <int ACT>
__global__ __launch_bounds__(512, 1) void ctw_kernel(const __bf16 *__restrict__ x, const __bf16 *__restrict__ w,
                                                     const float *__restrict__ scale, const float *__restrict__ shift,
                                                     __bf16 *__restrict__ y, int npar) {
    extern __shared__ __attribute__((aligned(256))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ps = 8 / npar;
    const int b = (int)blockIdx.x / ps, p0 = ((int)blockIdx.x % ps) * npar;
    const int NC = npar * 16;                      // chunks of this workgroup
    const unsigned lds0 = (unsigned)(unsigned long long)(lptr_t)smem;

    const u32x4 rsx = vv_make_rsrc(x + (size_t)b * (512 * 128), CW_X);
    const u32x4 rsw = vv_make_rsrc(w, 64 * 128 * 64 * 2);

    // ---- prologue: the sample (16 pieces of 4 voxels per wave, plane by plane), weight chunks 0..2, zero row, folded BN
    {
        const int pos = lane & 15, vsub = lane >> 4;
#pragma unroll 1
        for (int i = 0; i < 16; ++i) {
            const int it = i * 8 + wave;
            const int v = it * 4 + vsub;
            vv_dma16(rsx, (unsigned)(v * 256 + ((pos ^ (v & 15)) << 4)), lds0 + it * 1024);
        }
    }
    // weight pieces: this wave's rows 8 wave .. 8 wave + 7 of every chunk
    const unsigned wv = (unsigned)((wave * 8 + (lane >> 3)) * 128 + (((lane & 7) ^ (((wave * 8 + (lane >> 3)) >> 1) & 7)) << 4));
    auto issue_w = [&](int c, int s) {               // chunk c (clamped: the tail re-fetches the last chunk into a free stage)
        const int cc = c < NC ? c : NC - 1;
        const unsigned soff = (unsigned)__builtin_amdgcn_readfirstlane((p0 * 16 + cc) * CW_WST);
        vv_dma16(rsw, wv, soff, lds0 + CW_RING + s * CW_WST + wave * 1024);
    };
    issue_w(0, 0);
    issue_w(1, 1);
    issue_w(2, 2);
    if (tid < 16) *reinterpret_cast<uint4 *>(smem + CW_ZERO + tid * 16) = uint4{0u, 0u, 0u, 0u};
    if (tid >= 64 && tid < 128) {
        const int ch = tid - 64;
        *reinterpret_cast<float *>(smem + CW_SS + ch * 4) = scale ? scale[ch] : 1.f;
        *reinterpret_cast<float *>(smem + CW_SS + 256 + ch * 4) = shift ? shift[ch] : 0.f;
    }
    cw_wait_vm<0>();
    __syncthreads();

    // ---- consumer addressing
    const int fr = lane & 31, fh = lane >> 5;
    const int mh = fr >> 3, mw = fr & 7;
    const unsigned wl = lds0 + CW_RING + fr * 128 + ((fh ^ ((fr >> 1) & 7)) << 4);     // + stage * CW_WST, ^ (k-step << 5)
    const unsigned R0 = lds0 + (unsigned)((wave * 64 + mh * 8 + mw) << 8);             // this lane's own cell, row tile 0
    const unsigned ZR = lds0 + CW_ZERO;
    const int cb = mh * 8 + mw;
    // x row address ^ slot key of the cells this lane reads for tap A of parity (pd, ph, pw), row tiles 0 / 1.  The slot key
    // of voxel v is v & 15 = (8 zh + zw) & 15, the same for both row tiles (they are 4 h-rows = 32 voxels apart).
    auto tap_setup = [&](auto a_c, int pd, int ph, int pw, unsigned &t0, unsigned &t1) {
        constexpr int A = decltype(a_c)::value;
        const int dd = pd - ((A >> 2) & 1), dh = ph - ((A >> 1) & 1), dw = pw - (A & 1);
        const bool okw = (unsigned)(wave + dd) < 8u && (unsigned)(mw + dw) < 8u;
        const int sft = dd * 64 + dh * 8 + dw;
        const unsigned key = ((unsigned)((cb + dh * 8 + dw) ^ fh) & 15u) << 4;
        const unsigned r0 = okw && (unsigned)(mh + dh) < 8u ? R0 + (unsigned)(sft << 8) : ZR;
        const unsigned r1 = okw && (unsigned)(mh + 4 + dh) < 8u ? R0 + (unsigned)((sft + 32) << 8) : ZR;
        t0 = r0 ^ key;
        t1 = r1 ^ key;
    };

    f32x16 acc[2][2];                                // [nt][mt]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // ---- epilogue of one parity: lane = cell, registers walk channels (weights-first MFMA)
    auto epilogue = [&](int p) {
        const int pd = (p >> 2) & 1, ph = (p >> 1) & 1, pw = p & 1;
        char *yb = reinterpret_cast<char *>(y) + ((((size_t)b * 16 + 2 * wave + pd) * 16) * 16) * 128 + fh * 16;
        const char *ss = smem + CW_SS + fh * 16;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            u32x2 o[2][4];                           // [mt][g]: bf16 x 4 of channels nt*32 + 8g + 4fh ..
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 sc = *reinterpret_cast<const f32x4 *>(ss + (nt * 32 + 8 * g) * 4);
                const f32x4 sh = *reinterpret_cast<const f32x4 *>(ss + 256 + (nt * 32 + 8 * g) * 4);
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    bf16x4 q;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float t = acc[nt][mt][4 * g + e] * sc[e] + sh[e];
                        if (ACT == VV_ACT_ELU) { const float em = __expf(fminf(t, 0.f)) - 1.f; t = t > 0.f ? t : em; }
                        else if (ACT == VV_ACT_RELU) t = fmaxf(t, 0.f);
                        else if (ACT == VV_ACT_LRELU) t = t > 0.f ? t : 0.3f * t;
                        q[e] = static_cast<__bf16>(t);
                        acc[nt][mt][4 * g + e] = 0.f;
                    }
                    o[mt][g] = *reinterpret_cast<const u32x2 *>(&q);
                }
            }
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                char *row = yb + (size_t)(((2 * (mt * 4 + mh) + ph) * 16 + 2 * mw + pw) * 128) + nt * 64;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    u32x2 lo = o[mt][2 * j], hi = o[mt][2 * j + 1];
                    // lanes 32-63 of `lo` swap with lanes 0-31 of `hi`: the lower half then holds 8 consecutive channels
                    // [own group 2j | upper's group 2j], the upper half [lower's group 2j+1 | own group 2j+1]
                    auto rx = __builtin_amdgcn_permlane32_swap(lo[0], hi[0], false, false);
                    auto ry = __builtin_amdgcn_permlane32_swap(lo[1], hi[1], false, false);
                    *reinterpret_cast<u32x4 *>(row + j * 32) = u32x4{rx[0], ry[0], rx[1], ry[1]};
                }
            }
        }
    };

    // ---- main loop.  One parity = 8 taps x 2 K halves = 16 chunks, unrolled with the tap as a compile-time constant: the
    // scalar / address work per chunk is what a wave pays next to its 16 MFMAs (measured: a generic runtime-(parity, tap)
    // loop spent as long on it as on the MFMAs).
    u32x4 P[4], Q[4];
    unsigned ua0, ua1, ub0, ub1;                     // current tap's / next tap's row keys
    tap_setup(std::integral_constant<int, 0>{}, (p0 >> 2) & 1, (p0 >> 1) & 1, p0 & 1, ua0, ua1);
    unsigned ws = wl;                                // weight fragment base in the current chunk's stage
    int stg = 0;                                     // that stage
    int cw = 3;                                      // next chunk to fetch
    CW_LD(P, ua0, ua1, ws);                          // chunk 0, k-step 0
#pragma unroll 1
    for (int pi = 0; pi < npar; ++pi) {
        const int p = p0 + pi, pd = (p >> 2) & 1, ph = (p >> 1) & 1, pw = p & 1;
        const int pn = p + 1, pnd = (pn >> 2) & 1, pnh = (pn >> 1) & 1, pnw = pn & 1;
        auto chunk = [&](auto j_c) {
            constexpr int J = decltype(j_c)::value, A = J >> 1, KH = J & 1;
            const unsigned x0 = KH ? ua0 ^ 128u : ua0, x1 = KH ? ua1 ^ 128u : ua1;    // K half: k-steps 4..7 are slots 8..15
            CW_LD(Q, x0 ^ 32u, x1 ^ 32u, ws ^ 32u);
            CW_WAIT(P, 4);
            CW_MFMA4(P);
            if (KH == 0) {                           // the next tap's keys, far from the barrier
                if (A < 7) tap_setup(std::integral_constant<int, (A + 1) & 7>{}, pd, ph, pw, ub0, ub1);
                else tap_setup(std::integral_constant<int, 0>{}, pnd, pnh, pnw, ub0, ub1);
            }
            CW_LD(P, x0 ^ 64u, x1 ^ 64u, ws ^ 64u);
            CW_WAIT(Q, 4);
            CW_MFMA4(Q);
            CW_LD(Q, x0 ^ 96u, x1 ^ 96u, ws ^ 96u);
            CW_WAIT(P, 4);
            CW_MFMA4(P);
            CW_WAIT(Q, 0);                           // every LDS read of this chunk has returned: its stage may be refilled
            // The next chunk's piece has landed.  Issued after it: one more piece, and, in the two chunks that follow a
            // parity's epilogue, that epilogue's 8 stores (between the pieces in the in-order counter).  The count is only right
            // while the epilogue issues AT LEAST 8 vector-memory instructions (fewer = this wait no longer covers the piece);
            // tests/test_isa_lint.py pins, on the generated code, how many waits a piece survives before one covers it.
            if (J < 2 && pi > 0) cw_wait_vm<9>();
            else cw_wait_vm<1>();
            __builtin_amdgcn_s_barrier();
            issue_w(cw, stg);
            ++cw;
            stg = stg == CW_NST - 1 ? 0 : stg + 1;
            ws = wl + stg * CW_WST;
            if (KH == 0) {
                CW_LD(P, ua0 ^ 128u, ua1 ^ 128u, ws);
            } else {
                ua0 = ub0;
                ua1 = ub1;
                CW_LD(P, ua0, ua1, ws);
            }
            CW_MFMA4(Q);
        };
        chunk(std::integral_constant<int, 0>{});
        chunk(std::integral_constant<int, 1>{});
        chunk(std::integral_constant<int, 2>{});
        chunk(std::integral_constant<int, 3>{});
        chunk(std::integral_constant<int, 4>{});
        chunk(std::integral_constant<int, 5>{});
        chunk(std::integral_constant<int, 6>{});
        chunk(std::integral_constant<int, 7>{});
        chunk(std::integral_constant<int, 8>{});
        chunk(std::integral_constant<int, 9>{});
        chunk(std::integral_constant<int, 10>{});
        chunk(std::integral_constant<int, 11>{});
        chunk(std::integral_constant<int, 12>{});
        chunk(std::integral_constant<int, 13>{});
        chunk(std::integral_constant<int, 14>{});
        chunk(std::integral_constant<int, 15>{});
        epilogue(p);
    }
    CW_WAIT(P, 0);                                   // the look-ahead reads of the chunk after the last
    cw_wait_vm<0>();                                 // the tail's pieces still target this workgroup's LDS
}


// ---- the same kernel on v_mfma_f32_16x16x32_bf16: same chunk / ring / barrier scheme, x tile slot key (2 zw) & 15 (conflict-free
// for every tap shift in this fragment shape, profiles/microbench/ctw_swz.py; the key v & 15 of the 32x32x16 form is 2-way on odd w shifts); a wave's 64 x 64 tile
// is 4 x 4 accumulators of 16 x 16 (16 cells = two h-rows, 16 channels), 8 fragment reads and 16 MFMAs per 32-deep k-step.
// On random data the chip holds a higher clock on this shape (MI355X_MICROARCH.md, DVFS give-back item 7).
#define CW16_LD(F, XA, WA)                                                                                                          \
    asm volatile("ds_read_b128 %0, %8\n\tds_read_b128 %1, %9\n\tds_read_b128 %2, %10\n\tds_read_b128 %3, %11\n\t"                    \
                 "ds_read_b128 %4, %12\n\tds_read_b128 %5, %12 offset:2048\n\tds_read_b128 %6, %12 offset:4096\n\t"                  \
                 "ds_read_b128 %7, %12 offset:6144"                                                                                 \
                 : "=&v"(F[0]), "=&v"(F[1]), "=&v"(F[2]), "=&v"(F[3]), "=&v"(F[4]), "=&v"(F[5]), "=&v"(F[6]), "=&v"(F[7])            \
                 : "v"(XA[0]), "v"(XA[1]), "v"(XA[2]), "v"(XA[3]), "v"(WA)                                                           \
                 : "memory")
#define CW16_WAIT(F, N)                                                                                                             \
    asm volatile("s_waitcnt lgkmcnt(%8)"                                                                                            \
                 : "+v"(F[0]), "+v"(F[1]), "+v"(F[2]), "+v"(F[3]), "+v"(F[4]), "+v"(F[5]), "+v"(F[6]), "+v"(F[7])                    \
                 : "n"(N)                                                                                                           \
                 : "memory")
#define CW16_SB __builtin_amdgcn_sched_barrier(0)
#define CW16_RD(D, A, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(D) : "v"(A), "n"(OFF) : "memory")
#define CW16_MF(F, COT, CT)                                                                                                         \
    acc[COT][CT] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8 *>(&F[4 + COT]),                          \
                                                           *reinterpret_cast<const bf16x8 *>(&F[CT]), acc[COT][CT], 0, 0, 0)

// STATS (round 4, the training step's forward pass): also leaves the per-workgroup column sums (sum, sum of squares per output channel, of the
// bf16 values it stores) as partial[(workgroup * 2 + {0, 1}) * 64 + channel] -- the layout of bn_reduce_kernel's partials, so
// vv_bn_finalize_stats turns them into the batch statistics and the 134 MB statistics sweep over this layer's output is not run.
template <int ACT, bool STATS = false>
__global__ __launch_bounds__(512, 1) void ctw16_kernel(const __bf16 *__restrict__ x, const __bf16 *__restrict__ w,
                                                       const float *__restrict__ scale, const float *__restrict__ shift,
                                                       __bf16 *__restrict__ y, int npar, float *__restrict__ partial = nullptr) {
    extern __shared__ __attribute__((aligned(256))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ps = 8 / npar;
    const int b = (int)blockIdx.x / ps, p0 = ((int)blockIdx.x % ps) * npar;
    const int NC = npar * 16;
    const unsigned lds0 = (unsigned)(unsigned long long)(lptr_t)smem;

    const u32x4 rsx = vv_make_rsrc(x + (size_t)b * (512 * 128), CW_X);
    const u32x4 rsw = vv_make_rsrc(w, 64 * 128 * 64 * 2);
    {
        const int pos = lane & 15, vsub = lane >> 4;
#pragma unroll 1
        for (int i = 0; i < 16; ++i) {
            const int it = i * 8 + wave;
            const int v = it * 4 + vsub;
            vv_dma16(rsx, (unsigned)(v * 256 + ((pos ^ ((2 * v) & 15)) << 4)), lds0 + it * 1024);    // slot key (2 zw) & 15, zw = v & 7
        }
    }
    const unsigned wv = (unsigned)((wave * 8 + (lane >> 3)) * 128 + (((lane & 7) ^ (((wave * 8 + (lane >> 3)) >> 1) & 7)) << 4));
    auto issue_w = [&](int c, int s) {
        const int cc = c < NC ? c : NC - 1;
        const unsigned soff = (unsigned)__builtin_amdgcn_readfirstlane((p0 * 16 + cc) * CW_WST);
        vv_dma16(rsw, wv, soff, lds0 + CW_RING + s * CW_WST + wave * 1024);
    };
    issue_w(0, 0);
    issue_w(1, 1);
    issue_w(2, 2);
    if (tid < 16) *reinterpret_cast<uint4 *>(smem + CW_ZERO + tid * 16) = uint4{0u, 0u, 0u, 0u};
    if (tid >= 64 && tid < 128) {
        const int ch = tid - 64;
        *reinterpret_cast<float *>(smem + CW_SS + ch * 4) = scale ? scale[ch] : 1.f;
        *reinterpret_cast<float *>(smem + CW_SS + 256 + ch * 4) = shift ? shift[ch] : 0.f;
    }
    cw_wait_vm<0>();
    __syncthreads();

    // ---- consumer addressing: lane = (r, q): row r of a 16-row fragment, k quarter q
    const int r = lane & 15, q = lane >> 4;
    const int rh = r >> 3, rw = r & 7;
    const unsigned wl = lds0 + CW_RING + r * 128 + ((q ^ ((r >> 1) & 7)) << 4);        // + stage, + 2048 cot, ^ (k-step << 6)
    const unsigned R0 = lds0 + (unsigned)((wave * 64 + r) << 8);                       // this lane's own cell, cell tile 0
    const unsigned ZR = lds0 + CW_ZERO;
    // x row address ^ slot key of the 4 cells (cell tiles 0..3: h-rows 2 ct, 2 ct + 1) this lane reads for tap A
    auto tap_setup = [&](auto a_c, int pd, int ph, int pw, unsigned (&t)[4]) {
        constexpr int A = decltype(a_c)::value;
        const int dd = pd - ((A >> 2) & 1), dh = ph - ((A >> 1) & 1), dw = pw - (A & 1);
        const bool okw = (unsigned)(wave + dd) < 8u && (unsigned)(rw + dw) < 8u;
        const int sft = dd * 64 + dh * 8 + dw;
        const unsigned key = ((unsigned)((2 * (rw + dw)) ^ q) & 15u) << 4;
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
            const unsigned row = okw && (unsigned)(2 * ct + rh + dh) < 8u ? R0 + (unsigned)((sft + 16 * ct) << 8) : ZR;
            t[ct] = row ^ key;
        }
    };

    f32x4 acc[4][4];                                 // [cot][ct]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 st1[4], st2[4];                            // STATS: this lane's running sum / sum of squares of channels 16 cot + 4 q .. + 3
#pragma unroll
    for (int i = 0; i < 4; ++i) { st1[i] = f32x4{0.f, 0.f, 0.f, 0.f}; st2[i] = f32x4{0.f, 0.f, 0.f, 0.f}; }

    // ---- epilogue of one parity: lane (r, q) holds channels 16 cot + 4 q .. + 3 of cell 16 ct + r.  v_permlane16_swap of
    // the channel tiles (2c, 2c+1) gives every lane 8 consecutive channels: q = 0: tile 2c ch 0..7, 1: tile 2c+1 ch 0..7,
    // 2: tile 2c ch 8..15, 3: tile 2c+1 ch 8..15 -> one 16-byte store per lane and tile pair.
    auto epilogue = [&](int p) {
        const int pd = (p >> 2) & 1, ph = (p >> 1) & 1, pw = p & 1;
        const int qo = (q & 1) * 32 + (q >> 1) * 16;
        char *yb = reinterpret_cast<char *>(y) + ((((size_t)b * 16 + 2 * wave + pd) * 16 + 2 * rh + ph) * 16 + 2 * rw + pw) * 128 + qo;
        const char *ss = smem + CW_SS + q * 16;
#pragma unroll
        for (int c2 = 0; c2 < 2; ++c2) {
            u32x2 o[2][4];                           // [cot & 1][ct]
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int cot = 2 * c2 + h;
                const f32x4 sc = *reinterpret_cast<const f32x4 *>(ss + cot * 64);
                const f32x4 sh = *reinterpret_cast<const f32x4 *>(ss + 256 + cot * 64);
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) {
                    const f32x4 t = vv_bn_act4<ACT>(acc[cot][ct], sc, sh);
                    bf16x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = static_cast<__bf16>(t[e]);
                    if (STATS) {                     // of the ROUNDED values: what the BatchNorm that follows normalises
                        const f32x4 f = {(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
                        st1[cot] += f;
                        st2[cot] += f * f;
                    }
                    acc[cot][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
                    o[h][ct] = *reinterpret_cast<const u32x2 *>(&v);
                }
            }
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) {
                // odd 16-lane rows of the first operand swap with the even rows of the second
                auto rx = __builtin_amdgcn_permlane16_swap(o[0][ct][0], o[1][ct][0], false, false);
                auto ry = __builtin_amdgcn_permlane16_swap(o[0][ct][1], o[1][ct][1], false, false);
                *reinterpret_cast<u32x4 *>(yb + (size_t)(ct * 4 * 16 * 128) + c2 * 64) = u32x4{rx[0], ry[0], rx[1], ry[1]};
            }
        }
    };

    u32x4 P[8], Q[8];
    unsigned ua[4], ub[4];
    tap_setup(std::integral_constant<int, 0>{}, (p0 >> 2) & 1, (p0 >> 1) & 1, p0 & 1, ua);
    unsigned ws = wl;
    int stg = 0, cw = 3;
    CW16_LD(P, ua, ws);
#pragma unroll 1
    for (int pi = 0; pi < npar; ++pi) {
        const int p = p0 + pi, pd = (p >> 2) & 1, ph = (p >> 1) & 1, pw = p & 1;
        const int pn = p + 1, pnd = (pn >> 2) & 1, pnh = (pn >> 1) & 1, pnw = pn & 1;
        // One chunk = two 32-deep k-steps of 16 MFMAs.  The fragment reads of the NEXT k-step, the chunk's LDS-DMA piece and the
        // address arithmetic sit in the gaps between this k-step's MFMAs (two per gap), not in a burst ahead of them: with the
        // burst form both waves of a SIMD left the barrier into ~250 cycles of memory-instruction issue with the matrix pipe idle
        // (in-kernel stamps, profiles/microbench/mb_ctw_stamp.py: 1869 cycles per chunk against 1024 of MFMA work).
        auto chunk = [&](auto j_c) {
            constexpr int J = decltype(j_c)::value, A = J >> 1, KH = J & 1;
            unsigned xq[4];                          // second 32-deep k-step of this K half: slots +4
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) xq[ct] = ua[ct] ^ (KH ? 192u : 64u);
            const unsigned wq = ws ^ 64u;
            CW16_WAIT(P, 0);
            CW16_SB;
            CW16_MF(P, 0, 0); CW16_MF(P, 0, 1); CW16_SB; CW16_RD(Q[0], xq[0], 0); CW16_RD(Q[4], wq, 0); CW16_SB;
            CW16_MF(P, 0, 2); CW16_MF(P, 0, 3); CW16_SB; CW16_RD(Q[1], xq[1], 0); CW16_RD(Q[5], wq, 2048); CW16_SB;
            CW16_MF(P, 1, 0); CW16_MF(P, 1, 1); CW16_SB; CW16_RD(Q[2], xq[2], 0); CW16_RD(Q[6], wq, 4096); CW16_SB;
            CW16_MF(P, 1, 2); CW16_MF(P, 1, 3); CW16_SB; CW16_RD(Q[3], xq[3], 0); CW16_RD(Q[7], wq, 6144); CW16_SB;
            CW16_MF(P, 2, 0); CW16_MF(P, 2, 1); CW16_MF(P, 2, 2); CW16_MF(P, 2, 3);
            if (KH == 0) {                           // next tap's row addresses: VALU in the shadow of the MFMAs around it
                if (A < 7) tap_setup(std::integral_constant<int, (A + 1) & 7>{}, pd, ph, pw, ub);
                else tap_setup(std::integral_constant<int, 0>{}, pnd, pnh, pnw, ub);
            }
            CW16_MF(P, 3, 0); CW16_MF(P, 3, 1); CW16_MF(P, 3, 2); CW16_MF(P, 3, 3);
            CW16_SB;
            CW16_WAIT(Q, 0);
            if (J < 2 && pi > 0) cw_wait_vm<9>();
            else cw_wait_vm<1>();
            __builtin_amdgcn_s_barrier();
            CW16_SB;
            CW16_MF(Q, 0, 0); CW16_MF(Q, 0, 1); CW16_SB;
            issue_w(cw, stg);
            ++cw;
            stg = stg == CW_NST - 1 ? 0 : stg + 1;
            ws = wl + stg * CW_WST;
            if (KH == 0) {
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) xq[ct] = ua[ct] ^ 128u;
            } else {
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) xq[ct] = ua[ct] = ub[ct];
            }
            CW16_SB;
            CW16_MF(Q, 0, 2); CW16_MF(Q, 0, 3); CW16_SB; CW16_RD(P[0], xq[0], 0); CW16_RD(P[4], ws, 0); CW16_SB;
            CW16_MF(Q, 1, 0); CW16_MF(Q, 1, 1); CW16_SB; CW16_RD(P[1], xq[1], 0); CW16_RD(P[5], ws, 2048); CW16_SB;
            CW16_MF(Q, 1, 2); CW16_MF(Q, 1, 3); CW16_SB; CW16_RD(P[2], xq[2], 0); CW16_RD(P[6], ws, 4096); CW16_SB;
            CW16_MF(Q, 2, 0); CW16_MF(Q, 2, 1); CW16_SB; CW16_RD(P[3], xq[3], 0); CW16_RD(P[7], ws, 6144); CW16_SB;
            CW16_MF(Q, 2, 2); CW16_MF(Q, 2, 3);
            CW16_MF(Q, 3, 0); CW16_MF(Q, 3, 1); CW16_MF(Q, 3, 2); CW16_MF(Q, 3, 3);
            CW16_SB;
        };
        chunk(std::integral_constant<int, 0>{});
        chunk(std::integral_constant<int, 1>{});
        chunk(std::integral_constant<int, 2>{});
        chunk(std::integral_constant<int, 3>{});
        chunk(std::integral_constant<int, 4>{});
        chunk(std::integral_constant<int, 5>{});
        chunk(std::integral_constant<int, 6>{});
        chunk(std::integral_constant<int, 7>{});
        chunk(std::integral_constant<int, 8>{});
        chunk(std::integral_constant<int, 9>{});
        chunk(std::integral_constant<int, 10>{});
        chunk(std::integral_constant<int, 11>{});
        chunk(std::integral_constant<int, 12>{});
        chunk(std::integral_constant<int, 13>{});
        chunk(std::integral_constant<int, 14>{});
        chunk(std::integral_constant<int, 15>{});
        epilogue(p);
    }
    CW16_WAIT(P, 0);
    cw_wait_vm<0>();
    if (STATS) {
        // lanes (r, q), r = 0..15, hold the same channels for different cells: sum over r, then over the waves through LDS (free now)
#pragma unroll
        for (int cot = 0; cot < 4; ++cot)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float a = st1[cot][e], b2 = st2[cot][e];
#pragma unroll
                for (int o = 8; o > 0; o >>= 1) { a += __shfl_xor(a, o, 64); b2 += __shfl_xor(b2, o, 64); }
                st1[cot][e] = a; st2[cot][e] = b2;
            }
        __syncthreads();                             // every wave has left the main loop: the sample's tile is dead
        float *red = reinterpret_cast<float *>(smem);    // [8 waves][2][64]
        if (r == 0) {
#pragma unroll
            for (int cot = 0; cot < 4; ++cot) {
                *reinterpret_cast<f32x4 *>(red + (wave * 2 + 0) * 64 + 16 * cot + 4 * q) = st1[cot];
                *reinterpret_cast<f32x4 *>(red + (wave * 2 + 1) * 64 + 16 * cot + 4 * q) = st2[cot];
            }
        }
        __syncthreads();
        if (tid < 128) {
            const int which = tid >> 6, ch = tid & 63;
            float sacc = 0.f;
#pragma unroll
            for (int wv_ = 0; wv_ < 8; ++wv_) sacc += red[(wv_ * 2 + which) * 64 + ch];      // wave order: deterministic
            partial[((size_t)blockIdx.x * 2 + which) * 64 + ch] = sacc;
        }
    }
}


// ---- round 4: FOUR waves, one per SIMD, 128 cells x 64 channels each, two accumulator sets (`ctw4_kernel`).
// In-kernel stamps on ctw16_kernel put a workgroup at 207 k cycles against 131 k of MFMA work: ~35 k are the eight
// parity epilogues (VALU with the matrix pipe idle: at two waves per SIMD an epilogue slice only displaced the partner wave's MFMAs), the
// rest barrier skew and the LDS read bursts of two waves that leave every barrier in phase.  Here a SIMD holds ONE wave, so what sits
// between two of its MFMAs is its own to fill: the wave owns input planes 2w, 2w + 1 (8 cell tiles x 4 channel tiles = 32 accumulators of
// 16 x 16), reads 12 fragments per 32 MFMAs instead of 16, and keeps TWO accumulator sets (256 registers of the unified 512) so that the
// BatchNorm + activation + pack + store of parity p runs in the gaps between the MFMAs of parity p + 1 -- one output unit (two channel
// tiles of one cell tile: 8 values, one 16-byte store) per chunk, cut into pieces of one to three VALU instructions, one piece per gap.
// LDS images, slot keys, ring, barrier placement and the per-accumulator summation order are ctw16_kernel's: the outputs are the same bits.
// C4_ABL (diagnostic builds only, profiles/microbench/c4_ablate.py; 0 in the tree's build): 1 no epilogue pieces in the loop, 2 no
// barrier, 4 no weight LDS-DMA in the loop, 8 no fragment reads in the loop, 16 no MFMAs, 32 no activation math (stages 1..5), 64 no
// accumulator reads / packs / swaps / stores in the loop, 128 no v_exp -- wrong results, timing only
#ifndef C4_ABL
#define C4_ABL 0
#endif
template <int N, class Fn, int... I>
__device__ __forceinline__ void c4_static_for_impl(Fn &&fn, std::integer_sequence<int, I...>) {
    (fn(std::integral_constant<int, I>{}), ...);
}
template <int N, class Fn>
__device__ __forceinline__ void c4_static_for(Fn &&fn) {
    c4_static_for_impl<N>(fn, std::make_integer_sequence<int, N>{});
}
#define C4_SB __builtin_amdgcn_sched_barrier(0)
#define C4_RD(D, A, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(D) : "v"(A), "n"(OFF) : "memory")
#define C4_WAIT(F)                                                                                                                  \
    asm volatile("s_waitcnt lgkmcnt(0)"                                                                                             \
                 : "+v"(F[0]), "+v"(F[1]), "+v"(F[2]), "+v"(F[3]), "+v"(F[4]), "+v"(F[5]), "+v"(F[6]), "+v"(F[7]), "+v"(F[8]),       \
                   "+v"(F[9]), "+v"(F[10]), "+v"(F[11])                                                                             \
                 :                                                                                                                  \
                 : "memory")

template <int ACT>
__global__ __launch_bounds__(256, 1) void ctw4_kernel(const __bf16 *__restrict__ x, const __bf16 *__restrict__ w,
                                                      const float *__restrict__ scale, const float *__restrict__ shift,
                                                      __bf16 *__restrict__ y, int npar) {
    extern __shared__ __attribute__((aligned(256))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);          // 0..3: input planes 2 wave, 2 wave + 1
    const int ps = 8 / npar;
    const int b = (int)blockIdx.x / ps, p0 = ((int)blockIdx.x % ps) * npar;
    const int NC = npar * 16;
    const unsigned lds0 = (unsigned)(unsigned long long)(lptr_t)smem;

    const u32x4 rsx = vv_make_rsrc(x + (size_t)b * (512 * 128), CW_X);
    const u32x4 rsw = vv_make_rsrc(w, 64 * 128 * 64 * 2);
    const __amdgpu_buffer_rsrc_t rsy = __builtin_amdgcn_make_buffer_rsrc(y + (size_t)b * (4096 * 64), 0, 4096 * 64 * 2, 0x00020000);
    {   // the sample: 128 pieces of 4 voxels, 32 per wave
        const int pos = lane & 15, vsub = lane >> 4;
#pragma unroll 1
        for (int i = 0; i < 32; ++i) {
            const int it = i * 4 + wave;
            const int v = it * 4 + vsub;
            vv_dma16(rsx, (unsigned)(v * 256 + ((pos ^ ((2 * v) & 15)) << 4)), lds0 + it * 1024);
        }
    }
    // weight chunk [64 co][64 ci] = 8 pieces of 8 rows: this wave's pieces 2 wave, 2 wave + 1
    const int wr0 = wave * 16 + (lane >> 3), wr1 = wr0 + 8;
    const unsigned wv0 = (unsigned)(wr0 * 128 + (((lane & 7) ^ ((wr0 >> 1) & 7)) << 4));
    const unsigned wv1 = (unsigned)(wr1 * 128 + (((lane & 7) ^ ((wr1 >> 1) & 7)) << 4));
    auto issue_w = [&](int c, int s) {
        const int cc = c < NC ? c : NC - 1;                             // (the tail re-fetches the last chunk into a free stage)
        const unsigned soff = (unsigned)__builtin_amdgcn_readfirstlane((p0 * 16 + cc) * CW_WST);
        vv_dma16(rsw, wv0, soff, lds0 + CW_RING + s * CW_WST + wave * 2048);
        vv_dma16(rsw, wv1, soff, lds0 + CW_RING + s * CW_WST + wave * 2048 + 1024);
    };
    issue_w(0, 0);
    issue_w(1, 1);
    issue_w(2, 2);
    if (tid < 16) *reinterpret_cast<uint4 *>(smem + CW_ZERO + tid * 16) = uint4{0u, 0u, 0u, 0u};

    // ---- consumer addressing: lane = (r, q): row r of a 16-row fragment, k quarter q (as ctw16_kernel)
    const int r = lane & 15, q = lane >> 4;
    const int rh = r >> 3, rw = r & 7;
    if (tid >= 64 && tid < 128) {
        const int ch = tid - 64;
        *reinterpret_cast<float *>(smem + CW_SS + ch * 4) = scale ? scale[ch] : 1.f;
        *reinterpret_cast<float *>(smem + CW_SS + 256 + ch * 4) = shift ? shift[ch] : 0.f;
    }
    cw_wait_vm<0>();
    __syncthreads();
    // folded BatchNorm of this lane's channels 16 cot + 4 q .. + 3 for the channel-tile pair the epilogue is working on (cot = 2 c2,
    // 2 c2 + 1): re-read from LDS when the pair changes (twice per parity), by reads that the next k-step boundary's wait covers
    u32x4 bss[4];                                    // scale[2 c2], scale[2 c2 + 1], shift[2 c2], shift[2 c2 + 1]
    const unsigned ssa = lds0 + CW_SS + q * 16;
    auto load_ss = [](auto c2_c, u32x4(&d)[4], unsigned a) {     // (asm operands must be parameters of a generic lambda, not captures)
        constexpr int C2 = decltype(c2_c)::value;
        C4_RD(d[0], a, C2 * 128);
        C4_RD(d[1], a, C2 * 128 + 64);
        C4_RD(d[2], a, 256 + C2 * 128);
        C4_RD(d[3], a, 256 + C2 * 128 + 64);
    };
    load_ss(std::integral_constant<int, 0>{}, bss, ssa);
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bss[0]), "+v"(bss[1]), "+v"(bss[2]), "+v"(bss[3]) : : "memory");

    const unsigned wl = lds0 + CW_RING + r * 128 + ((q ^ ((r >> 1) & 7)) << 4);        // + stage, + 2048 cot, ^ (k-step << 6)
    const unsigned R0 = lds0 + (unsigned)((wave * 128 + r) << 8);                      // this lane's own cell, cell tile 0
    const unsigned ZR = lds0 + CW_ZERO;
    // ---- fragment row addresses of a tap.  Tile ct = plane 2 wave + (ct >> 2), h-rows 2 (ct & 3) + rh; tap A = (ad, ah, aw) of parity
    // (pd, ph, pw) reads cell + (dd, dh, dw), d* = p* - a*.  What depends on the lane is small and fixed per PARITY: the w shift has two
    // values (key and validity of column rw + dw), and an h shift can leave the grid only in one tile class (dh = -1 in tiles with
    // (ct & 3) == 0, lanes rh == 0; dh = +1 in tiles with (ct & 3) == 3, lanes rh == 1).  So a parity prepares, per aw, R0 ^ key,
    // ZR ^ key and two lane predicates (the compiler keeps them as scalar masks); a (tap, tile) is then ONE add of a wave-uniform
    // byte shift and ONE select.
    unsigned tr0[2], tzr[2];
    bool tokw[2], tokwh[2];
    auto parity_setup = [&](int ph, int pw) {
#pragma unroll
        for (int aw = 0; aw < 2; ++aw) {
            const int dw = pw - aw;
            const unsigned key = ((unsigned)((2 * (rw + dw)) ^ q) & 15u) << 4;
            tr0[aw] = R0 ^ key;
            tzr[aw] = ZR ^ key;
            tokw[aw] = (unsigned)(rw + dw) < 8u;
            tokwh[aw] = tokw[aw] & (ph ? rh == 0 : rh == 1);      // ... and the lane's h-row stays inside in the special tile class
        }
    };
    auto tap_setup2 = [&](auto a_c, auto ct0_c, int pd, int ph, int pw, unsigned (&t)[8]) {
        constexpr int A = decltype(a_c)::value, CT0 = decltype(ct0_c)::value, AD = (A >> 2) & 1, AH = (A >> 1) & 1, AW = A & 1;
        const int dd = pd - AD, dh = ph - AH, dw = pw - AW;
        const int sft = (dd * 64 + dh * 8 + dw) << 8;                     // wave-uniform
#pragma unroll
        for (int ct = CT0; ct < CT0 + 2; ++ct) {
            const bool plane = (unsigned)(2 * wave + (ct >> 2) + dd) < 8u;                                        // wave-uniform
            const bool special = ph ? (AH == 0 && (ct & 3) == 3) : (AH == 1 && (ct & 3) == 0);                    // wave-uniform
            const bool ok = (special ? tokwh[AW] : tokw[AW]) & plane;
            t[ct] = ok ? tr0[AW] + (unsigned)(sft + ct * 4096) : tzr[AW];
        }
    };
    auto tap_setup = [&](auto a_c, int pd, int ph, int pw, unsigned (&t)[8]) {
        tap_setup2(a_c, std::integral_constant<int, 0>{}, pd, ph, pw, t);
        tap_setup2(a_c, std::integral_constant<int, 2>{}, pd, ph, pw, t);
        tap_setup2(a_c, std::integral_constant<int, 4>{}, pd, ph, pw, t);
        tap_setup2(a_c, std::integral_constant<int, 6>{}, pd, ph, pw, t);
    };

    f32x4 accA[4][8], accB[4][8];                    // [cot][ct]; a parity accumulates into one set while the other is written out
    u32x4 P[12], Q[12];                              // fragments: [0..7] x cell tiles, [8..11] weight channel tiles
    unsigned ua[8], ub[8];
    unsigned ws = wl;
    int stg = 0, cw = 3;

    // ---- the epilogue of one output unit U = (c2, ct) of the PREVIOUS parity (accumulator set PREV) as a program of 32 GAPS, two
    // INDEPENDENT instructions per gap.  One wave alone overlaps about two simple VALU instructions with a 16x16x32 MFMA, and a dependent
    // pair in one gap stalls (profiles/microbench/issue_model.hip: 2 v_add per gap free, the third + 4 cycles, a v_xor feeding the
    // ds_read behind it + 9).  So the unit's two accumulators (h = 0, 1: channel tiles 2 c2 + h) are two streams that advance one
    // stage per gap, side by side; a stream's next stage reads what its previous gap wrote.  Element e of stream h is value e of
    // accumulator [2 c2 + h][ct]: channel 16 (2 c2 + h) + 4 q + e of cell 16 ct + r.  Stages of an element (the arithmetic of
    // vv_bn_act4, in its order): fma | min | * log2 e | exp2 | - 1 | max | +.  Gaps 0..27: 4 elements x 7 stages; 28, 29: bf16 packs;
    // 30: the two lane swaps; 31: the 16-byte store.
    // The accumulators live in AGPRs (both sets: all 256 of them) and come out through v_accvgpr_read, written as asm with an "a"
    // operand: left to itself the allocator keeps one set in VGPRs (VALU instructions read it directly) and spills what no longer fits.
    // A unit's 8 values are read in the pack / swap / store gaps of the unit BEFORE it (2 per gap); unit 0 reads its own in stage 0.
    float st_t[2] = {0.f, 0.f}, st_m[2] = {0.f, 0.f}, st_u[2] = {0.f, 0.f}, st_r[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    float av[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#define C4_ACCRD(D, SRC) asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(D) : "a"(SRC))
    u32x2 eo[2] = {u32x2{0u, 0u}, u32x2{0u, 0u}};
    unsigned yvo = 0;                                // this lane's byte offset inside the sample's output for the previous parity
    auto epi_gap = [&](auto &PREV, auto u_c, auto g_c, auto tail_c) {
        constexpr int U = decltype(u_c)::value, G = decltype(g_c)::value, C2 = U >> 3, CT = U & 7;
        constexpr bool TAIL = decltype(tail_c)::value;      // the last parity's epilogue, on its own after the loop
        if constexpr (G < 28 && (C4_ABL & 512)) {    // diagnostic: the same number of VALU instructions as plain asm v_fma on two private registers
            constexpr int E = G / 7, S = G % 7;
            if constexpr (S == 0) { st_r[0][E] = av[0][E]; st_r[1][E] = av[1][E]; }
            asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(st_t[0]) : "v"(st_u[0]), "v"(st_u[1]));
            asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(st_t[1]) : "v"(st_u[0]), "v"(st_u[1]));
        } else
        if constexpr (G < 28) {
            constexpr int E = G / 7, S = G % 7;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                if constexpr (S == 0) {
                    if constexpr (U == 0) C4_ACCRD(av[h][E], PREV[2 * C2 + h][CT][E]);
                    st_t[h] = __builtin_fmaf(av[h][E], __builtin_bit_cast(f32x4, bss[h])[E], __builtin_bit_cast(f32x4, bss[2 + h])[E]);
                } else if constexpr (ACT == VV_ACT_ELU && !((C4_ABL & 32) && S < 6) && !((C4_ABL & 128) && S == 3)) {
                    if constexpr (S == 1) st_m[h] = fminf(st_t[h], 0.f);
                    else if constexpr (S == 2) st_m[h] = st_m[h] * 1.4426950408889634f;
                    else if constexpr (S == 3) st_m[h] = __builtin_amdgcn_exp2f(st_m[h]);
                    else if constexpr (S == 4) st_m[h] = st_m[h] - 1.f;
                    else if constexpr (S == 5) st_u[h] = fmaxf(st_t[h], 0.f);
                    else st_r[h][E] = st_u[h] + st_m[h];
                } else if constexpr (S == 6) {
                    if (ACT == VV_ACT_RELU) st_r[h][E] = fmaxf(st_t[h], 0.f);
                    else if (ACT == VV_ACT_LRELU) st_r[h][E] = st_t[h] > 0.f ? st_t[h] : 0.3f * st_t[h];
                    else st_r[h][E] = st_t[h];
                }
            }
        }
        if constexpr (G >= 28 && (C4_ABL & 64)) {
            if constexpr (G == 31) { eo[0][0] ^= __builtin_bit_cast(unsigned, st_r[0][0] + st_r[0][1] + st_r[0][2] + st_r[0][3]); eo[1][1] ^= __builtin_bit_cast(unsigned, st_r[1][0] + st_r[1][1] + st_r[1][2] + st_r[1][3] + av[0][0] + av[1][1]); }
        } else
        if constexpr (G >= 28 && U < 15) {           // the NEXT unit's accumulator values, two per gap
            constexpr int UN = U + 1, E = G - 28;
            C4_ACCRD(av[0][E], PREV[2 * (UN >> 3)][UN & 7][E]);
            C4_ACCRD(av[1][E], PREV[2 * (UN >> 3) + 1][UN & 7][E]);
        }
        if constexpr (C4_ABL & 64) {
        } else if constexpr (G == 28 || G == 29) {   // pack values (0, 1) / (2, 3) of both accumulators
            constexpr int e0 = 2 * (G - 28);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
                const bf16x2 v = {static_cast<__bf16>(st_r[h][e0]), static_cast<__bf16>(st_r[h][e0 + 1])};
                eo[h][G - 28] = __builtin_bit_cast(unsigned, v);
            }
        } else if constexpr (G == 30) {              // odd 16-lane rows of the first operand swap with the even rows of the second
            auto rx = __builtin_amdgcn_permlane16_swap(eo[0][0], eo[1][0], false, false);
            auto ry = __builtin_amdgcn_permlane16_swap(eo[0][1], eo[1][1], false, false);
            eo[0] = u32x2{rx[0], ry[0]};
            eo[1] = u32x2{rx[1], ry[1]};
        } else if constexpr (G == 31) {
            constexpr int UO = (CT >> 2) * (2 * 16 * 16 * 128) + (CT & 3) * (4 * 16 * 128) + C2 * 64;
            __builtin_amdgcn_raw_buffer_store_b128(u32x4{eo[0][0], eo[0][1], eo[1][0], eo[1][1]}, rsy, yvo, UO, 0);
            // the next unit works on the other channel-tile pair.  Riding along with the MFMAs, the next k-step boundary's lgkmcnt(0) covers
            // these reads; in the tail nothing else waits (and after its last unit nothing reads them: a dead look-ahead is not issued)
            if constexpr (CT == 7 && !(TAIL && C2 == 1)) {
                load_ss(std::integral_constant<int, 1 - C2>{}, bss, ssa);
                if constexpr (TAIL) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bss[0]), "+v"(bss[1]), "+v"(bss[2]), "+v"(bss[3]) : : "memory");
            }
        }
    };
    auto set_yvo = [&](int p) {
        const int pd = (p >> 2) & 1, ph = (p >> 1) & 1, pw = p & 1;
        yvo = (unsigned)(((((4 * wave + pd) * 16 + 2 * rh + ph) * 16 + 2 * rw + pw) * 128) + (q & 1) * 32 + (q >> 1) * 16);
    };

    // ---- one 32-deep k-step: 32 MFMAs of fragments F into ACC.  Gap 0 prepares the first read address; gaps 1..12 hold one fragment read
    // each (into G, for the next k-step) next to the address arithmetic of the read AFTER it -- never the read's own; gaps 13..31 are
    // `slot(0..18)`
    auto kstep = [&](auto &ACC, u32x4(&F)[12], u32x4(&G)[12], const unsigned(&xa)[8], auto kx_c, unsigned wa, auto first_c, auto &&slot) {
        constexpr unsigned KX = (unsigned)decltype(kx_c)::value;
        constexpr bool FIRST = decltype(first_c)::value;
        unsigned ra[2] = {0u, 0u};
        c4_static_for<32>([&](auto i_c) {
            constexpr int i = decltype(i_c)::value, cot = i >> 3, ct = i & 7;
            if constexpr (C4_ABL & 16) {
                if constexpr (FIRST) ACC[cot][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
            } else if constexpr (FIRST)
                ACC[cot][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8 *>(&F[8 + cot]), *reinterpret_cast<const bf16x8 *>(&F[ct]),
                                                                       f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
            else
                ACC[cot][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8 *>(&F[8 + cot]), *reinterpret_cast<const bf16x8 *>(&F[ct]),
                                                                       ACC[cot][ct], 0, 0, 0);
            C4_SB;
            if constexpr (C4_ABL & 8) {
            } else if constexpr (i <= 12) {
                if constexpr (i >= 1 && i <= 8) C4_RD(G[i - 1], ra[(i - 1) & 1], 0);
                else if constexpr (i >= 9) C4_RD(G[i - 1], wa, (i - 9) * 2048);
                if constexpr (i < 8) ra[i & 1] = xa[i] ^ KX;
            }
            if constexpr (i >= 13) slot(std::integral_constant<int, i - 13>{});
            C4_SB;
        });
    };

    // ---- one parity: 16 chunks (tap x K half) of two k-steps.  SET: accumulator set; EPI: the previous parity's epilogue rides along
    auto parity_body = [&](auto set_c, auto epi_c, int pi) {
        constexpr int SET = decltype(set_c)::value;
        constexpr bool EPI = decltype(epi_c)::value;
        auto &ACC = *(SET ? &accB : &accA);
        auto &PREV = *(SET ? &accA : &accB);
        const int p = p0 + pi, pd = (p >> 2) & 1, ph = (p >> 1) & 1, pw = p & 1;
        const int pn = p + 1, pnd = (pn >> 2) & 1, pnh = (pn >> 1) & 1, pnw = pn & 1;
        if (EPI) set_yvo(p - 1);
        c4_static_for<16>([&](auto j_c) {
            constexpr int J = decltype(j_c)::value, A = J >> 1, KH = J & 1;
            // k-step 0 (fragments P) | reads of k-step 1 into Q | epilogue gaps 0..18 of unit J
            C4_WAIT(P);
            C4_SB;
            kstep(ACC, P, Q, ua, std::integral_constant<int, KH ? 192 : 64>{}, ws ^ 64u, std::integral_constant<bool, J == 0>{}, [&](auto sl_c) {
                if constexpr (EPI && !(C4_ABL & 1)) epi_gap(PREV, j_c, sl_c, std::false_type{});
            });
            C4_WAIT(Q);
            // the next chunk's two pieces have landed: younger than them are this wave's two pieces of the chunk after it and, with the
            // epilogue riding along, one store per chunk (two since).  A smaller count is always safe; the first two chunks of a parity
            // follow chunks whose store count differs, so they take the strict form.  voxvae/isa_lint.py pins what a piece survives.
            if (EPI && J >= 2 && !(C4_ABL & 1)) cw_wait_vm<4>();
            else cw_wait_vm<2>();
            if constexpr (!(C4_ABL & 2)) __builtin_amdgcn_s_barrier();
            C4_SB;
            if constexpr (!(C4_ABL & 4)) issue_w(cw, stg);
            ++cw;
            stg = stg == CW_NST - 1 ? 0 : stg + 1;
            ws = wl + stg * CW_WST;
            if (KH == 1) {
#pragma unroll
                for (int ct = 0; ct < 8; ++ct) ua[ct] = ub[ct];
            }
            C4_SB;
            // k-step 1 (fragments Q) | reads of the next chunk's k-step 0 into P | epilogue gaps 19..31, then the next tap's addresses
            kstep(ACC, Q, P, ua, std::integral_constant<int, KH ? 0 : 128>{}, ws, std::false_type{}, [&](auto sl_c) {
                constexpr int sl = decltype(sl_c)::value;
                if constexpr (EPI && sl < 13 && !(C4_ABL & 1)) epi_gap(PREV, j_c, std::integral_constant<int, 19 + sl>{}, std::false_type{});
                if constexpr (KH == 0 && sl >= 13 && sl < 17) {         // next tap's row addresses, two cell tiles per gap
                    if constexpr (A < 7) {
                        tap_setup2(std::integral_constant<int, (A + 1) & 7>{}, std::integral_constant<int, 2 * (sl - 13)>{}, pd, ph, pw, ub);
                    } else {                                            // the last tap's chunk prepares the NEXT parity's first tap
                        if constexpr (sl == 13) parity_setup(pnh, pnw);
                        tap_setup2(std::integral_constant<int, 0>{}, std::integral_constant<int, 2 * (sl - 13)>{}, pnd, pnh, pnw, ub);
                    }
                }
            });
        });
    };

    parity_setup((p0 >> 1) & 1, p0 & 1);
    tap_setup(std::integral_constant<int, 0>{}, (p0 >> 2) & 1, (p0 >> 1) & 1, p0 & 1, ua);
#pragma unroll
    for (int k = 0; k < 8; ++k) C4_RD(P[k], ua[k], 0);
    C4_RD(P[8], ws, 0);
    C4_RD(P[9], ws, 2048);
    C4_RD(P[10], ws, 4096);
    C4_RD(P[11], ws, 6144);
    parity_body(std::integral_constant<int, 0>{}, std::false_type{}, 0);
#pragma unroll 1
    for (int pi = 1; pi < npar; pi += 2) {
        parity_body(std::integral_constant<int, 1>{}, std::true_type{}, pi);
        if (pi + 1 < npar) parity_body(std::integral_constant<int, 0>{}, std::true_type{}, pi + 1);
    }
    C4_WAIT(P);                                      // the look-ahead reads of the chunk after the last
    // the last parity's epilogue, on its own
    set_yvo(p0 + npar - 1);
    if ((npar - 1) & 1) {
        c4_static_for<16>([&](auto u_c) { c4_static_for<32>([&](auto k_c) { epi_gap(accB, u_c, k_c, std::true_type{}); }); });
    } else {
        c4_static_for<16>([&](auto u_c) { c4_static_for<32>([&](auto k_c) { epi_gap(accA, u_c, k_c, std::true_type{}); }); });
    }
    cw_wait_vm<0>();                                 // the tail's pieces still target this workgroup's LDS
}


}  // namespace

VV_EXPORT int vv_convT3d_k4s2_whole_supported(int side, int cin, int cout, int dtype) {
    return dtype == VV_BF16 && side == 8 && cin == 128 && cout == 64;
}

// w_skip: vv_pack_convT_k4s2_skip's image.  One workgroup per (sample, parity split).
VV_EXPORT int vv_convT3d_k4s2_whole_fwd(const void *x, const void *w_skip, const float *scale, const float *shift, void *y, int batch,
                                        int side, int cin, int cout, int act, int dtype, void *stream) {
    if (!x || !w_skip || !y) return VV_ERR_NULL;
    if (!vv_convT3d_k4s2_whole_supported(side, cin, cout, dtype) || batch <= 0) return VV_ERR_SHAPE;
    if (!vv_aligned16(x) || !vv_aligned16(w_skip) || !vv_aligned16(y)) return VV_ERR_ALIGN;
    // One workgroup per sample once the batch fills the chip (one round of 256), else the parities are split until it does.
    // History: with TWO streams two rounds (every sample loaded twice) were 2-4 % faster at batch 256 (the other stream's kernels got
    // CUs between the rounds); with the THREE streams of round 3 one round is as fast or faster (0.470-0.473 vs 0.472-0.485 ms/step,
    // profiles/microbench/ab_ps3.sh), 1.5 % faster one batch at a time, and reads every sample once: HBM traffic 1.30x -> ~1.05x the
    // algorithmic bytes.
    int ps = 1;
    while (ps < 8 && (long)batch * ps < 256) ps *= 2;
    if (const char *e = vv_hook("VV_CTW_PS")) {
        const int v = atoi(e);
        if (v == 1 || v == 2 || v == 4 || v == 8) ps = v;
    }
    // VV_CTW_SHAPE (test hook): 16 = the eight-wave kernel on v_mfma_f32_16x16x32_bf16 (default), 32 = the eight-wave kernel on
    // v_mfma_f32_32x32x16_bf16, 4 = four waves / one per SIMD / epilogue in the MFMA gaps on 16x16x32 (measured slower)
    const char *se = vv_hook("VV_CTW_SHAPE");
    const int shape = se ? atoi(se) : 16;
    const bool shape16 = shape != 32;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    auto launch = [&](auto act_c) {
        constexpr int ACT = decltype(act_c)::value;
        static const bool attr = [] {
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&ctw_kernel<ACT>), hipFuncAttributeMaxDynamicSharedMemorySize, CW_LDS);
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&ctw16_kernel<ACT>), hipFuncAttributeMaxDynamicSharedMemorySize, CW_LDS);
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&ctw4_kernel<ACT>), hipFuncAttributeMaxDynamicSharedMemorySize, CW_LDS);
            return true;
        }();
        (void)attr;
        if (shape != 16 && shape != 32)
            VV_LAUNCH(ctw4_kernel<ACT>, dim3((unsigned)batch * ps), dim3(256), CW_LDS, st, reinterpret_cast<const __bf16 *>(x),
                      reinterpret_cast<const __bf16 *>(w_skip), scale, shift, reinterpret_cast<__bf16 *>(y), 8 / ps);
        else if (shape16)
            VV_LAUNCH(ctw16_kernel<ACT>, dim3((unsigned)batch * ps), dim3(512), CW_LDS, st, reinterpret_cast<const __bf16 *>(x),
                      reinterpret_cast<const __bf16 *>(w_skip), scale, shift, reinterpret_cast<__bf16 *>(y), 8 / ps);
        else
            VV_LAUNCH(ctw_kernel<ACT>, dim3((unsigned)batch * ps), dim3(512), CW_LDS, st, reinterpret_cast<const __bf16 *>(x),
                      reinterpret_cast<const __bf16 *>(w_skip), scale, shift, reinterpret_cast<__bf16 *>(y), 8 / ps);
    };
    switch (act) {
        case VV_ACT_ELU: launch(std::integral_constant<int, VV_ACT_ELU>{}); break;
        case VV_ACT_RELU: launch(std::integral_constant<int, VV_ACT_RELU>{}); break;
        case VV_ACT_LRELU: launch(std::integral_constant<int, VV_ACT_LRELU>{}); break;
        default: launch(std::integral_constant<int, VV_ACT_NONE>{}); break;
    }
    return vv_launch_status();
}

// The training step's forward form: raw output (no folded BN, no activation: BatchNorm with batch statistics follows) + the
// per-workgroup column sums of that output for vv_bn_finalize_stats.
VV_EXPORT int vv_convT3d_k4s2_whole_stats_blocks(int batch) { return batch > 0 ? batch : 0; }

VV_EXPORT int vv_convT3d_k4s2_whole_stats_fwd(const void *x, const void *w_skip, void *y, float *stats_partial, size_t stats_bytes, int batch,
                                              int side, int cin, int cout, int dtype, void *stream) {
    if (!x || !w_skip || !y || !stats_partial) return VV_ERR_NULL;
    if (!vv_convT3d_k4s2_whole_supported(side, cin, cout, dtype) || batch <= 0) return VV_ERR_SHAPE;
    if (!vv_aligned16(x) || !vv_aligned16(w_skip) || !vv_aligned16(y) || !vv_aligned16(stats_partial)) return VV_ERR_ALIGN;
    if (stats_bytes < (size_t)batch * 2 * 64 * sizeof(float)) return VV_ERR_WORKSPACE;
    static const bool attr = [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&ctw16_kernel<VV_ACT_NONE, true>), hipFuncAttributeMaxDynamicSharedMemorySize, CW_LDS);
        return true;
    }();
    (void)attr;
    VV_LAUNCH((ctw16_kernel<VV_ACT_NONE, true>), dim3((unsigned)batch), dim3(512), CW_LDS, reinterpret_cast<hipStream_t>(stream),
              reinterpret_cast<const __bf16 *>(x), reinterpret_cast<const __bf16 *>(w_skip), nullptr, nullptr, reinterpret_cast<__bf16 *>(y), 8,
              stats_partial);
    return vv_launch_status();
}

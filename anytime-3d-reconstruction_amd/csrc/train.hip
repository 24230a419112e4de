// Training-path kernels (reference nolbo.py:1411-1447 `fit`): BatchNorm with batch statistics (forward + backward),
// activation backward, weight gradients as reduction-over-rows GEMMs on the exact-f32 MFMA, BCE / reparameterisation
// backward, Adam.  Data gradients reuse the forward implicit-GEMM kernels (the data gradient of a stride-2 conv IS the
// transposed conv with the same Keras kernel array, and vice versa).  float32 only this round.  gfx950.
#include <stdlib.h>

#include "common.h"

namespace {

// ------------------------------------------------------------------------------------------------ BatchNorm
// Column reductions over x[R][C] (C % 4 == 0, C <= 1024): workgroup = a slab of rows, lane = 4 channels.
// MODE 0: sum x, sum x^2.   MODE 1 (backward): du = dy * act'(u), u = x*scale + shift; sum du, sum du * xhat.
// 4 consecutive elements of a float32 or bf16 tensor as float4 (element index 4*i4 .. +3), and the store back.
template <typename T>
__device__ __forceinline__ f32x4 ld4(const T *p, long i4) {
    if constexpr (sizeof(T) == 4) {
        return reinterpret_cast<const f32x4 *>(p)[i4];
    } else {
        const bf16x4 v = reinterpret_cast<const bf16x4 *>(p)[i4];
        return f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
    }
}
template <typename T>
__device__ __forceinline__ void st4(T *p, long i4, f32x4 v) {
    if constexpr (sizeof(T) == 4) {
        reinterpret_cast<f32x4 *>(p)[i4] = v;
    } else {
        bf16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = static_cast<__bf16>(v[e]);
        reinterpret_cast<bf16x4 *>(p)[i4] = o;
    }
}
// runtime-typed variant for the weight-gradient staging (dtype flag in the argument block)
__device__ __forceinline__ f32x4 ld4_rt(const void *p, long elem, int is_bf16) {
    return is_bf16 ? ld4(reinterpret_cast<const __bf16 *>(p) + elem, 0) : ld4(reinterpret_cast<const float *>(p) + elem, 0);
}

// Vector width of the BatchNorm sweeps: 16 bytes per lane and load (4 floats / 8 bf16).
template <typename T> struct BnVec { static constexpr int V = sizeof(T) == 2 ? 8 : 4; };
template <typename T>
__device__ __forceinline__ void ldv(const T *p, long iv, float (&o)[BnVec<T>::V]) {
    if constexpr (sizeof(T) == 4) {
        const f32x4 v = reinterpret_cast<const f32x4 *>(p)[iv];
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = v[e];
    } else {
        const bf16x8 v = reinterpret_cast<const bf16x8 *>(p)[iv];
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (float)v[e];
    }
}
template <typename T>
__device__ __forceinline__ void stv(T *p, long iv, const float (&o)[BnVec<T>::V]) {
    if constexpr (sizeof(T) == 4) {
        reinterpret_cast<f32x4 *>(p)[iv] = f32x4{o[0], o[1], o[2], o[3]};
    } else {
        bf16x8 v;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = static_cast<__bf16>(o[e]);
        reinterpret_cast<bf16x8 *>(p)[iv] = v;
    }
}
template <int V>
__device__ __forceinline__ void ldp(const float *p, int c, float (&o)[V]) {      // V consecutive per-channel parameters
#pragma unroll
    for (int q = 0; q < V; q += 4) {
        const f32x4 v = *reinterpret_cast<const f32x4 *>(p + c + q);
#pragma unroll
        for (int e = 0; e < 4; ++e) o[q + e] = v[e];
    }
}
template <typename T>
__device__ __forceinline__ float bn_act_grad(float u, int act) {                  // act'(u)
    // ELU'(u) = 1 for u > 0, exp(u) otherwise = exp(min(u, 0)) for every u (exp(0) is exactly 1): no compare + select per element
    if (act == VV_ACT_ELU) return sizeof(T) == 4 ? expf(fminf(u, 0.f)) : __expf(fminf(u, 0.f));
    if (act == VV_ACT_RELU) return u > 0.f ? 1.f : 0.f;
    if (act == VV_ACT_LRELU) return u > 0.f ? 1.f : 0.3f;
    return 1.f;
}

// Per-block partial sums over a row range.  MODE 0: (sum x, sum x^2).  MODE 1: (sum du, sum du * xhat), du = dy * act'.
// A thread owns V consecutive channels; 256 / (C / V) rows run in parallel, four rows per thread in flight.
template <int MODE, typename T, int ACT = -1>     // ACT >= 0: the activation as a compile-time constant (the backward reduction is VALU-bound:
                                                  // 3.9 TB/s with the run-time switch per element against 5-6 TB/s for its sibling sweeps)
__global__ __launch_bounds__(256) void bn_reduce_kernel(const T *__restrict__ x, const T *__restrict__ dy,
                                                        const float *__restrict__ scale, const float *__restrict__ shift,
                                                        const float *__restrict__ mean, const float *__restrict__ rstd,
                                                        float *__restrict__ partial, long R, int C, int rows_per_block, int act) {
    constexpr int V = BnVec<T>::V;
    __shared__ float red[2][256 * V];
    const int cvn = C / V;                   // vector columns per row
    const int tid = threadIdx.x;
    const int rows_par = 256 / cvn > 0 ? 256 / cvn : 1;   // rows handled in parallel (C <= 256 V)
    const int cv = tid % cvn, rsub = tid / cvn;
    const long r0 = (long)blockIdx.x * rows_per_block;
    const long r1 = r0 + rows_per_block < R ? r0 + rows_per_block : R;
    float s0[V], s1[V], sc[V], sh[V], mu[V], rs[V];
#pragma unroll
    for (int e = 0; e < V; ++e) { s0[e] = 0.f; s1[e] = 0.f; sc[e] = 1.f; sh[e] = 0.f; mu[e] = 0.f; rs[e] = 1.f; }
    if (MODE == 1) {
        ldp<V>(scale, cv * V, sc);
        ldp<V>(shift, cv * V, sh);
        ldp<V>(mean, cv * V, mu);
        ldp<V>(rstd, cv * V, rs);
    }
    auto fold = [&](const float (&v)[V], const float (&g)[V]) {
#pragma unroll
        for (int e = 0; e < V; ++e) {
            if (MODE == 0) {
                s0[e] += v[e];
                s1[e] += v[e] * v[e];
            } else {
                const float d = g[e] * bn_act_grad<T>(v[e] * sc[e] + sh[e], ACT >= 0 ? ACT : act);
                s0[e] += d;
                s1[e] += d * ((v[e] - mu[e]) * rs[e]);
            }
        }
    };
    if (rsub < rows_par) {
        long r = r0 + rsub;
        // four rows per trip: eight independent loads in flight per thread (the sweep is HBM-bound)
        for (; r + 3L * rows_par < r1; r += 4L * rows_par) {
            float v[4][V], g[4][V];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                ldv<T>(x + (r + (long)k * rows_par) * C, cv, v[k]);
                if (MODE == 1) ldv<T>(dy + (r + (long)k * rows_par) * C, cv, g[k]);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) fold(v[k], g[k]);
        }
        for (; r < r1; r += rows_par) {
            float v[V], g[V];
            ldv<T>(x + r * C, cv, v);
            if (MODE == 1) ldv<T>(dy + r * C, cv, g);
            fold(v, g);
        }
    }
#pragma unroll
    for (int e = 0; e < V; ++e) { red[0][tid * V + e] = s0[e]; red[1][tid * V + e] = s1[e]; }
    __syncthreads();
    if (tid < cvn) {
        float a[V], b[V];
#pragma unroll
        for (int e = 0; e < V; ++e) { a[e] = 0.f; b[e] = 0.f; }
        for (int k = 0; k < rows_par && k * cvn + tid < 256; ++k)
#pragma unroll
            for (int e = 0; e < V; ++e) { a[e] += red[0][(k * cvn + tid) * V + e]; b[e] += red[1][(k * cvn + tid) * V + e]; }
#pragma unroll
        for (int e = 0; e < V; ++e) {
            partial[((size_t)blockIdx.x * 2 + 0) * C + tid * V + e] = a[e];
            partial[((size_t)blockIdx.x * 2 + 1) * C + tid * V + e] = b[e];
        }
    }
}

// Sum of the per-block partials of BN_FC channels in double, by 256 threads = 4 channels x 64 slices of the block list
// (a 1024-block list costs 16 trips of 4 independent loads, not 64 dependent ones); the 64 slice sums of a channel are
// added in slice order (deterministic).  Result valid in threads 0..3 (channel = c0 + tid).
constexpr int BN_FC = 4;
__device__ __forceinline__ void bn_partial_sums(const float *__restrict__ partial, int nblk, int C, int c0, double &s, double &ss) {
    __shared__ double red[2][64][BN_FC + 1];
    const int tid = threadIdx.x, cl = tid & (BN_FC - 1), sl = tid >> 2, c = c0 + cl;
    double a = 0.0, b = 0.0;
    if (c < C) {
        int k = sl;
        for (; k + 192 < nblk; k += 256) {
            float va[4], vb[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                va[u] = partial[((size_t)(k + 64 * u) * 2) * C + c];
                vb[u] = partial[((size_t)(k + 64 * u) * 2 + 1) * C + c];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) { a += va[u]; b += vb[u]; }
        }
        for (; k < nblk; k += 64) { a += partial[((size_t)k * 2) * C + c]; b += partial[((size_t)k * 2 + 1) * C + c]; }
    }
    red[0][sl][cl] = a; red[1][sl][cl] = b;
    __syncthreads();
    s = 0.0; ss = 0.0;
    if (tid < BN_FC)
        for (int k = 0; k < 64; ++k) { s += red[0][k][tid]; ss += red[1][k][tid]; }
}

// Forward finalise: batch mean / biased variance -> scale, shift, rstd; moving statistics update (momentum).
__global__ __launch_bounds__(256) void bn_stats_finalize_kernel(const float *__restrict__ partial, int nblk, long R, int C, const float *gamma,
                                         const float *beta, float eps, float momentum, float *mean, float *var, float *rstd,
                                         float *scale, float *shift, float *moving_mean, float *moving_var) {
    double s, ss;
    bn_partial_sums(partial, nblk, C, blockIdx.x * BN_FC, s, ss);
    const int c = blockIdx.x * BN_FC + threadIdx.x;
    if (threadIdx.x >= BN_FC || c >= C) return;
    const double m = s / (double)R;
    double v = ss / (double)R - m * m;
    if (v < 0.0) v = 0.0;
    const float r = (float)(1.0 / sqrt(v + (double)eps));
    mean[c] = (float)m; var[c] = (float)v; rstd[c] = r;
    scale[c] = gamma[c] * r;
    shift[c] = beta[c] - (float)m * gamma[c] * r;
    if (moving_mean) moving_mean[c] = moving_mean[c] * momentum + (float)m * (1.f - momentum);
    if (moving_var) moving_var[c] = moving_var[c] * momentum + (float)v * (1.f - momentum);
}

// Backward finalise: dbeta = sum du, dgamma = sum du*xhat
__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const float *__restrict__ partial, int nblk, int C, float *dgamma, float *dbeta) {
    double s, ss;
    bn_partial_sums(partial, nblk, C, blockIdx.x * BN_FC, s, ss);
    const int c = blockIdx.x * BN_FC + threadIdx.x;
    if (threadIdx.x >= BN_FC || c >= C) return;
    dbeta[c] = (float)s;
    dgamma[c] = (float)ss;
}

// y = act(x*scale + shift).  A thread owns V consecutive channels for the whole sweep and walks rows: the per-channel
// vectors are loaded once per thread (the element-wise form re-read them for every vector: 12 parameter loads next to the
// 2 payload loads of the backward kernel, which held it at 4.2 TB/s), four rows in flight per trip.
template <typename T>
__global__ __launch_bounds__(256) void bn_act_fwd_kernel(const T *__restrict__ x, const float *__restrict__ scale,
                                                         const float *__restrict__ shift, T *__restrict__ y, long R, int C, int act) {
    constexpr int V = BnVec<T>::V;
    const int cvn = C / V, rows_par = 256 / cvn > 0 ? 256 / cvn : 1;
    const int cv = threadIdx.x % cvn, rsub = threadIdx.x / cvn;
    if (rsub >= rows_par) return;
    float sc[V], sh[V];
    ldp<V>(scale, cv * V, sc);
    ldp<V>(shift, cv * V, sh);
    // a block sweeps one contiguous row range (DRAM pages stay local), rows_par rows per step
    const long rpb = ((R + gridDim.x - 1) / gridDim.x + rows_par - 1) / rows_par * rows_par;
    const long rend = (blockIdx.x + 1) * rpb < R ? (blockIdx.x + 1) * rpb : R;
    R = rend;
    const long stride = rows_par;
    long r = (long)blockIdx.x * rpb + rsub;
    for (; r + 3 * stride < R; r += 4 * stride) {
        float v[4][V];
#pragma unroll
        for (int k = 0; k < 4; ++k) ldv<T>(x + (r + k * stride) * C, cv, v[k]);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float o[V];
#pragma unroll
            for (int e = 0; e < V; ++e) o[e] = vv_apply_act(v[k][e] * sc[e] + sh[e], act);
            stv<T>(y + (r + k * stride) * C, cv, o);
        }
    }
    for (; r < R; r += stride) {
        float v[V], o[V];
        ldv<T>(x + r * C, cv, v);
#pragma unroll
        for (int e = 0; e < V; ++e) o[e] = vv_apply_act(v[e] * sc[e] + sh[e], act);
        stv<T>(y + r * C, cv, o);
    }
}

// dx = gamma*rstd * (du - dbeta/R - xhat*dgamma/R),  du = dy*act'(x*scale+shift); same thread layout, with the per-channel
// constants folded to five: dx = sc * (du - k1 - (x - mu) * k2), k1 = dbeta/R, k2 = dgamma*rstd/R.
template <typename T>
__global__ __launch_bounds__(256) void bn_act_bwd_kernel(const T *__restrict__ x, const T *__restrict__ dy, const float *__restrict__ scale,
                                                         const float *__restrict__ shift, const float *__restrict__ mean,
                                                         const float *__restrict__ rstd, const float *__restrict__ dgamma,
                                                         const float *__restrict__ dbeta, T *__restrict__ dx, long R, int C, float invR, int act) {
    constexpr int V = BnVec<T>::V;
    const int cvn = C / V, rows_par = 256 / cvn > 0 ? 256 / cvn : 1;
    const int cv = threadIdx.x % cvn, rsub = threadIdx.x / cvn;
    if (rsub >= rows_par) return;
    float sc[V], sh[V], mu[V], k1[V], k2[V];
    {
        float rs[V], dg[V], db[V];
        ldp<V>(scale, cv * V, sc);
        ldp<V>(shift, cv * V, sh);
        ldp<V>(mean, cv * V, mu);
        ldp<V>(rstd, cv * V, rs);
        ldp<V>(dgamma, cv * V, dg);
        ldp<V>(dbeta, cv * V, db);
#pragma unroll
        for (int e = 0; e < V; ++e) { k1[e] = db[e] * invR; k2[e] = dg[e] * invR * rs[e]; }
    }
    auto one = [&](const float (&v)[V], const float (&g)[V], float (&o)[V]) {
#pragma unroll
        for (int e = 0; e < V; ++e) {
            const float d = g[e] * bn_act_grad<T>(v[e] * sc[e] + sh[e], act);
            o[e] = sc[e] * (d - k1[e] - (v[e] - mu[e]) * k2[e]);
        }
    };
    const long rpb = ((R + gridDim.x - 1) / gridDim.x + rows_par - 1) / rows_par * rows_par;   // contiguous row range per block
    const long rend = (blockIdx.x + 1) * rpb < R ? (blockIdx.x + 1) * rpb : R;
    R = rend;
    const long stride = rows_par;
    long r = (long)blockIdx.x * rpb + rsub;
    for (; r + stride < R; r += 2 * stride) {
        float v[2][V], g[2][V], o[V];
#pragma unroll
        for (int k = 0; k < 2; ++k) { ldv<T>(x + (r + k * stride) * C, cv, v[k]); ldv<T>(dy + (r + k * stride) * C, cv, g[k]); }
#pragma unroll
        for (int k = 0; k < 2; ++k) { one(v[k], g[k], o); stv<T>(dx + (r + k * stride) * C, cv, o); }
    }
    for (; r < R; r += stride) {
        float v[V], g[V], o[V];
        ldv<T>(x + r * C, cv, v);
        ldv<T>(dy + r * C, cv, g);
        one(v, g, o);
        stv<T>(dx + r * C, cv, o);
    }
}

// ------------------------------------------------------------------------------------------------ weight gradients
// dW[m][n] = sum_r A[r][m] * G[r][n]   (reduction over rows r = samples x positions), exact-f32 MFMA 32x32x2.
//   AMODE 0: A[r][m] dense, row pitch lda                                  (Dense kernels, pooled / panel layers)
//   AMODE 1: A[r][(t,ci)] = src[b, 2o-1+t, ci] (zero in the SAME padding), r = (b,o) over the HALF-size grid: the
//            weight gradient of a stride-2 Conv3D (src = layer input, G = dL/d(conv out)) -> Keras [t][ci][co];
//            with src = dL/d(out) and G = layer input it is the weight gradient of a stride-2 Conv3DTranspose
//            -> Keras [t][co][ci].
//   AMODE 2: as 1 with one source channel (m = tap): first conv / last transposed conv.
// Tile 64 (m) x BN (n), 4 waves (2x2), reduction chunks of 32 rows staged in LDS as [r][m] / [r][n] (the f32 MFMA
// wants lane = m, k = row: rows of the chunk ARE k, no transposition needed).  grid.y splits the reduction; slabs are
// summed in split order by wgrad_reduce_kernel (deterministic).
struct WgradArgs {
    const void *A;      // float32 or bf16 (a_bf16)
    const void *G;      // float32 or bf16 (g_bf16)
    float *slabs;       // [splits][M][N]
    long R;             // reduction rows
    int M, N;
    int lda;            // AMODE 0 row pitch (elements)
    int din_log2, cin;  // AMODE 1/2: source grid side (log2) and channels
    int rows_per_split;
    int a_bf16, g_bf16;
};

template <int AMODE, int BN>
__global__ __launch_bounds__(256) void wgrad_kernel(const WgradArgs a) {
    constexpr int BM = 64, BR = 32, TN = BN / 64;
    __shared__ __attribute__((aligned(16))) float As[BR][BM + 4];
    __shared__ __attribute__((aligned(16))) float Gs[BR][BN + 4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int ntn = (a.N + BN - 1) / BN;
    const int tile_n = blockIdx.x % ntn, tile_m = blockIdx.x / ntn;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const long r_begin = (long)blockIdx.y * a.rows_per_split;
    const long r_end = r_begin + a.rows_per_split < a.R ? r_begin + a.rows_per_split : a.R;
    const int li = a.din_log2, n = 1 << li, lo = li - 1, omsk = (1 << lo) - 1;

    f32x16 acc[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[j][q] = 0.f;

    // staging roles: A chunk = 32 rows x 64 floats = 512 float4 slots -> 2 per thread; G chunk = 32 x BN -> BN/32 per thread
    const int fr = lane & 31, fh = lane >> 5;
    for (long rc = r_begin; rc < r_end; rc += BR) {
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int slot = tid + 256 * it, rr = slot >> 4, c4 = slot & 15;
            const long r = rc + rr;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (r < r_end) {
                if (AMODE == 0) {
                    const int m = m0 + c4 * 4;
                    if (m < a.M) v = ld4_rt(a.A, r * a.lda + m, a.a_bf16);
                } else {
                    const int ow = (int)(r & omsk), oh = (int)((r >> lo) & omsk), od = (int)((r >> (2 * lo)) & omsk);
                    const long b = r >> (3 * lo);
                    if (AMODE == 1) {
                        const int m = m0 + c4 * 4, t = m / a.cin, ci = m % a.cin;   // cin % 64 == 0: a tile stays inside one tap
                        const int id = 2 * od - 1 + (t >> 4), ih = 2 * oh - 1 + ((t >> 2) & 3), iw = 2 * ow - 1 + (t & 3);
                        if (m < a.M && (unsigned)id < (unsigned)n && (unsigned)ih < (unsigned)n && (unsigned)iw < (unsigned)n)
                            v = ld4_rt(a.A, ((((b << li) + id << li) + ih << li) + iw) * a.cin + ci, a.a_bf16);
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int t = c4 * 4 + e;
                            const int id = 2 * od - 1 + (t >> 4), ih = 2 * oh - 1 + ((t >> 2) & 3), iw = 2 * ow - 1 + (t & 3);
                            if ((unsigned)id < (unsigned)n && (unsigned)ih < (unsigned)n && (unsigned)iw < (unsigned)n)
                                v[e] = reinterpret_cast<const float *>(a.A)[(((b << li) + id << li) + ih << li) + iw];   // cin = 1 sources are float32 grids
                        }
                    }
                }
            }
            *reinterpret_cast<f32x4 *>(&As[rr][c4 * 4]) = v;
        }
#pragma unroll
        for (int it = 0; it < BN / 32; ++it) {
            const int slot = tid + 256 * it, rr = slot / (BN / 4), c4 = slot % (BN / 4);
            const long r = rc + rr;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            const int nn = n0 + c4 * 4;
            if (r < r_end && nn < a.N) v = ld4_rt(a.G, r * a.N + nn, a.g_bf16);
            *reinterpret_cast<f32x4 *>(&Gs[rr][c4 * 4]) = v;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < BR; k += 2) {
            const float av = As[k + fh][wm * 32 + fr];
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const float gv = Gs[k + fh][wn * (BN / 2) + j * 32 + fr];
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, gv, acc[j], 0, 0, 0);   // D[m][n]: lane = n, regs walk m
            }
        }
        __syncthreads();
    }
    float *slab = a.slabs + (size_t)blockIdx.y * a.M * a.N;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int nn = n0 + wn * (BN / 2) + j * 32 + fr;
        if (nn >= a.N) continue;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int m = m0 + wm * 32 + (q & 3) + 8 * (q >> 2) + 4 * fh;
            if (m < a.M) slab[(size_t)m * a.N + nn] = acc[j][q];
        }
    }
}

// ---- bf16 weight gradients on v_mfma_f32_32x32x16_bf16.  Both operands of dW = A^T G are k-strided (k = the row index
// r): the LDS tiles stay row-major as staged, [column block of 32][row][32 bf16] (64-byte rows, so the 32 lanes of a
// half cover 256 B exactly once), and the MFMA fragments are read with the hardware transpose ds_read_b64_tr_b16:
// lane 4q+p of a 16-lane group supplies row q / columns 4p..4p+3 of a 4 x 16 block and receives column (lane & 15) of
// the 4 rows -- two reads give the 8 consecutive k of a lane.  128 x 128 tile per 256-thread workgroup (2 x 2 waves of
// 64 x 64), 64-row chunks staged by LDS-DMA into a 2-deep ring (zeros for padded taps / rows past R through the
// buffer range check), split over the reduction into slabs summed in fixed order by wgrad_reduce_kernel.
struct WgradBArgs {
    const void *A, *G;
    float *slabs;
    long R;
    int M, N, lda, din_log2, cin, rows_per_split;
    unsigned a_bytes, g_bytes;
};

template <int AMODE, bool ONE = false>      // 0: A[r][m] dense (pitch lda); 1: A[r][(t,ci)] = src[b, 2o-1+t, ci], cin % 64 == 0;
                          // 2: A[r][t] = src[b, 2o-1+t] of a single-channel float32 grid (M = 64), built in registers
__global__ __launch_bounds__(256, ONE ? 4 : 2) void wgrad_bf16_kernel(const WgradBArgs a) {
    constexpr int BM = 128, BN = 128, BR = 64, OPB = BR * 128 * 2;        // one operand chunk: 16 KiB
    // ONE (AMODE 2 with N <= 64, chosen at launch): a single 64 x 64 output tile.  Only column blocks 0, 1 of either operand exist, so a
    // stage is 16 KiB (A | G, 8 KiB each) instead of 32, the whole ring 32 KiB, and FOUR workgroups fit a CU: the kernel waits for one
    // chunk at a time (vmcnt(0) + barrier per 64 rows), so what it achieves is set by the chunks in flight per CU.
    constexpr int STG = ONE ? 16384 : 2 * OPB, GOFF = ONE ? 8192 : OPB;
    extern __shared__ __attribute__((aligned(16))) char smem[];           // [2 stages][A | G]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int ntn = (a.N + BN - 1) / BN;
    const int tile_n = blockIdx.x % ntn, tile_m = blockIdx.x / ntn;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const long r_begin = (long)blockIdx.y * a.rows_per_split;
    const long r_end = r_begin + a.rows_per_split < a.R ? r_begin + a.rows_per_split : a.R;
    const int li = a.din_log2, n = 1 << li, lo = li - 1, omsk = (1 << lo) - 1;
    constexpr bool one_tile = ONE;                          // single-channel layer, one 64 x 64 output tile: K split over the waves (below)
    const u32x4 rsa = vv_make_rsrc(a.A, a.a_bytes), rsg = vv_make_rsrc(a.G, a.g_bytes);
    const unsigned lds0 = (unsigned)(unsigned long long)(lptr_t)smem;

    // staging: piece = (column block cb, 16-row group rg): 16 rows x 64 B; 16 pieces per operand, 4 per wave each
    const int prow = lane >> 2, pch = lane & 3;
    auto stage = [&](long rc, int st) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int piece = wave * 4 + i, cb = piece >> 2, rg = piece & 3;
            if (one_tile && cb >= 2) continue;             // single-channel layers with <= 64 channels: column blocks 2, 3 are never read
            const long r = rc + rg * 16 + prow;
            const int mcol = m0 + cb * 32, ncol = n0 + cb * 32;
            unsigned va = 0xFFFFFFF0u, vg = 0xFFFFFFF0u;
            if (r < r_end) {
                if (mcol < a.M) {
                    if (AMODE == 0) {
                        va = (unsigned)((r * a.lda + mcol + pch * 8) * 2);
                    } else {
                        const int ow = (int)(r & omsk), oh = (int)((r >> lo) & omsk), od = (int)((r >> (2 * lo)) & omsk);
                        const long b = r >> (3 * lo);
                        const int t = mcol / a.cin, ci = mcol - t * a.cin;
                        const int id = 2 * od - 1 + (t >> 4), ih = 2 * oh - 1 + ((t >> 2) & 3), iw = 2 * ow - 1 + (t & 3);
                        if ((unsigned)id < (unsigned)n && (unsigned)ih < (unsigned)n && (unsigned)iw < (unsigned)n)
                            va = (unsigned)((((((b << li) + id << li) + ih << li) + iw) * a.cin + ci + pch * 8) * 2);
                    }
                }
                if (ncol < a.N) vg = (unsigned)((r * a.N + ncol + pch * 8) * 2);
            }
            const unsigned dst = lds0 + st * STG + cb * 4096 + rg * 1024;
            if (AMODE != 2) vv_dma16(rsa, va, dst);
            vv_dma16(rsg, vg, dst + GOFF);
        }
    };
    // AMODE 2: thread = (chunk row tid >> 2, tap plane td = tid & 3) owns 16 taps (th, tw) = 32 bytes of the row.  The
    // occupancy is loaded one chunk ahead (four 16-byte loads of the 4-voxel run 2 ow - 1 .. 2 ow + 2, read at a shifted
    // base on the two grid edges so that no load leaves the row), converted and written after the MFMAs of the chunk.
    const float *srcf = reinterpret_cast<const float *>(a.A);
    f32x4 xr[4];
    int xsh = 0;
    auto load_x = [&](long rc) {
        const long r = rc + (tid >> 2);
        const int td = tid & 3;
        const int ow = (int)(r & omsk), oh = (int)((r >> lo) & omsk), od = (int)((r >> (2 * lo)) & omsk);
        const long b = r >> (3 * lo);
        const int id = 2 * od - 1 + td, iw0 = 2 * ow - 1;
        xsh = iw0 < 0 ? 1 : (iw0 + 3 >= n ? -1 : 0);
#pragma unroll
        for (int th = 0; th < 4; ++th) {
            const int ih = 2 * oh - 1 + th;
            const bool ok = r < r_end && (unsigned)id < (unsigned)n && (unsigned)ih < (unsigned)n;
            xr[th] = ok ? *reinterpret_cast<const f32x4 *>(srcf + ((((b << li) + id << li) + ih) << li) + iw0 + xsh) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    auto store_x = [&](int st) {
        const int row = tid >> 2, td = tid & 3;
        bf16x8 o[2];
#pragma unroll
        for (int th = 0; th < 4; ++th) {
            const f32x4 l = xr[th];
            const f32x4 v = xsh == 0 ? l : (xsh > 0 ? f32x4{0.f, l[0], l[1], l[2]} : f32x4{l[1], l[2], l[3], 0.f});
#pragma unroll
            for (int e = 0; e < 4; ++e) o[th >> 1][(th & 1) * 4 + e] = static_cast<__bf16>(v[e]);
        }
        char *dst = smem + st * STG + (td >> 1) * 4096 + row * 64 + (td & 1) * 32;
        *reinterpret_cast<bf16x8 *>(dst) = o[0];
        *reinterpret_cast<bf16x8 *>(dst + 16) = o[1];
    };
    if (AMODE == 2 && !ONE) {                              // taps 64..127 of the 128-wide tile do not exist: column blocks 2, 3 stay zero
        const bf16x8 z = {};
#pragma unroll
        for (int q = 0; q < 4; ++q) *reinterpret_cast<bf16x8 *>(smem + (tid >> 7) * (2 * OPB) + 8192 + ((tid & 127) * 4 + q) * 16) = z;
    }

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;

    // transposed-read address of this lane inside a column block, for k-step 0: row 8 fh + q, columns 16 (g & 1) + 4p
    const int g4 = lane >> 4, q4 = (lane & 15) >> 2, p4 = lane & 3;
    const unsigned troff = ((g4 >> 1) * 8 + q4) * 64 + ((g4 & 1) * 16 + p4 * 4) * 2;
    long rc = r_begin;
    int st = 0;
    if (rc < r_end) {
        if (AMODE == 2) { load_x(rc); store_x(0); }
        stage(rc, 0);
    }
    for (; rc < r_end; rc += BR, st ^= 1) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                     // chunk landed for everyone; previous chunk's reads are done
        const bool more = rc + BR < r_end;
        if (more) {
            if (AMODE == 2) load_x(rc + BR);
            stage(rc + BR, st ^ 1);
        }
        const unsigned sa = lds0 + st * STG, sg = sa + GOFF;
        if (one_tile) {
            // M = 64, N <= 64: ONE 64 x 64 tile.  The generic mapping (2 x 2 waves of 64 x 64 on a 128 x 128 tile) left three of the
            // four waves multiplying zeros and the fourth one waiting for sixteen serialised fragment reads per chunk; here the
            // four waves split the chunk's K (one 16-row k-step each), issue their eight transposed reads together, and their
            // accumulators meet through LDS after the last chunk (in wave order: deterministic).
            const unsigned ka = sa + wave * 1024 + troff, kg = sg + wave * 1024 + troff;
            u32x2 r0, r1, r2, r3, r4, r5, r6, r7;
            asm volatile("ds_read_b64_tr_b16 %0, %8\n\tds_read_b64_tr_b16 %1, %8 offset:256\n\t"
                         "ds_read_b64_tr_b16 %2, %8 offset:4096\n\tds_read_b64_tr_b16 %3, %8 offset:4352\n\t"
                         "ds_read_b64_tr_b16 %4, %9\n\tds_read_b64_tr_b16 %5, %9 offset:256\n\t"
                         "ds_read_b64_tr_b16 %6, %9 offset:4096\n\tds_read_b64_tr_b16 %7, %9 offset:4352\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(r4), "=&v"(r5), "=&v"(r6), "=&v"(r7) : "v"(ka), "v"(kg) : "memory");
            const u32x4 a0 = {r0[0], r0[1], r1[0], r1[1]}, a1 = {r2[0], r2[1], r3[0], r3[1]};
            const u32x4 g0 = {r4[0], r4[1], r5[0], r5[1]}, g1 = {r6[0], r6[1], r7[0], r7[1]};
            const bf16x8 fa[2] = {*reinterpret_cast<const bf16x8 *>(&a0), *reinterpret_cast<const bf16x8 *>(&a1)};
            const bf16x8 fg[2] = {*reinterpret_cast<const bf16x8 *>(&g0), *reinterpret_cast<const bf16x8 *>(&g1)};
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fg[j], acc[i][j], 0, 0, 0);   // D[m][n]
        } else
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            // the eight transposed reads of a k-step are issued together and waited for once (one wait per fragment serialised
            // eight LDS round trips per k-step)
            const unsigned ka = sa + wm * 8192 + ks * 1024 + troff, kg = sg + wn * 8192 + ks * 1024 + troff;
            u32x2 r0, r1, r2, r3, r4, r5, r6, r7;
            asm volatile("ds_read_b64_tr_b16 %0, %8\n\tds_read_b64_tr_b16 %1, %8 offset:256\n\t"
                         "ds_read_b64_tr_b16 %2, %8 offset:4096\n\tds_read_b64_tr_b16 %3, %8 offset:4352\n\t"
                         "ds_read_b64_tr_b16 %4, %9\n\tds_read_b64_tr_b16 %5, %9 offset:256\n\t"
                         "ds_read_b64_tr_b16 %6, %9 offset:4096\n\tds_read_b64_tr_b16 %7, %9 offset:4352\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(r4), "=&v"(r5), "=&v"(r6), "=&v"(r7) : "v"(ka), "v"(kg) : "memory");
            const u32x4 a0 = {r0[0], r0[1], r1[0], r1[1]}, a1 = {r2[0], r2[1], r3[0], r3[1]};
            const u32x4 g0 = {r4[0], r4[1], r5[0], r5[1]}, g1 = {r6[0], r6[1], r7[0], r7[1]};
            const bf16x8 fa[2] = {*reinterpret_cast<const bf16x8 *>(&a0), *reinterpret_cast<const bf16x8 *>(&a1)};
            const bf16x8 fg[2] = {*reinterpret_cast<const bf16x8 *>(&g0), *reinterpret_cast<const bf16x8 *>(&g1)};
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fg[j], acc[i][j], 0, 0, 0);   // D[m][n]
        }
        if (AMODE == 2 && more) store_x(st ^ 1);
    }
    const int fr = lane & 31, fh = lane >> 5;
    float *slab = a.slabs + (size_t)blockIdx.y * a.M * a.N;
    if (one_tile) {
        // the four waves' accumulators meet in the 32 KiB of the ring: waves 0, 1 store, waves 2, 3 add on top (same lane -> same
        // element), then every thread adds the two halves: (w0 + w2) + (w1 + w3), a fixed order
        float *red = reinterpret_cast<float *>(smem) + (wave & 1) * 4096;
        __syncthreads();                                     // every wave is done with the stages
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            if ((wave >> 1) == pass) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int q = 0; q < 16; ++q) {
                            float *e = red + (i * 32 + (q & 3) + 8 * (q >> 2) + 4 * fh) * 64 + j * 32 + fr;
                            *e = pass == 0 ? acc[i][j][q] : *e + acc[i][j][q];
                        }
            }
            __syncthreads();
        }
        const float *r0 = reinterpret_cast<const float *>(smem);
        for (int idx = tid; idx < 4096; idx += 256) {
            const int m = idx >> 6, nn = idx & 63;
            if (nn < a.N) slab[(size_t)m * a.N + nn] = r0[idx] + r0[4096 + idx];
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int nn = n0 + (wn * 2 + j) * 32 + fr;
            if (nn >= a.N) continue;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int m = m0 + (wm * 2 + i) * 32 + (q & 3) + 8 * (q >> 2) + 4 * fh;
                if (m < a.M) slab[(size_t)m * a.N + nn] = acc[i][j][q];
            }
        }
}

__global__ void wgrad_reduce_kernel(const float *__restrict__ slabs, float *__restrict__ out, long n, int splits, float alpha,
                                    int accumulate) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        float s = 0.f;
        int k = 0;
        for (; k + 4 <= splits; k += 4) {       // four loads in flight, added in split order
            const float v0 = slabs[(size_t)k * n + i], v1 = slabs[(size_t)(k + 1) * n + i], v2 = slabs[(size_t)(k + 2) * n + i],
                        v3 = slabs[(size_t)(k + 3) * n + i];
            s += v0; s += v1; s += v2; s += v3;
        }
        for (; k < splits; ++k) s += slabs[(size_t)k * n + i];
        out[i] = accumulate ? out[i] + alpha * s : alpha * s;
    }
}

// Few outputs, many slabs (the single-channel layers: 4096 outputs x 256 splits): 64 outputs x 4 slices of the split
// list per workgroup; the 4 slice sums are added in slice order.
__global__ __launch_bounds__(256) void wgrad_reduce_sliced_kernel(const float *__restrict__ slabs, float *__restrict__ out, long n,
                                                                  int splits, float alpha, int accumulate) {
    __shared__ float red[4][64];
    const int ol = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const long i = (long)blockIdx.x * 64 + ol;
    const int per = (splits + 3) / 4, k0 = sl * per, k1 = k0 + per < splits ? k0 + per : splits;
    float s = 0.f;
    if (i < n) {
        int k = k0;
        for (; k + 8 <= k1; k += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = slabs[(size_t)(k + u) * n + i];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; k < k1; ++k) s += slabs[(size_t)k * n + i];
    }
    red[sl][ol] = s;
    __syncthreads();
    if (sl == 0 && i < n) {
        const float t = ((red[0][ol] + red[1][ol]) + red[2][ol]) + red[3][ol];
        out[i] = accumulate ? out[i] + alpha * t : alpha * t;
    }
}

// Adjoints of the two "layer as a dense panel" packings (vv_pack_conv_k4s1_meanpool / vv_pack_convT_k4s1_dense).
__global__ void unpack_meanpool_grad_kernel(const float *__restrict__ dpanel, float *__restrict__ dw, int side, int cin, int cout) {
    // dw[t][ci][co] = (1/S^3) sum_{(i,o) : i - o + 1 = t per axis} dpanel[co][i*cin + ci]
    const int S3 = side * side * side;
    const long total = (long)64 * cin * cout;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int co = (int)(idx % cout), ci = (int)((idx / cout) % cin), t = (int)(idx / ((long)cout * cin));
        const int td = t >> 4, th = (t >> 2) & 3, tw = t & 3;
        float s = 0.f;
        for (int od = 0; od < side; ++od) {
            const int id = od + td - 1;
            if ((unsigned)id >= (unsigned)side) continue;
            for (int oh = 0; oh < side; ++oh) {
                const int ih = oh + th - 1;
                if ((unsigned)ih >= (unsigned)side) continue;
                for (int ow = 0; ow < side; ++ow) {
                    const int iw = ow + tw - 1;
                    if ((unsigned)iw >= (unsigned)side) continue;
                    s += dpanel[(size_t)co * S3 * cin + (size_t)((id * side + ih) * side + iw) * cin + ci];
                }
            }
        }
        dw[idx] = s / (float)S3;
    }
}

// The same through a 64 x 64 LDS tile (cin % 64 == 0, cout % 64 == 0): block (tap t, 64 ci, 64 co) sums the valid (i, o) pairs with
// 16-byte reads along ci (a dpanel row is contiguous in k = i * cin + ci) and writes dw rows contiguous in co.
__global__ __launch_bounds__(256) void unpack_meanpool_grad_tiled_kernel(const float *__restrict__ dpanel, float *__restrict__ dw, int side, int cin, int cout) {
    __shared__ float tile[64][65];
    const int S3 = side * side * side, tid = threadIdx.x;
    const int nci = cin >> 6;
    const int t = blockIdx.x / nci, ci0 = (blockIdx.x % nci) * 64, co0 = blockIdx.y * 64;
    const int td = t >> 4, th = (t >> 2) & 3, tw = t & 3;
    const int c4 = tid & 15, r = tid >> 4;                 // ci quad, co row (16 per pass)
    f32x4 s[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) s[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int od = 0; od < side; ++od) {
        const int id = od + td - 1;
        if ((unsigned)id >= (unsigned)side) continue;
        for (int oh = 0; oh < side; ++oh) {
            const int ih = oh + th - 1;
            if ((unsigned)ih >= (unsigned)side) continue;
            for (int ow = 0; ow < side; ++ow) {
                const int iw = ow + tw - 1;
                if ((unsigned)iw >= (unsigned)side) continue;
                const float *src = dpanel + (size_t)co0 * S3 * cin + (size_t)((id * side + ih) * side + iw) * cin + ci0 + 4 * c4;
#pragma unroll
                for (int j = 0; j < 4; ++j) s[j] += *reinterpret_cast<const f32x4 *>(src + (size_t)(r + 16 * j) * S3 * cin);
            }
        }
    }
    const float inv = 1.0f / (float)S3;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) tile[4 * c4 + e][r + 16 * j] = s[j][e] * inv;      // [ci][co]
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int cil = r + 16 * j;
        *reinterpret_cast<f32x4 *>(dw + ((size_t)t * cin + ci0 + cil) * cout + co0 + 4 * c4) =
            f32x4{tile[cil][4 * c4], tile[cil][4 * c4 + 1], tile[cil][4 * c4 + 2], tile[cil][4 * c4 + 3]};
    }
}

__global__ void unpack_convT_dense_grad_kernel(const float *__restrict__ dpanel, float *__restrict__ dw, int side, int cin, int cout) {
    // dw[t][co][ci] = sum_{(o,j) : o - j + 1 = t per axis} dpanel[(o,co)][(j,ci)]
    const int S3 = side * side * side, K = S3 * cin;
    const long total = (long)64 * cin * cout;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int ci = (int)(idx % cin), co = (int)((idx / cin) % cout), t = (int)(idx / ((long)cout * cin));
        const int td = t >> 4, th = (t >> 2) & 3, tw = t & 3;
        float s = 0.f;
        for (int jd = 0; jd < side; ++jd) {
            const int od = jd + td - 1;
            if ((unsigned)od >= (unsigned)side) continue;
            for (int jh = 0; jh < side; ++jh) {
                const int oh = jh + th - 1;
                if ((unsigned)oh >= (unsigned)side) continue;
                for (int jw = 0; jw < side; ++jw) {
                    const int ow = jw + tw - 1;
                    if ((unsigned)ow >= (unsigned)side) continue;
                    const int o = (od * side + oh) * side + ow, j = (jd * side + jh) * side + jw;
                    s += dpanel[((size_t)o * cout + co) * K + (size_t)j * cin + ci];
                }
            }
        }
        dw[idx] = s;
    }
}

__global__ void transpose_kernel(const float *__restrict__ in, float *__restrict__ out, int rows, int cols) {
    __shared__ float tile[32][33];
    const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
    for (int j = threadIdx.y; j < 32; j += 8) {
        const int r = by + j, c = bx + threadIdx.x;
        tile[j][threadIdx.x] = (r < rows && c < cols) ? in[(size_t)r * cols + c] : 0.f;
    }
    __syncthreads();
    for (int j = threadIdx.y; j < 32; j += 8) {
        const int r = bx + j, c = by + threadIdx.x;   // out is [cols][rows]
        if (r < cols && c < rows) out[(size_t)r * rows + c] = tile[threadIdx.x][j];
    }
}

// ------------------------------------------------------------------------------------------------ losses backward
// d(mean_b bce_b)/dlogit: bce = -(g y log q + (1-g)(1-y) log(1-q)), q = clip(sigmoid(l), eps, 1-eps); the clip passes
// the gradient only inside [eps, 1-eps] (tf.clip_by_value).  function.py:73-82, nolbo.py:1432-1433.
__global__ void bce_bwd_kernel(const float *__restrict__ probs, const float *__restrict__ target, float *__restrict__ dlogit,
                               long n, float gamma, float epsilon, float inv_batch) {
    const float hi = 1.0f - epsilon;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float p = probs[i], y = target[i];
        float g = 0.f;
        if (p >= epsilon && p <= hi) g = (-gamma * y * (1.f - p) + (1.f - gamma) * (1.f - y) * p) * inv_batch;
        dlogit[i] = g;
    }
}

// Backward of slice | clip | sampling | dropout | mean_b KL  (nolbo.py:1417-1436; function.py:35-38, 84-98)
__global__ void reparam_kl_bwd_kernel(const float *__restrict__ enc_out, const float *__restrict__ eps, const float *__restrict__ dz,
                                      const float *__restrict__ drop_mask, float drop_scale, float *__restrict__ d_enc_out,
                                      int B, int L, float inv_batch) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * L) return;
    const int b = i / L, j = i % L;
    const float mu = enc_out[(size_t)b * 2 * L + j], raw = enc_out[(size_t)b * 2 * L + L + j];
    const float lv = fminf(fmaxf(raw, -10.f), 10.f);
    float g = dz[i];
    if (drop_mask) g *= drop_mask[i] * drop_scale;
    const float e = expf(lv);
    d_enc_out[(size_t)b * 2 * L + j] = g + mu * inv_batch;
    const float dlv = g * (0.5f * sqrtf(e) * eps[i]) + 0.5f * (e - 1.f) * inv_batch;
    d_enc_out[(size_t)b * 2 * L + L + j] = (raw >= -10.f && raw <= 10.f) ? dlv : 0.f;
}

// Keras Adam (beta1, beta2, epsilon 1e-7; lr_t = lr*sqrt(1-b2^t)/(1-b1^t)):  nolbo.py:1402, 1441
__global__ void adam_kernel(float *__restrict__ p, const float *__restrict__ g, float *__restrict__ m, float *__restrict__ v,
                            long n, float lr_t, float b1, float b2, float eps) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float gi = g[i];
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi; v[i] = vi;
        p[i] -= lr_t * mi / (sqrtf(vi) + eps);
    }
}

// All variables in ONE launch: the caller supplies a device table of chunks (pointers already offset, <= 16384 elements
// each); one workgroup per chunk.  Same arithmetic as adam_kernel.
struct AdamChunk { float *p; const float *g; float *m; float *v; long n; };
__global__ __launch_bounds__(256) void adam_multi_kernel(const AdamChunk *__restrict__ table, float lr_t, float b1, float b2, float eps) {
    const AdamChunk c = table[blockIdx.x];
    for (long i = threadIdx.x; i < c.n; i += 256) {
        const float gi = c.g[i];
        const float mi = b1 * c.m[i] + (1.f - b1) * gi;
        const float vi = b2 * c.v[i] + (1.f - b2) * gi * gi;
        c.m[i] = mi; c.v[i] = vi;
        c.p[i] -= lr_t * mi / (sqrtf(vi) + eps);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T *__restrict__ x, float *__restrict__ out, long R, int C) {
    // out[c] = sum_r x[r][c] (Dense bias gradient; R = batch): 32 columns x 8 row slices per workgroup, slice sums added in order
    __shared__ float red[8][32];
    const int cl = threadIdx.x & 31, sl = threadIdx.x >> 5, c = blockIdx.x * 32 + cl;
    float s = 0.f;
    if (c < C) {
        long r = sl;
        for (; r + 24 < R; r += 32) {
            const float v0 = (float)x[r * C + c], v1 = (float)x[(r + 8) * C + c], v2 = (float)x[(r + 16) * C + c], v3 = (float)x[(r + 24) * C + c];
            s += v0; s += v1; s += v2; s += v3;
        }
        for (; r < R; r += 8) s += (float)x[r * C + c];
    }
    red[sl][cl] = s;
    __syncthreads();
    if (sl == 0 && c < C) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) t += red[k][cl];
        out[c] = t;
    }
}

inline int grid_1d(long n) {
    long g = (n + 255) / 256;
    return (int)(g < 1 ? 1 : (g > 16384 ? 16384 : g));
}

// Row blocks of the two BatchNorm sweeps: 256 rows per block on the long layers (at most 1024 blocks), but never fewer
// blocks than keep ~16 rows per block -- a 2048-row x 512-channel layer is 128 blocks, not 8.
// A thread owns V consecutive channels (4 floats / 8 bf16) and a workgroup spans whole rows: C % V == 0, C / V <= 256
inline bool bn_shape_ok(long rows, int channels, int dtype) {
    const int V = dtype == VV_BF16 ? 8 : 4;
    if (rows <= 0 || channels <= 0 || channels % V || channels / V > 256) return false;
    return channels / V >= 64 || 256 % (channels / V) == 0;
}

// Grid of the two element-wise sweeps: a block covers 256 / (C / V) rows per step; two steps per block on the short
// layers, at most 4096 blocks (16 per CU) on the long ones
inline int bn_sweep_blocks(long R, int C, int V) {
    const int cvn = C / V;
    const int rows_par = 256 / cvn > 0 ? 256 / cvn : 1;
    long nb = (R + 2L * rows_par - 1) / (2L * rows_par);
    static const long cap = vv_hook("VV_BN_SWEEP") ? atol(vv_hook("VV_BN_SWEEP")) : 4096;
    return (int)(nb > cap ? cap : (nb < 1 ? 1 : nb));
}

inline int bn_blocks(long R, int C, int V = 4) {
    const int cvn = C / V;
    const int rows_par = 256 / cvn > 0 ? 256 / cvn : 1;
    long per = 8L * rows_par;                   // two trips of the 4-rows-in-flight loop
    if (per < 16) per = 16;
    if (per > 256) per = 256;
    long nb = (R + per - 1) / per;
    static const long cap = vv_hook("VV_BN_NB") ? atol(vv_hook("VV_BN_NB")) : 1024;
    return (int)(nb > cap ? cap : (nb < 1 ? 1 : nb));
}

}  // namespace

VV_EXPORT size_t vv_bn_workspace_bytes(long rows, int channels) { return (size_t)bn_blocks(rows, channels) * 2 * channels * sizeof(float); }

VV_EXPORT int vv_bn_train_stats(const void *x, long rows, int channels, const float *gamma, const float *beta, float eps,
                                float momentum, float *mean, float *var, float *rstd, float *scale, float *shift,
                                float *moving_mean, float *moving_var, int dtype, void *workspace, size_t workspace_bytes, void *stream) {
    if (!x || !gamma || !beta || !mean || !var || !rstd || !scale || !shift) return VV_ERR_NULL;
    if (dtype != VV_F32 && dtype != VV_BF16) return VV_ERR_DTYPE;
    if (!bn_shape_ok(rows, channels, dtype)) return VV_ERR_SHAPE;
    if (!workspace || workspace_bytes < vv_bn_workspace_bytes(rows, channels)) return VV_ERR_WORKSPACE;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int nb = bn_blocks(rows, channels, dtype == VV_BF16 ? 8 : 4);     // <= the float32 count the workspace is sized for
    const int rpb = (int)((rows + nb - 1) / nb);
    float *part = reinterpret_cast<float *>(workspace);
    if (dtype == VV_BF16)
        VV_LAUNCH((bn_reduce_kernel<0, __bf16>), dim3(nb), dim3(256), 0, st, reinterpret_cast<const __bf16 *>(x), nullptr, nullptr, nullptr,
                  nullptr, nullptr, part, rows, channels, rpb, 0);
    else
        VV_LAUNCH((bn_reduce_kernel<0, float>), dim3(nb), dim3(256), 0, st, reinterpret_cast<const float *>(x), nullptr, nullptr, nullptr,
                  nullptr, nullptr, part, rows, channels, rpb, 0);
    VV_LAUNCH(bn_stats_finalize_kernel, dim3((channels + BN_FC - 1) / BN_FC), dim3(256), 0, st, part, nb, rows, channels, gamma, beta, eps,
              momentum, mean, var, rstd, scale, shift, moving_mean, moving_var);
    return vv_launch_status();
}

// The second half of vv_bn_train_stats for a producer that left the per-block column sums itself (vv_convT3d_k4s2_whole_stats_fwd):
// partial[(block * 2 + {0: sum, 1: sum of squares}) * channels + channel] over `nblocks` blocks that together cover `rows` rows.
VV_EXPORT int vv_bn_finalize_stats(const float *partial, int nblocks, long rows, int channels, const float *gamma, const float *beta, float eps,
                                   float momentum, float *mean, float *var, float *rstd, float *scale, float *shift, float *moving_mean,
                                   float *moving_var, void *stream) {
    if (!partial || !gamma || !beta || !mean || !var || !rstd || !scale || !shift) return VV_ERR_NULL;
    if (nblocks <= 0 || rows <= 0 || channels <= 0) return VV_ERR_SHAPE;
    VV_LAUNCH(bn_stats_finalize_kernel, dim3((channels + BN_FC - 1) / BN_FC), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), partial, nblocks,
              rows, channels, gamma, beta, eps, momentum, mean, var, rstd, scale, shift, moving_mean, moving_var);
    return vv_launch_status();
}

VV_EXPORT int vv_bn_act_fwd(const void *x, const float *scale, const float *shift, void *y, long rows, int channels, int act,
                            int dtype, void *stream) {
    if (!x || !scale || !shift || !y) return VV_ERR_NULL;
    if (dtype != VV_F32 && dtype != VV_BF16) return VV_ERR_DTYPE;
    if (!bn_shape_ok(rows, channels, dtype)) return VV_ERR_SHAPE;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int grid = bn_sweep_blocks(rows, channels, dtype == VV_BF16 ? 8 : 4);
    if (dtype == VV_BF16)
        VV_LAUNCH(bn_act_fwd_kernel<__bf16>, dim3(grid), dim3(256), 0, st, reinterpret_cast<const __bf16 *>(x), scale, shift,
                  reinterpret_cast<__bf16 *>(y), rows, channels, act);
    else
        VV_LAUNCH(bn_act_fwd_kernel<float>, dim3(grid), dim3(256), 0, st, reinterpret_cast<const float *>(x), scale, shift,
                  reinterpret_cast<float *>(y), rows, channels, act);
    return vv_launch_status();
}

VV_EXPORT int vv_bn_act_bwd(const void *x, const void *dy, const float *scale, const float *shift, const float *mean,
                            const float *rstd, float *dgamma, float *dbeta, void *dx, long rows, int channels, int act,
                            int dtype, void *workspace, size_t workspace_bytes, void *stream) {
    if (!x || !dy || !scale || !shift || !mean || !rstd || !dgamma || !dbeta || !dx) return VV_ERR_NULL;
    if (dtype != VV_F32 && dtype != VV_BF16) return VV_ERR_DTYPE;
    if (!bn_shape_ok(rows, channels, dtype)) return VV_ERR_SHAPE;
    if (!workspace || workspace_bytes < vv_bn_workspace_bytes(rows, channels)) return VV_ERR_WORKSPACE;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int nb = bn_blocks(rows, channels, dtype == VV_BF16 ? 8 : 4);     // <= the float32 count the workspace is sized for
    const int rpb = (int)((rows + nb - 1) / nb);
    float *part = reinterpret_cast<float *>(workspace);
    const int sweep = bn_sweep_blocks(rows, channels, dtype == VV_BF16 ? 8 : 4);
    if (dtype == VV_BF16) {
        const __bf16 *xb = reinterpret_cast<const __bf16 *>(x), *dyb = reinterpret_cast<const __bf16 *>(dy);
        if (act == VV_ACT_ELU) VV_LAUNCH((bn_reduce_kernel<1, __bf16, VV_ACT_ELU>), dim3(nb), dim3(256), 0, st, xb, dyb, scale, shift, mean, rstd, part, rows, channels, rpb, act);
        else VV_LAUNCH((bn_reduce_kernel<1, __bf16>), dim3(nb), dim3(256), 0, st, xb, dyb, scale, shift, mean, rstd, part, rows, channels, rpb, act);
        VV_LAUNCH(bn_bwd_finalize_kernel, dim3((channels + BN_FC - 1) / BN_FC), dim3(256), 0, st, part, nb, channels, dgamma, dbeta);
        VV_LAUNCH(bn_act_bwd_kernel<__bf16>, dim3(sweep), dim3(256), 0, st, xb, dyb, scale, shift, mean, rstd, dgamma, dbeta,
                  reinterpret_cast<__bf16 *>(dx), rows, channels, 1.0f / (float)rows, act);
    } else {
        const float *xf = reinterpret_cast<const float *>(x), *dyf = reinterpret_cast<const float *>(dy);
        VV_LAUNCH((bn_reduce_kernel<1, float>), dim3(nb), dim3(256), 0, st, xf, dyf, scale, shift, mean, rstd, part, rows, channels, rpb, act);
        VV_LAUNCH(bn_bwd_finalize_kernel, dim3((channels + BN_FC - 1) / BN_FC), dim3(256), 0, st, part, nb, channels, dgamma, dbeta);
        VV_LAUNCH(bn_act_bwd_kernel<float>, dim3(sweep), dim3(256), 0, st, xf, dyf, scale, shift, mean, rstd, dgamma, dbeta,
                  reinterpret_cast<float *>(dx), rows, channels, 1.0f / (float)rows, act);
    }
    return vv_launch_status();
}

namespace {
struct WgradPlan { int bn, splits, rps; size_t ws; };
WgradPlan wgrad_plan(long R, int M, int N) {
    WgradPlan p;
    p.bn = (N % 128 == 0) ? 128 : 64;
    const long tiles = (long)((M + 63) / 64) * ((N + p.bn - 1) / p.bn);
    long chunks = (R + 31) / 32;
    long splits = 1;
    while (tiles * splits < 1024 && splits * 2 <= chunks && splits < 256) splits *= 2;
    p.rps = (int)(((chunks + splits - 1) / splits) * 32);
    p.splits = (int)((R + p.rps - 1) / p.rps);
    p.ws = (size_t)p.splits * M * N * sizeof(float);
    return p;
}

// bf16 kernel: 128 x 128 tiles, 64-row chunks
WgradPlan wgrad_plan_bf16(long R, int M, int N) {
    WgradPlan p;
    p.bn = 128;
    const long tiles = (long)((M + 127) / 128) * ((N + 127) / 128);
    long chunks = (R + 63) / 64;
    long splits = 1;
    // one or two tiles (the single-channel layers, M = N = 64): the kernel is a latency-bound stream over the rows, so
    // four workgroups per CU's worth of splits instead of two
    const long want = tiles <= 2 ? 1024 : 512;
    while (tiles * splits < want && splits * 2 <= chunks && splits < want) splits *= 2;
    p.rps = (int)(((chunks + splits - 1) / splits) * 64);
    p.splits = (int)((R + p.rps - 1) / p.rps);
    p.ws = (size_t)p.splits * M * N * sizeof(float);
    return p;
}

bool wgrad_bf16_ok(const void *A, const void *G, long R, int M, int N, int lda, int cin, int amode, size_t a_elems) {
    if (vv_hook("VV_WGRAD_F32")) return false;
    if (M % 32 || N % 32) return false;
    if (amode == 0 && lda % 8) return false;
    if (amode == 1 && cin % 64) return false;
    if (a_elems * 2 >= 0xFFFFFFF0ull || (size_t)R * N * 2 >= 0xFFFFFFF0ull) return false;
    return vv_aligned16(A) && vv_aligned16(G);
}

void launch_wgrad_reduce(const float *slabs, float *out, long n, int splits, float alpha, int accumulate, hipStream_t st) {
    if (splits >= 16 && n <= 65536)
        VV_LAUNCH(wgrad_reduce_sliced_kernel, dim3((unsigned)((n + 63) / 64)), dim3(256), 0, st, slabs, out, n, splits, alpha, accumulate);
    else
        VV_LAUNCH(wgrad_reduce_kernel, dim3(grid_1d(n)), dim3(256), 0, st, slabs, out, n, splits, alpha, accumulate);
}

template <int AMODE>
int launch_wgrad(const WgradArgs &a, const WgradPlan &p, float *out, float alpha, int accumulate, hipStream_t st) {
    const int tiles = ((a.M + 63) / 64) * ((a.N + p.bn - 1) / p.bn);
    if (p.bn == 128) VV_LAUNCH((wgrad_kernel<AMODE, 128>), dim3(tiles, p.splits), dim3(256), 0, st, a);
    else VV_LAUNCH((wgrad_kernel<AMODE, 64>), dim3(tiles, p.splits), dim3(256), 0, st, a);
    const long n = (long)a.M * a.N;
    launch_wgrad_reduce(a.slabs, out, n, p.splits, alpha, accumulate, st);
    return vv_launch_status();
}

template <int AMODE>
int launch_wgrad_bf16(const WgradBArgs &a_in, const WgradPlan &p, float *out, hipStream_t st) {
    WgradBArgs a = a_in;
    if (p.splits == 1) a.slabs = out;              // one share: its "slab" IS the result (the reduce pass was a 33 MB copy on the 256 <-> 512 layers)
    const int tiles = ((a.M + 127) / 128) * ((a.N + 127) / 128);
    static const bool attr = [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&wgrad_bf16_kernel<AMODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
        return true;
    }();
    (void)attr;
    if (AMODE == 2 && a.N <= 64) VV_LAUNCH((wgrad_bf16_kernel<2, true>), dim3(tiles, p.splits), dim3(256), 32768, st, a);
    else VV_LAUNCH((wgrad_bf16_kernel<AMODE>), dim3(tiles, p.splits), dim3(256), 65536, st, a);
    const long n = (long)a.M * a.N;
    if (p.splits > 1) launch_wgrad_reduce(a.slabs, out, n, p.splits, 1.f, 0, st);
    return vv_launch_status();
}
}  // namespace

VV_EXPORT size_t vv_wgrad_workspace_bytes(long rows, int m, int n) {
    const size_t a = wgrad_plan(rows, m, n).ws;     // the caller need not know which kernel runs
    size_t b = wgrad_plan_bf16(rows, m, n).ws;
    const size_t c = m % 4096 == 0 ? vv_wgrad_phase_ws(rows, m / 64, n) : 0;     // conv layers: the phase kernel's slabs
    if (c > b) b = c;
    return a > b ? a : b;
}

VV_EXPORT int vv_wgrad_dense(const void *a, const void *g, float *dw, long rows, int m, int n, int lda, int a_dtype, int g_dtype,
                             void *workspace, size_t workspace_bytes, void *stream) {
    if (!a || !g || !dw) return VV_ERR_NULL;
    if ((a_dtype != VV_F32 && a_dtype != VV_BF16) || (g_dtype != VV_F32 && g_dtype != VV_BF16)) return VV_ERR_DTYPE;
    if (rows <= 0 || m <= 0 || n <= 0 || m % 4 || n % 4 || lda % 4) return VV_ERR_SHAPE;
    if (!workspace || workspace_bytes < vv_wgrad_workspace_bytes(rows, m, n)) return VV_ERR_WORKSPACE;
    if (a_dtype == VV_BF16 && g_dtype == VV_BF16 && wgrad_bf16_ok(a, g, rows, m, n, lda, 0, 0, (size_t)rows * lda)) {
        const WgradPlan pb = wgrad_plan_bf16(rows, m, n);
        WgradBArgs wb{a, g, reinterpret_cast<float *>(workspace), rows, m, n, lda, 0, 0, pb.rps, (unsigned)((size_t)rows * lda * 2),
                      (unsigned)((size_t)rows * n * 2)};
        return launch_wgrad_bf16<0>(wb, pb, dw, reinterpret_cast<hipStream_t>(stream));
    }
    const WgradPlan p = wgrad_plan(rows, m, n);
    WgradArgs w{a, g, reinterpret_cast<float *>(workspace), rows, m, n, lda, 0, 0, p.rps, a_dtype == VV_BF16, g_dtype == VV_BF16};
    return launch_wgrad<0>(w, p, dw, 1.f, 0, reinterpret_cast<hipStream_t>(stream));
}

VV_EXPORT int vv_wgrad_conv_k4s2(const void *src, const void *g, float *dw, int batch, int side, int cin, int cout, int src_dtype,
                                 int g_dtype, void *workspace, size_t workspace_bytes, void *stream) {
    if (!src || !g || !dw) return VV_ERR_NULL;
    if ((src_dtype != VV_F32 && src_dtype != VV_BF16) || (g_dtype != VV_F32 && g_dtype != VV_BF16)) return VV_ERR_DTYPE;
    if (cin == 1 && src_dtype != VV_F32) return VV_ERR_DTYPE;               // single-channel sources are float32 grids
    if (batch <= 0 || side < 2 || !vv_is_pow2(side) || cout % 4 || (cin != 1 && cin % 64)) return VV_ERR_SHAPE;
    const int o = side / 2;
    const long rows = (long)batch * o * o * o;
    const int m = 64 * cin;
    if (!workspace || workspace_bytes < vv_wgrad_workspace_bytes(rows, m, cout)) return VV_ERR_WORKSPACE;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const size_t src_elems = (size_t)batch * side * side * side * cin;
    if (cin != 1 && src_dtype == VV_BF16 && g_dtype == VV_BF16 && vv_wgrad_phase_ok(src, g, batch, side, cin, cout)) {
        int splits = 0;
        vv_wgrad_phase_launch(src, g, reinterpret_cast<float *>(workspace), batch, side, cin, cout, &splits, st);
        launch_wgrad_reduce(reinterpret_cast<const float *>(workspace), dw, (long)m * cout, splits, 1.f, 0, st);
        return vv_launch_status();
    }
    if (cin != 1 && src_dtype == VV_BF16 && g_dtype == VV_BF16 && wgrad_bf16_ok(src, g, rows, m, cout, 0, cin, 1, src_elems)) {
        const WgradPlan pb = wgrad_plan_bf16(rows, m, cout);
        WgradBArgs wb{src, g, reinterpret_cast<float *>(workspace), rows, m, cout, 0, vv_log2(side), cin, pb.rps,
                      (unsigned)(src_elems * 2), (unsigned)((size_t)rows * cout * 2)};
        return launch_wgrad_bf16<1>(wb, pb, dw, st);
    }
    if (cin == 1 && side >= 4 && g_dtype == VV_BF16 && cout % 32 == 0 && (size_t)rows * cout * 2 < 0xFFFFFFF0ull && vv_aligned16(g) && !vv_hook("VV_WGRAD_F32")) {
        // single input channel: the 64-tap rows are built from the float32 grid inside the bf16 kernel's staging
        const WgradPlan pb = wgrad_plan_bf16(rows, 64, cout);
        WgradBArgs wb{src, g, reinterpret_cast<float *>(workspace), rows, 64, cout, 0, vv_log2(side), 1, pb.rps, 0u,
                      (unsigned)((size_t)rows * cout * 2)};
        return launch_wgrad_bf16<2>(wb, pb, dw, st);
    }
    const WgradPlan p = wgrad_plan(rows, m, cout);
    WgradArgs w{src, g, reinterpret_cast<float *>(workspace), rows, m, cout, 0, vv_log2(side), cin, p.rps, src_dtype == VV_BF16,
                g_dtype == VV_BF16};
    return cin == 1 ? launch_wgrad<2>(w, p, dw, 1.f, 0, st) : launch_wgrad<1>(w, p, dw, 1.f, 0, st);
}

VV_EXPORT int vv_unpack_meanpool_grad(const float *dpanel, float *dw, int side, int cin, int cout, void *stream) {
    if (!dpanel || !dw) return VV_ERR_NULL;
    if (cin % 64 == 0 && cout % 64 == 0 && side > 0 && vv_aligned16(dpanel) && vv_aligned16(dw)) {
        VV_LAUNCH(unpack_meanpool_grad_tiled_kernel, dim3(64 * (cin / 64), cout / 64), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), dpanel, dw,
                  side, cin, cout);
        return vv_launch_status();
    }
    VV_LAUNCH(unpack_meanpool_grad_kernel, dim3(grid_1d((long)64 * cin * cout)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
              dpanel, dw, side, cin, cout);
    return vv_launch_status();
}

VV_EXPORT int vv_unpack_convT_dense_grad(const float *dpanel, float *dw, int side, int cin, int cout, void *stream) {
    if (!dpanel || !dw) return VV_ERR_NULL;
    VV_LAUNCH(unpack_convT_dense_grad_kernel, dim3(grid_1d((long)64 * cin * cout)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
              dpanel, dw, side, cin, cout);
    return vv_launch_status();
}

VV_EXPORT int vv_transpose_f32(const float *in, float *out, int rows, int cols, void *stream) {
    if (!in || !out) return VV_ERR_NULL;
    if (rows <= 0 || cols <= 0) return VV_ERR_SHAPE;
    VV_LAUNCH(transpose_kernel, dim3((cols + 31) / 32, (rows + 31) / 32), dim3(32, 8), 0, reinterpret_cast<hipStream_t>(stream), in, out, rows, cols);
    return vv_launch_status();
}

VV_EXPORT int vv_bce_bwd(const float *probs, const float *target, float *dlogit, int batch, long voxels, float gamma, float epsilon,
                         float inv_batch, void *stream) {
    if (!probs || !target || !dlogit) return VV_ERR_NULL;
    if (batch <= 0 || voxels <= 0) return VV_ERR_SHAPE;
    const long n = (long)batch * voxels;
    VV_LAUNCH(bce_bwd_kernel, dim3(grid_1d(n)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), probs, target, dlogit, n, gamma,
              epsilon, inv_batch);
    return vv_launch_status();
}

VV_EXPORT int vv_reparam_kl_bwd(const float *enc_out, const float *eps, const float *dz, const float *drop_mask, float drop_scale,
                                float *d_enc_out, int batch, int latent, float inv_batch, void *stream) {
    if (!enc_out || !eps || !dz || !d_enc_out) return VV_ERR_NULL;
    if (batch <= 0 || latent <= 0) return VV_ERR_SHAPE;
    VV_LAUNCH(reparam_kl_bwd_kernel, dim3((batch * latent + 255) / 256), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), enc_out,
              eps, dz, drop_mask, drop_scale, d_enc_out, batch, latent, inv_batch);
    return vv_launch_status();
}

VV_EXPORT int vv_adam_step(float *param, const float *grad, float *m, float *v, long n, float lr_t, float beta1, float beta2,
                           float epsilon, void *stream) {
    if (!param || !grad || !m || !v) return VV_ERR_NULL;
    if (n <= 0) return VV_ERR_SHAPE;
    VV_LAUNCH(adam_kernel, dim3(grid_1d(n)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), param, grad, m, v, n, lr_t, beta1,
              beta2, epsilon);
    return vv_launch_status();
}

VV_EXPORT int vv_adam_step_multi(const void *chunk_table, int nchunks, float lr_t, float beta1, float beta2, float epsilon, void *stream) {
    if (!chunk_table) return VV_ERR_NULL;
    if (nchunks <= 0) return VV_ERR_SHAPE;
    VV_LAUNCH(adam_multi_kernel, dim3(nchunks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
              reinterpret_cast<const AdamChunk *>(chunk_table), lr_t, beta1, beta2, epsilon);
    return vv_launch_status();
}

VV_EXPORT int vv_colsum(const void *x, float *out, long rows, int cols, int dtype, void *stream) {
    if (!x || !out) return VV_ERR_NULL;
    if (dtype != VV_F32 && dtype != VV_BF16) return VV_ERR_DTYPE;
    if (rows <= 0 || cols <= 0) return VV_ERR_SHAPE;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (dtype == VV_BF16) VV_LAUNCH(colsum_kernel<__bf16>, dim3((cols + 31) / 32), dim3(256), 0, st, reinterpret_cast<const __bf16 *>(x), out, rows, cols);
    else VV_LAUNCH(colsum_kernel<float>, dim3((cols + 31) / 32), dim3(256), 0, st, reinterpret_cast<const float *>(x), out, rows, cols);
    return vv_launch_status();
}

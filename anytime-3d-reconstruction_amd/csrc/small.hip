// Weight packing, BatchNorm folding, the fused latent op and the metric finalisation: small, HBM/latency-bound
// kernels around the MFMA convolutions.  gfx950 only.
#include "common.h"

namespace {

template <typename T>
__global__ void pack_conv_k4_kernel(const float *__restrict__ w, T *__restrict__ out, int cin, int cout) {
    // out[co][t*cin + ci] = w[t][ci][co]
    const size_t total = (size_t)64 * cin * cout;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int K = 64 * cin;
        const int co = (int)(i / K), k = (int)(i % K);
        const int t = k / cin, ci = k % cin;
        vv_store(out, i, w[((size_t)t * cin + ci) * cout + co]);
    }
}

// The same transposition through a 64 x 64 LDS tile (cout % 64 == 0; K = 64 cin always is): float4 reads along co,
// 16 consecutive k per thread on the way out -- both sides coalesced (the element-wise form reads with a stride of cout
// floats and spent 36 us on the 8 M-element layer).
template <typename T>
__global__ __launch_bounds__(256) void pack_conv_k4_tiled_kernel(const float *__restrict__ w, T *__restrict__ out, int cin, int cout) {
    __shared__ float tile[64][65];
    const int K = 64 * cin, tid = threadIdx.x;
    const int k0 = blockIdx.x * 64, co0 = blockIdx.y * 64;
    const int c4 = tid & 15, r = tid >> 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int kl = r + 16 * j;
        const f32x4 v = *reinterpret_cast<const f32x4 *>(w + (size_t)(k0 + kl) * cout + co0 + 4 * c4);
#pragma unroll
        for (int e = 0; e < 4; ++e) tile[kl][4 * c4 + e] = v[e];
    }
    __syncthreads();
    const int col = tid >> 2, kq = (tid & 3) * 16;
    T *dst = out + (size_t)(co0 + col) * K + k0 + kq;
#pragma unroll
    for (int e = 0; e < 16; e += 4) {
        const f32x4 v = {tile[kq + e][col], tile[kq + e + 1][col], tile[kq + e + 2][col], tile[kq + e + 3][col]};
        if constexpr (sizeof(T) == 4) {
            *reinterpret_cast<f32x4 *>(dst + e) = v;
        } else if constexpr (sizeof(T) == 1) {
            *reinterpret_cast<unsigned *>(dst + e) = vv_pack_fp8x4(v);
        } else {
            bf16x4 o;
#pragma unroll
            for (int u = 0; u < 4; ++u) o[u] = static_cast<__bf16>(v[u]);
            *reinterpret_cast<bf16x4 *>(dst + e) = o;
        }
    }
}

// four consecutive ci per thread (cin % 4 == 0): 16-byte reads, 8/16-byte writes, a quarter of the index arithmetic
template <typename T>
__global__ void pack_convT_k4s2_vec_kernel(const float *__restrict__ w, T *__restrict__ out, int cin, int cout) {
    const int K = 8 * cin, K4 = K >> 2;
    const size_t total4 = (size_t)8 * cout * K4;
    for (size_t i4 = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i4 < total4; i4 += (size_t)gridDim.x * blockDim.x) {
        const int k = (int)(i4 % K4) * 4;
        const size_t pc = i4 / K4;
        const int co = (int)(pc % cout), p = (int)(pc / cout);
        const int a = k / cin, ci = k % cin;
        const int td = 1 - ((p >> 2) & 1) + 2 * ((a >> 2) & 1);
        const int th = 1 - ((p >> 1) & 1) + 2 * ((a >> 1) & 1);
        const int tw = 1 - (p & 1) + 2 * (a & 1);
        const int t = (td * 4 + th) * 4 + tw;
        const f32x4 v = *reinterpret_cast<const f32x4 *>(w + ((size_t)t * cout + co) * cin + ci);
        if constexpr (sizeof(T) == 4) {
            reinterpret_cast<f32x4 *>(out)[i4] = v;
        } else if constexpr (sizeof(T) == 1) {
            reinterpret_cast<unsigned *>(out)[i4] = vv_pack_fp8x4(v);
        } else {
            bf16x4 o;
#pragma unroll
            for (int u = 0; u < 4; ++u) o[u] = static_cast<__bf16>(v[u]);
            reinterpret_cast<bf16x4 *>(out)[i4] = o;
        }
    }
}

template <typename T>
__global__ void pack_convT_k4s2_kernel(const float *__restrict__ w, T *__restrict__ out, int cin, int cout) {
    // out[p][co][a*cin + ci] = w[t(p,a)][co][ci],  t = 1 - p + 2a per axis
    const int K = 8 * cin;
    const size_t total = (size_t)8 * cout * K;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int k = (int)(i % K);
        const size_t pc = i / K;
        const int co = (int)(pc % cout), p = (int)(pc / cout);
        const int a = k / cin, ci = k % cin;
        const int td = 1 - ((p >> 2) & 1) + 2 * ((a >> 2) & 1);
        const int th = 1 - ((p >> 1) & 1) + 2 * ((a >> 1) & 1);
        const int tw = 1 - (p & 1) + 2 * (a & 1);
        const int t = (td * 4 + th) * 4 + tw;
        vv_store(out, i, w[((size_t)t * cout + co) * cin + ci]);
    }
}

template <typename T>
__global__ void pack_conv_k4s1_meanpool_kernel(const float *__restrict__ w, T *__restrict__ out, int side, int cin, int cout) {
    // out[co][i*cin + ci] = (1/S^3) * sum_{o : 0 <= i - o + 1 <= 3 per axis} w[i - o + 1][ci][co]
    const int S3 = side * side * side, K = S3 * cin;
    const size_t total = (size_t)cout * K;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int co = (int)(idx / K), k = (int)(idx % K);
        const int i = k / cin, ci = k % cin;
        const int iw = i % side, ih = (i / side) % side, id = i / (side * side);
        float s = 0.f;
        for (int od = 0; od < side; ++od) {
            const int td = id - od + 1;
            if (td < 0 || td > 3) continue;
            for (int oh = 0; oh < side; ++oh) {
                const int th = ih - oh + 1;
                if (th < 0 || th > 3) continue;
                for (int ow = 0; ow < side; ++ow) {
                    const int tw = iw - ow + 1;
                    if (tw < 0 || tw > 3) continue;
                    s += w[((size_t)((td * 4 + th) * 4 + tw) * cin + ci) * cout + co];
                }
            }
        }
        vv_store(out, idx, s / (float)S3);
    }
}

// The same panel through a 64 x 64 LDS tile (cin % 64 == 0, cout % 64 == 0): block (input position i, 64 ci, 64 co) sums the valid
// taps with 16-byte reads along co and writes 64 co-rows of 64 consecutive k -- both sides coalesced.  The element-wise form reads
// with a stride of cout floats and took 20 us for the 4 M-element layer; the training step packs it twice per step.
template <typename T>
__global__ __launch_bounds__(256) void pack_conv_k4s1_meanpool_tiled_kernel(const float *__restrict__ w, T *__restrict__ out, int side, int cin, int cout) {
    __shared__ float tile[64][65];
    const int S3 = side * side * side, K = S3 * cin, tid = threadIdx.x;
    const int nci = cin >> 6;
    const int i = blockIdx.x / nci, ci0 = (blockIdx.x % nci) * 64, co0 = blockIdx.y * 64;
    const int iw = i % side, ih = (i / side) % side, id = i / (side * side);
    const int c4 = tid & 15, r = tid >> 4;
    f32x4 s[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) s[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int od = 0; od < side; ++od) {
        const int td = id - od + 1;
        if (td < 0 || td > 3) continue;
        for (int oh = 0; oh < side; ++oh) {
            const int th = ih - oh + 1;
            if (th < 0 || th > 3) continue;
            for (int ow = 0; ow < side; ++ow) {
                const int tw = iw - ow + 1;
                if (tw < 0 || tw > 3) continue;
                const float *wt = w + ((size_t)((td * 4 + th) * 4 + tw) * cin + ci0) * cout + co0 + 4 * c4;
#pragma unroll
                for (int j = 0; j < 4; ++j) s[j] += *reinterpret_cast<const f32x4 *>(wt + (size_t)(r + 16 * j) * cout);
            }
        }
    }
    const float inv = 1.0f / (float)S3;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) tile[r + 16 * j][4 * c4 + e] = s[j][e] * inv;
    __syncthreads();
    const int col = tid >> 2, kq = (tid & 3) * 16;
#pragma unroll
    for (int e = 0; e < 16; ++e) vv_store(out, (size_t)(co0 + col) * K + (size_t)i * cin + ci0 + kq + e, tile[kq + e][col]);
}

template <typename T>
__global__ void pack_conv_k4s1_full_kernel(const float *__restrict__ w, T *__restrict__ out, int side, int cin, int cout) {
    // out[(o,co)][(i,ci)] = w[i - o + 1][ci][co] (0 when a tap index leaves 0..3): the final encoder conv (k4 s1 SAME, pad 1 before /
    // 2 after) position by position, for the pooling modes that are not linear (final_pool = 'max')
    const int S3 = side * side * side, K = S3 * cin;
    const size_t total = (size_t)S3 * cout * K;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int k = (int)(idx % K);
        const int n = (int)(idx / K);
        const int o = n / cout, co = n % cout;
        const int i = k / cin, ci = k % cin;
        const int td = i / (side * side) - o / (side * side) + 1, th = (i / side) % side - (o / side) % side + 1, tw = i % side - o % side + 1;
        const bool ok = td >= 0 && td <= 3 && th >= 0 && th <= 3 && tw >= 0 && tw <= 3;
        vv_store(out, idx, ok ? w[((size_t)((td * 4 + th) * 4 + tw) * cin + ci) * cout + co] : 0.f);
    }
}

// out[b][c] = max over positions p of x[b][p][c] (tf.reduce_max over the spatial axes, autoencoder3D.py:92-93)
// tf.nn.sigmoid on the encoder output (autoencoder3D.py:97-99, final_activation 'sigmoid'); in place allowed
__global__ void sigmoid_f32_kernel(const float *__restrict__ x, float *__restrict__ y, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) y[i] = 1.0f / (1.0f + expf(-x[i]));
}

__global__ void max_over_positions_kernel(const float *__restrict__ x, float *__restrict__ out, int batch, int npos, int channels) {
    const size_t total = (size_t)batch * channels;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const size_t b = idx / channels, c = idx % channels;
        float m = x[(b * npos) * channels + c];
        for (int p = 1; p < npos; ++p) m = fmaxf(m, x[(b * npos + p) * channels + c]);
        out[idx] = m;
    }
}

template <typename T>
__global__ void pack_convT_k4s1_dense_kernel(const float *__restrict__ w, T *__restrict__ out, int side, int cin, int cout) {
    // out[(o,co)][(j,ci)] = w[o - j + 1][co][ci] (0 when a tap index leaves 0..3)
    const int S3 = side * side * side, K = S3 * cin;
    const size_t total = (size_t)S3 * cout * K;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int k = (int)(idx % K);
        const size_t n = idx / K;
        const int co = (int)(n % cout), o = (int)(n / cout);
        const int j = k / cin, ci = k % cin;
        const int td = o / (side * side) - j / (side * side) + 1;
        const int th = (o / side) % side - (j / side) % side + 1;
        const int tw = o % side - j % side + 1;
        float v = 0.f;
        if ((unsigned)td < 4u && (unsigned)th < 4u && (unsigned)tw < 4u)
            v = w[((size_t)((td * 4 + th) * 4 + tw) * cout + co) * cin + ci];
        vv_store(out, idx, v);
    }
}

template <typename T>
__global__ void pack_dense_kernel(const float *__restrict__ w, T *__restrict__ out, int in, int outn) {
    const size_t total = (size_t)in * outn;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int o = (int)(idx / in), i = (int)(idx % in);
        vv_store(out, idx, w[(size_t)i * outn + o]);
    }
}

__global__ void fold_bn_kernel(const float *gamma, const float *beta, const float *mean, const float *var,
                               const float *bias, float eps, float *scale, float *shift, int c, int repeat) {
    const int total = c * repeat;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int ch = i % c;
        const float sc = gamma[ch] / sqrtf(var[ch] + eps);
        scale[i] = sc;
        shift[i] = beta[ch] + ((bias ? bias[ch] : 0.f) - mean[ch]) * sc;
    }
}

// One wave per sample row: slice | clip | sqrt(exp(lv))*eps | dropout | KL row sum by wave shuffles.
template <typename TA>
__global__ __launch_bounds__(256) void reparam_kl_kernel(const float *__restrict__ enc_out, const float *__restrict__ eps,
                                                         const float *__restrict__ drop_mask, float drop_scale,
                                                         float *__restrict__ z, TA *__restrict__ z_act,
                                                         float *__restrict__ kl, float *__restrict__ mean_out,
                                                         float *__restrict__ logvar_out, int batch, int L) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (b >= batch) return;
    float s = 0.f;
    for (int j = lane; j < L; j += 64) {
        const float mu = enc_out[(size_t)b * 2 * L + j];
        float lv = enc_out[(size_t)b * 2 * L + L + j];
        lv = fminf(fmaxf(lv, -10.f), 10.f);                    // nolbo.py:1420
        const float e = expf(lv);
        float zz = mu + sqrtf(e) * eps[(size_t)b * L + j];     // function.py:37
        if (drop_mask) zz = zz * drop_mask[(size_t)b * L + j] * drop_scale;  // nolbo.py:1423-1425
        z[(size_t)b * L + j] = zz;
        if (z_act) vv_store(z_act, (size_t)b * L + j, zz);
        if (mean_out) mean_out[(size_t)b * L + j] = mu;
        if (logvar_out) logvar_out[(size_t)b * L + j] = lv;
        s += 0.5f * (0.f - lv) + (e + mu * mu) / 2.0f - 0.5f;  // function.py:96 with target N(0, I)
    }
    s = vv_wave_sum(s);
    if (kl && lane == 0) kl[b] = s;
}

// nolbo.py:1498-1501 batch means, summed in sample order by one wave (deterministic).
__global__ __launch_bounds__(64) void shape_metrics_kernel(const float *__restrict__ stats, float *__restrict__ out4, int batch) {
    const int lane = threadIdx.x;
    float bce = 0.f, pr = 0.f, rc = 0.f, iou = 0.f;
    for (int b = lane; b < batch; b += 64) {
        const float l = stats[b * 4 + 0], tp = stats[b * 4 + 1], fp = stats[b * 4 + 2], fn = stats[b * 4 + 3];
        bce += l;
        pr += tp / (tp + fp + 1e-10f);
        rc += tp / (tp + fn + 1e-10f);
        iou += tp / fmaxf(tp + fp + fn, 1.f);
    }
    bce = vv_wave_sum(bce); pr = vv_wave_sum(pr); rc = vv_wave_sum(rc); iou = vv_wave_sum(iou);
    if (lane == 0) {
        out4[0] = bce / batch; out4[1] = pr / batch; out4[2] = rc / batch; out4[3] = iou / batch;
    }
}

inline int grid_for(size_t total) {
    size_t g = (total + 255) / 256;
    return (int)(g < 1 ? 1 : (g > 8192 ? 8192 : g));
}

// Bit-packed occupancy storage: voxel v of a sample is bit (v & 7) of byte v >> 3 (numpy
// packbits, bitorder='little').  unpack: one thread per byte -> 8 floats (two 16-byte stores), rows gathered through idx.
__global__ void unpack_bits_gather_kernel(const unsigned char *__restrict__ packed, const int *__restrict__ idx, float *__restrict__ out,
                                          long bytes_per_sample, long total_bytes) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total_bytes; i += (long)gridDim.x * blockDim.x) {
        const long b = i / bytes_per_sample, k = i - b * bytes_per_sample;
        const long src = (idx ? (long)idx[b] : b) * bytes_per_sample + k;
        const unsigned v = packed[src];
        f32x4 lo, hi;
#pragma unroll
        for (int e = 0; e < 4; ++e) { lo[e] = (float)((v >> e) & 1u); hi[e] = (float)((v >> (4 + e)) & 1u); }
        f32x4 *o = reinterpret_cast<f32x4 *>(out + i * 8);
        o[0] = lo;
        o[1] = hi;
    }
}

__global__ void pack_bits_kernel(const float *__restrict__ x, unsigned char *__restrict__ packed, float threshold, long total_bytes) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total_bytes; i += (long)gridDim.x * blockDim.x) {
        const f32x4 lo = reinterpret_cast<const f32x4 *>(x + i * 8)[0], hi = reinterpret_cast<const f32x4 *>(x + i * 8)[1];
        unsigned v = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) v |= (lo[e] >= threshold ? 1u : 0u) << e | (hi[e] >= threshold ? 1u : 0u) << (4 + e);
        packed[i] = (unsigned char)v;
    }
}

template <typename TS, typename TD>
__global__ void convert_kernel(const TS *__restrict__ src, TD *__restrict__ dst, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) vv_store(dst, (size_t)i, vv_load_f32(src, (size_t)i));
}

// bf16 -> fp8, 8 elements per thread (n % 8 == 0): the hand-over from the bf16-only layers to the fp8 MFMA layers
__global__ void convert_bf16_fp8_kernel(const bf16x8 *__restrict__ src, u32x2 *__restrict__ dst, long n8) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
        const bf16x8 v = src[i];
        u32x2 o;
        o[0] = vv_pack_fp8x4(f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]});
        o[1] = vv_pack_fp8x4(f32x4{(float)v[4], (float)v[5], (float)v[6], (float)v[7]});
        dst[i] = o;
    }
}

}  // namespace

#define VV_PACK_DISPATCH(KERNEL, TOTAL, ...)                                                                   \
    do {                                                                                                       \
        if (!w_keras || !packed) return VV_ERR_NULL;                                                           \
        if (dtype != VV_F32 && dtype != VV_BF16 && dtype != VV_FP8) return VV_ERR_DTYPE;                       \
        hipStream_t st = reinterpret_cast<hipStream_t>(stream);                                                \
        if (dtype == VV_BF16)                                                                                  \
            VV_LAUNCH((KERNEL<__bf16>), dim3(grid_for(TOTAL)), dim3(256), 0, st, w_keras,             \
                               reinterpret_cast<__bf16 *>(packed), __VA_ARGS__);                               \
        else if (dtype == VV_FP8)                                                                              \
            VV_LAUNCH((KERNEL<vv_fp8>), dim3(grid_for(TOTAL)), dim3(256), 0, st, w_keras,             \
                               reinterpret_cast<vv_fp8 *>(packed), __VA_ARGS__);                               \
        else                                                                                                   \
            VV_LAUNCH((KERNEL<float>), dim3(grid_for(TOTAL)), dim3(256), 0, st, w_keras,              \
                               reinterpret_cast<float *>(packed), __VA_ARGS__);                                \
        return vv_launch_status();                                                                             \
    } while (0)

thread_local int vv_tls_last_hip_error = 0;

VV_EXPORT int vv_abi_version(void) { return 1; }

VV_EXPORT const char *vv_last_hip_error(void) { return hipGetErrorString((hipError_t)vv_tls_last_hip_error); }

VV_EXPORT const char *vv_status_string(int s) {
    switch (s) {
        case VV_OK: return "ok";
        case VV_ERR_NULL: return "required pointer is NULL";
        case VV_ERR_SHAPE: return "unsupported or inconsistent shape";
        case VV_ERR_DTYPE: return "unsupported dtype";
        case VV_ERR_ALIGN: return "pointer not 16-byte aligned";
        case VV_ERR_WORKSPACE: return "workspace missing or too small";
        case VV_ERR_LAUNCH: return "kernel launch failed";
        default: return "unknown status";
    }
}

VV_EXPORT int vv_pack_conv_k4(const float *w_keras, void *packed, int cin, int cout, int dtype, void *stream) {
    if (cin <= 0 || cout <= 0) return VV_ERR_SHAPE;
    if (w_keras && packed && cout % 64 == 0 && (dtype == VV_F32 || dtype == VV_BF16 || dtype == VV_FP8) && vv_aligned16(w_keras) && vv_aligned16(packed)) {
        hipStream_t st = reinterpret_cast<hipStream_t>(stream);
        const dim3 grid(cin, cout / 64);                 // K / 64 = cin
        if (dtype == VV_FP8) VV_LAUNCH(pack_conv_k4_tiled_kernel<vv_fp8>, grid, dim3(256), 0, st, w_keras, reinterpret_cast<vv_fp8 *>(packed), cin, cout);
        else if (dtype == VV_BF16) VV_LAUNCH(pack_conv_k4_tiled_kernel<__bf16>, grid, dim3(256), 0, st, w_keras, reinterpret_cast<__bf16 *>(packed), cin, cout);
        else VV_LAUNCH(pack_conv_k4_tiled_kernel<float>, grid, dim3(256), 0, st, w_keras, reinterpret_cast<float *>(packed), cin, cout);
        return vv_launch_status();
    }
    VV_PACK_DISPATCH(pack_conv_k4_kernel, (size_t)64 * cin * cout, cin, cout);
}

VV_EXPORT int vv_pack_convT_k4s2(const float *w_keras, void *packed, int cin, int cout, int dtype, void *stream) {
    if (cin <= 0 || cout <= 0) return VV_ERR_SHAPE;
    if (cin % 4 == 0 && vv_aligned16(w_keras) && vv_aligned16(packed))
        VV_PACK_DISPATCH(pack_convT_k4s2_vec_kernel, (size_t)16 * cin * cout, cin, cout);
    VV_PACK_DISPATCH(pack_convT_k4s2_kernel, (size_t)64 * cin * cout, cin, cout);
}

VV_EXPORT int vv_pack_conv_k4s1_meanpool(const float *w_keras, void *packed, int side, int cin, int cout, int dtype,
                                         void *stream) {
    if (cin <= 0 || cout <= 0 || side <= 0) return VV_ERR_SHAPE;
    if (w_keras && packed && cin % 64 == 0 && cout % 64 == 0 && (dtype == VV_F32 || dtype == VV_BF16) && vv_aligned16(w_keras)) {
        hipStream_t st = reinterpret_cast<hipStream_t>(stream);
        const dim3 grid(side * side * side * (cin / 64), cout / 64);
        if (dtype == VV_BF16) VV_LAUNCH(pack_conv_k4s1_meanpool_tiled_kernel<__bf16>, grid, dim3(256), 0, st, w_keras, reinterpret_cast<__bf16 *>(packed), side, cin, cout);
        else VV_LAUNCH(pack_conv_k4s1_meanpool_tiled_kernel<float>, grid, dim3(256), 0, st, w_keras, reinterpret_cast<float *>(packed), side, cin, cout);
        return vv_launch_status();
    }
    VV_PACK_DISPATCH(pack_conv_k4s1_meanpool_kernel, (size_t)side * side * side * cin * cout, side, cin, cout);
}

VV_EXPORT int vv_pack_conv_k4s1_full(const float *w_keras, void *packed, int side, int cin, int cout, int dtype, void *stream) {
    if (cin <= 0 || cout <= 0 || side <= 0) return VV_ERR_SHAPE;
    const size_t s3 = (size_t)side * side * side;
    VV_PACK_DISPATCH(pack_conv_k4s1_full_kernel, s3 * cout * s3 * cin, side, cin, cout);
}

VV_EXPORT int vv_sigmoid_f32(const float *x, float *y, long n, void *stream) {
    if (!x || !y) return VV_ERR_NULL;
    if (n <= 0) return VV_ERR_SHAPE;
    VV_LAUNCH(sigmoid_f32_kernel, dim3(grid_for((size_t)n)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x, y, n);
    return vv_launch_status();
}

VV_EXPORT int vv_max_over_positions(const float *x, float *out, int batch, int npos, int channels, void *stream) {
    if (!x || !out) return VV_ERR_NULL;
    if (batch <= 0 || npos <= 0 || channels <= 0) return VV_ERR_SHAPE;
    VV_LAUNCH(max_over_positions_kernel, dim3(grid_for((size_t)batch * channels)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x, out,
              batch, npos, channels);
    return vv_launch_status();
}

VV_EXPORT int vv_pack_convT_k4s1_dense(const float *w_keras, void *packed, int side, int cin, int cout, int dtype,
                                       void *stream) {
    if (cin <= 0 || cout <= 0 || side <= 0) return VV_ERR_SHAPE;
    const size_t s3 = (size_t)side * side * side;
    VV_PACK_DISPATCH(pack_convT_k4s1_dense_kernel, s3 * cout * s3 * cin, side, cin, cout);
}

VV_EXPORT int vv_pack_dense(const float *w_keras, void *packed, int in, int out, int dtype, void *stream) {
    if (in <= 0 || out <= 0) return VV_ERR_SHAPE;
    VV_PACK_DISPATCH(pack_dense_kernel, (size_t)in * out, in, out);
}

VV_EXPORT int vv_fold_bn(const float *gamma, const float *beta, const float *mean, const float *var, const float *bias,
                         float eps, float *scale, float *shift, int channels, int repeat, void *stream) {
    if (!gamma || !beta || !mean || !var || !scale || !shift) return VV_ERR_NULL;
    if (channels <= 0 || repeat <= 0) return VV_ERR_SHAPE;
    VV_LAUNCH(fold_bn_kernel, dim3(grid_for((size_t)channels * repeat)), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), gamma, beta, mean, var, bias, eps, scale, shift, channels,
                       repeat);
    return vv_launch_status();
}

VV_EXPORT int vv_reparam_kl_fwd(const float *enc_out, const float *eps, const float *drop_mask, float drop_scale, float *z,
                                void *z_act, int act_dtype, float *kl, float *mean, float *logvar, int batch, int latent,
                                void *stream) {
    if (!enc_out || !eps || !z) return VV_ERR_NULL;
    if (batch <= 0 || latent <= 0) return VV_ERR_SHAPE;
    if (z_act && act_dtype != VV_F32 && act_dtype != VV_BF16) return VV_ERR_DTYPE;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const dim3 grid((batch + 3) / 4), block(256);
    if (z_act && act_dtype == VV_BF16)
        VV_LAUNCH((reparam_kl_kernel<__bf16>), grid, block, 0, st, enc_out, eps, drop_mask, drop_scale, z,
                           reinterpret_cast<__bf16 *>(z_act), kl, mean, logvar, batch, latent);
    else
        VV_LAUNCH((reparam_kl_kernel<float>), grid, block, 0, st, enc_out, eps, drop_mask, drop_scale, z,
                           reinterpret_cast<float *>(z_act), kl, mean, logvar, batch, latent);
    return vv_launch_status();
}

VV_EXPORT int vv_shape_metrics(const float *stats, float *out4, int batch, void *stream) {
    if (!stats || !out4) return VV_ERR_NULL;
    if (batch <= 0) return VV_ERR_SHAPE;
    VV_LAUNCH(shape_metrics_kernel, dim3(1), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), stats, out4, batch);
    return vv_launch_status();
}

VV_EXPORT int vv_unpack_bits_gather(const void *packed, const int *index, float *out, int batch, long voxels, void *stream) {
    if (!packed || !out) return VV_ERR_NULL;
    if (batch <= 0 || voxels <= 0 || voxels % 8) return VV_ERR_SHAPE;
    if (!vv_aligned16(out)) return VV_ERR_ALIGN;
    const long bps = voxels / 8, total = bps * batch;
    VV_LAUNCH(unpack_bits_gather_kernel, dim3(grid_for((size_t)total)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
              reinterpret_cast<const unsigned char *>(packed), index, out, bps, total);
    return vv_launch_status();
}

VV_EXPORT int vv_pack_bits(const float *x, void *packed, float threshold, long n, void *stream) {
    if (!x || !packed) return VV_ERR_NULL;
    if (n <= 0 || n % 8) return VV_ERR_SHAPE;
    if (!vv_aligned16(x)) return VV_ERR_ALIGN;
    VV_LAUNCH(pack_bits_kernel, dim3(grid_for((size_t)(n / 8))), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x,
              reinterpret_cast<unsigned char *>(packed), threshold, n / 8);
    return vv_launch_status();
}

VV_EXPORT int vv_convert(const void *src, void *dst, long n, int src_dtype, int dst_dtype, void *stream) {
    if (!src || !dst) return VV_ERR_NULL;
    if (src_dtype != VV_F32 && src_dtype != VV_BF16 && src_dtype != VV_FP8) return VV_ERR_DTYPE;
    if (dst_dtype != VV_F32 && dst_dtype != VV_BF16 && dst_dtype != VV_FP8) return VV_ERR_DTYPE;
    if (n <= 0) return VV_ERR_SHAPE;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const dim3 g(grid_for((size_t)n)), b(256);
    if (src_dtype == VV_BF16 && dst_dtype == VV_FP8 && n % 8 == 0 && vv_aligned16(src) && vv_aligned16(dst))
        VV_LAUNCH(convert_bf16_fp8_kernel, dim3(grid_for((size_t)(n / 8))), b, 0, st, reinterpret_cast<const bf16x8 *>(src), reinterpret_cast<u32x2 *>(dst), n / 8);
    else if (dst_dtype == VV_FP8 && src_dtype == VV_F32)
        VV_LAUNCH((convert_kernel<float, vv_fp8>), g, b, 0, st, reinterpret_cast<const float *>(src), reinterpret_cast<vv_fp8 *>(dst), n);
    else if (dst_dtype == VV_FP8 && src_dtype == VV_BF16)
        VV_LAUNCH((convert_kernel<__bf16, vv_fp8>), g, b, 0, st, reinterpret_cast<const __bf16 *>(src), reinterpret_cast<vv_fp8 *>(dst), n);
    else if (src_dtype == VV_FP8 && dst_dtype == VV_F32)
        VV_LAUNCH((convert_kernel<vv_fp8, float>), g, b, 0, st, reinterpret_cast<const vv_fp8 *>(src), reinterpret_cast<float *>(dst), n);
    else if (src_dtype == VV_FP8 && dst_dtype == VV_BF16)
        VV_LAUNCH((convert_kernel<vv_fp8, __bf16>), g, b, 0, st, reinterpret_cast<const vv_fp8 *>(src), reinterpret_cast<__bf16 *>(dst), n);
    else if (src_dtype == VV_FP8)
        return VV_ERR_DTYPE;
    else if (src_dtype == VV_F32 && dst_dtype == VV_BF16)
        VV_LAUNCH((convert_kernel<float, __bf16>), g, b, 0, st, reinterpret_cast<const float *>(src), reinterpret_cast<__bf16 *>(dst), n);
    else if (src_dtype == VV_BF16 && dst_dtype == VV_F32)
        VV_LAUNCH((convert_kernel<__bf16, float>), g, b, 0, st, reinterpret_cast<const __bf16 *>(src), reinterpret_cast<float *>(dst), n);
    else if (src_dtype == VV_F32)
        VV_LAUNCH((convert_kernel<float, float>), g, b, 0, st, reinterpret_cast<const float *>(src), reinterpret_cast<float *>(dst), n);
    else
        VV_LAUNCH((convert_kernel<__bf16, __bf16>), g, b, 0, st, reinterpret_cast<const __bf16 *>(src), reinterpret_cast<__bf16 *>(dst), n);
    return vv_launch_status();
}

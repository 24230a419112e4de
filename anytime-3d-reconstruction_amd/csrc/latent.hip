// Latent-space ops of the missing-modality evaluation (nolbo.py:1472-1518) and the stand-alone loss ops of
// src/module/function.py.  Tiny, latency-bound kernels: one wave (or one workgroup) per sample, reductions by
// wave shuffles.  gfx950 only.
#include "common.h"

namespace {

// z = z*mask; z = where(z == 0, mean_c(P), z)   -- nolbo.py:1474-1482 (also rewrites genuine zeros, as the
// reference does).
template <typename TA>
__global__ void mask_fill_kernel(const float *__restrict__ z, const float *__restrict__ mask, const float *__restrict__ protos,
                                 int C, float *__restrict__ out, TA *__restrict__ out_act, int B, int L) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * L) return;
    const int j = i % L;
    float m = 0.f;
    for (int c = 0; c < C; ++c) m += protos[c * L + j];
    m /= (float)C;
    float v = z[i] * mask[i];
    v = (v == 0.f) ? m : v;
    out[i] = v;
    if (out_act) vv_store(out_act, (size_t)i, v);
}

// argmin_c sum_j mask_j * (z_j - P_cj)^2, first minimum (tf.argmin)   -- nolbo.py:1489-1493, 1505-1506
__global__ __launch_bounds__(64) void nearest_category_kernel(const float *__restrict__ z, const float *__restrict__ mask,
                                                              const float *__restrict__ protos, int C, int *__restrict__ idx,
                                                              int L) {
    const int b = blockIdx.x, lane = threadIdx.x;
    float best = INFINITY;
    int besti = 0x7fffffff;
    for (int c = lane; c < C; c += 64) {
        float d = 0.f;
        for (int j = 0; j < L; ++j) {
            const float t = z[(size_t)b * L + j] - protos[(size_t)c * L + j];
            d += (mask ? mask[(size_t)b * L + j] : 1.f) * (t * t);
        }
        if (d < best) { best = d; besti = c; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ob = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(besti, o, 64);
        if (ob < best || (ob == best && oi < besti)) { best = ob; besti = oi; }
    }
    if (lane == 0) idx[b] = besti;
}

// z_prior = P[idx] + eps2 (sampling with logVar = 0); z_corr = where(mask == 0, z_prior, z)   -- nolbo.py:1507-1510
template <typename TA>
__global__ void latent_correct_kernel(const float *__restrict__ z, const float *__restrict__ mask, const float *__restrict__ protos,
                                      const int *__restrict__ idx, const float *__restrict__ eps2, float *__restrict__ out,
                                      TA *__restrict__ out_act, int B, int L) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * L) return;
    const int b = i / L, j = i % L;
    const float prior = protos[(size_t)idx[b] * L + j] + sqrtf(expf(0.f)) * eps2[i];
    const float v = (mask[i] == 0.f) ? prior : z[i];
    out[i] = v;
    if (out_act) vv_store(out_act, (size_t)i, v);
}

// mean_b [idx_b == argmax_c onehot_bc]   -- nolbo.py:1493-1494
__global__ __launch_bounds__(64) void category_accuracy_kernel(const int *__restrict__ idx, const float *__restrict__ onehot,
                                                               int C, float *__restrict__ acc, int B) {
    const int lane = threadIdx.x;
    float hit = 0.f;
    for (int b = lane; b < B; b += 64) {
        int am = 0;
        float mv = onehot[(size_t)b * C];
        for (int c = 1; c < C; ++c) {
            const float v = onehot[(size_t)b * C + c];
            if (v > mv) { mv = v; am = c; }
        }
        hit += (idx[b] == am) ? 1.f : 0.f;
    }
    hit = vv_wave_sum(hit);
    if (lane == 0) acc[0] = hit / (float)B;
}

__device__ __forceinline__ float block_sum_256(float v, float *red) {
    v = vv_wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

// binary_loss on PROBABILITIES   -- function.py:73-82
__global__ __launch_bounds__(256) void binary_loss_kernel(const float *__restrict__ pred, const float *__restrict__ target,
                                                          float epsilon, float gamma, float b_range, float *__restrict__ out, long V) {
    __shared__ float red[4];
    const int b = blockIdx.x;
    const float hi = 1.0f - epsilon;
    float s = 0.f;
    for (long v = threadIdx.x; v < V; v += 256) {
        const float yt = -b_range + (2.0f * b_range + 1.0f) * target[(size_t)b * V + v];
        const float yp = fminf(fmaxf(pred[(size_t)b * V + v], epsilon), hi);
        s += gamma * yt * logf(yp) + (1.0f - gamma) * (1.0f - yt) * logf(1.0f - yp);
    }
    s = block_sum_256(s, red);
    if (threadIdx.x == 0) out[b] = -s;
}

// voxelPrecisionRecall   -- function.py:100-115
__global__ __launch_bounds__(256) void precision_recall_kernel(const float *__restrict__ target, const float *__restrict__ pred,
                                                               float prob, float *__restrict__ tp, float *__restrict__ fp,
                                                               float *__restrict__ fn, long V) {
    __shared__ float red[4];
    const int b = blockIdx.x;
    float a = 0.f, c = 0.f, d = 0.f;
    for (long v = threadIdx.x; v < V; v += 256) {
        const float y = target[(size_t)b * V + v];
        const float yh = pred[(size_t)b * V + v] >= prob ? 1.f : 0.f;
        a += y * yh; c += (1.f - y) * yh; d += y * (1.f - yh);
    }
    a = block_sum_256(a, red);
    c = block_sum_256(c, red);
    d = block_sum_256(d, red);
    if (threadIdx.x == 0) { tp[b] = a; fp[b] = c; fn[b] = d; }
}

// kl_loss   -- function.py:84-98
__global__ __launch_bounds__(64) void kl_loss_kernel(const float *__restrict__ mean, const float *__restrict__ logvar,
                                                     const float *__restrict__ mean_t, const float *__restrict__ logvar_t,
                                                     float *__restrict__ out, int L) {
    const int b = blockIdx.x, lane = threadIdx.x;
    float s = 0.f;
    for (int j = lane; j < L; j += 64) {
        const size_t i = (size_t)b * L + j;
        const float d = mean[i] - mean_t[i];
        s += 0.5f * (logvar_t[i] - logvar[i]) + (expf(logvar[i]) + d * d) / (2.0f * expf(logvar_t[i])) - 0.5f;
    }
    s = vv_wave_sum(s);
    if (lane == 0) out[b] = s;
}

// sampling   -- function.py:35-38 (epsilon injected)
__global__ void sampling_kernel(const float *__restrict__ mu, const float *__restrict__ logvar, const float *__restrict__ eps,
                                float *__restrict__ out, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = mu[i] + sqrtf(expf(logvar[i])) * eps[i];
}

// regulizer_loss (function.py:40-71): out[i] = sum_j same(i,j) * hinge(sum_l |m_i - m_j| / exp(0.5 lv_i) - dist)^2 with
// hinge(d) = d > 0 ? 0 : d; same(i,j) = 1 when the class rows are identical (or no class input).  One workgroup per i.
__global__ __launch_bounds__(256) void regulizer_loss_kernel(const float *__restrict__ mean, const float *__restrict__ logvar,
                                                             const float *__restrict__ cls, float dist, float *__restrict__ out,
                                                             int batch, int latent, int cdim) {
    __shared__ float red[4];
    const int i = blockIdx.x, tid = threadIdx.x;
    float acc = 0.f;
    for (int j = tid; j < batch; j += 256) {
        float d = 0.f;
        for (int l = 0; l < latent; ++l)
            d += fabsf(mean[(size_t)i * latent + l] - mean[(size_t)j * latent + l]) / expf(0.5f * logvar[(size_t)i * latent + l]);
        d -= dist;
        float v = d > 0.f ? 0.f : d * d;
        if (cls) {
            float cd = 0.f;
            for (int c = 0; c < cdim; ++c) cd += fabsf(cls[(size_t)i * cdim + c] - cls[(size_t)j * cdim + c]);
            if (cd > 0.f) v = 0.f;
        }
        acc += v;
    }
    acc = vv_wave_sum(acc);
    if ((tid & 63) == 0) red[tid >> 6] = acc;
    __syncthreads();
    if (tid == 0) out[i] = red[0] + red[1] + red[2] + red[3];
}

}  // namespace

VV_EXPORT int vv_latent_mask_fill(const float *z, const float *mask, const float *prototypes, int classes, float *z_out,
                                  void *z_act, int act_dtype, int batch, int latent, void *stream) {
    if (!z || !mask || !prototypes || !z_out) return VV_ERR_NULL;
    if (batch <= 0 || latent <= 0 || classes <= 0) return VV_ERR_SHAPE;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const dim3 grid((batch * latent + 255) / 256), block(256);
    if (z_act && act_dtype == VV_BF16)
        VV_LAUNCH((mask_fill_kernel<__bf16>), grid, block, 0, st, z, mask, prototypes, classes, z_out,
                           reinterpret_cast<__bf16 *>(z_act), batch, latent);
    else
        VV_LAUNCH((mask_fill_kernel<float>), grid, block, 0, st, z, mask, prototypes, classes, z_out,
                           reinterpret_cast<float *>(z_act), batch, latent);
    return vv_launch_status();
}

VV_EXPORT int vv_nearest_category(const float *z, const float *mask, const float *prototypes, int classes, int *argmin,
                                  int batch, int latent, void *stream) {
    if (!z || !prototypes || !argmin) return VV_ERR_NULL;
    if (batch <= 0 || latent <= 0 || classes <= 0) return VV_ERR_SHAPE;
    VV_LAUNCH(nearest_category_kernel, dim3(batch), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), z, mask,
                       prototypes, classes, argmin, latent);
    return vv_launch_status();
}

VV_EXPORT int vv_latent_correct(const float *z, const float *mask, const float *prototypes, const int *argmin, const float *eps2,
                                float *z_corr, void *z_act, int act_dtype, int batch, int latent, void *stream) {
    if (!z || !mask || !prototypes || !argmin || !eps2 || !z_corr) return VV_ERR_NULL;
    if (batch <= 0 || latent <= 0) return VV_ERR_SHAPE;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const dim3 grid((batch * latent + 255) / 256), block(256);
    if (z_act && act_dtype == VV_BF16)
        VV_LAUNCH((latent_correct_kernel<__bf16>), grid, block, 0, st, z, mask, prototypes, argmin, eps2, z_corr,
                           reinterpret_cast<__bf16 *>(z_act), batch, latent);
    else
        VV_LAUNCH((latent_correct_kernel<float>), grid, block, 0, st, z, mask, prototypes, argmin, eps2, z_corr,
                           reinterpret_cast<float *>(z_act), batch, latent);
    return vv_launch_status();
}

VV_EXPORT int vv_category_accuracy(const int *argmin, const float *onehot, int classes, float *acc, int batch, void *stream) {
    if (!argmin || !onehot || !acc) return VV_ERR_NULL;
    if (batch <= 0 || classes <= 0) return VV_ERR_SHAPE;
    VV_LAUNCH(category_accuracy_kernel, dim3(1), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), argmin, onehot,
                       classes, acc, batch);
    return vv_launch_status();
}

VV_EXPORT int vv_binary_loss(const float *pred, const float *target, float epsilon, float gamma, float b_range, float *out,
                             int batch, long voxels, void *stream) {
    if (!pred || !target || !out) return VV_ERR_NULL;
    if (batch <= 0 || voxels <= 0) return VV_ERR_SHAPE;
    VV_LAUNCH(binary_loss_kernel, dim3(batch), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), pred, target,
                       epsilon, gamma, b_range, out, voxels);
    return vv_launch_status();
}

VV_EXPORT int vv_voxel_precision_recall(const float *target, const float *pred, float prob, float *tp, float *fp, float *fn,
                                        int batch, long voxels, void *stream) {
    if (!pred || !target || !tp || !fp || !fn) return VV_ERR_NULL;
    if (batch <= 0 || voxels <= 0) return VV_ERR_SHAPE;
    VV_LAUNCH(precision_recall_kernel, dim3(batch), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), target, pred,
                       prob, tp, fp, fn, voxels);
    return vv_launch_status();
}

VV_EXPORT int vv_regulizer_loss(const float *mean, const float *logvar, const float *class_input, float dist_in_z_space,
                                float *out, int batch, int latent, int class_dim, void *stream) {
    if (!mean || !logvar || !out) return VV_ERR_NULL;
    if (batch <= 0 || latent <= 0 || (class_input && class_dim <= 0)) return VV_ERR_SHAPE;
    VV_LAUNCH(regulizer_loss_kernel, dim3(batch), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), mean, logvar, class_input,
              dist_in_z_space, out, batch, latent, class_dim);
    return vv_launch_status();
}

VV_EXPORT int vv_kl_loss(const float *mean, const float *logvar, const float *mean_target, const float *logvar_target,
                         float *out, int batch, int latent, void *stream) {
    if (!mean || !logvar || !mean_target || !logvar_target || !out) return VV_ERR_NULL;
    if (batch <= 0 || latent <= 0) return VV_ERR_SHAPE;
    VV_LAUNCH(kl_loss_kernel, dim3(batch), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), mean, logvar,
                       mean_target, logvar_target, out, latent);
    return vv_launch_status();
}

VV_EXPORT int vv_sampling(const float *mu, const float *logvar, const float *eps, float *out, long n, void *stream) {
    if (!mu || !logvar || !eps || !out) return VV_ERR_NULL;
    if (n <= 0) return VV_ERR_SHAPE;
    VV_LAUNCH(sampling_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       mu, logvar, eps, out, n);
    return vv_launch_status();
}

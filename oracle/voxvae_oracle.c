/* TEST INFRASTRUCTURE ONLY -- fp32 CPU restatement of the voxel VAE hot path (plain C + OpenMP).
 *
 * PARITY UNPINNED: the reference is Python on TensorFlow; TensorFlow cannot be installed here and the
 * reference holds no golden vectors.  This file restates, in the reference's own
 * arithmetic type (float32) and with direct gather formulas, the ops that oracle/numpy_oracle.py states
 * at definition level; the two and a torch-CPU statement must agree (tests/test_oracle_*.py).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 * It is also the "port" CPU baseline of bench.py (timed on the GPU box's host cores).
 *
 * Reference citations are relative to /root/reference.  Layout: channels-last (NDHWC), cubic grids.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <omp.h>

#define VVO_API __attribute__((visibility("default")))

/* TensorFlow 'SAME' padding rule (not in the tree; TF documented semantics). */
static void same_pad(int n, int k, int s, int *out, int *before) {
    int o = (n + s - 1) / s;
    int total = (o - 1) * s + k - n;
    if (total < 0) total = 0;
    *out = o;
    *before = total / 2;
}

VVO_API int vvo_num_threads(void) { return omp_get_max_threads(); }

/* Conv3D(padding='same', use_bias=False): src/net_core/autoencoder3D.py:27-30, 86-88.
 * x [B,D,D,D,Ci]; w [k,k,k,Ci,Co] (Keras kernel layout); y [B,O,O,O,Co], O = ceil(D/s).
 * y[o,co] = sum_{t,ci} x[s*o - pb + t, ci] * w[t,ci,co]   (taps that fall in the padding skipped).
 * One (b, od, oh) row of outputs is accumulated at a time so each weight row w[t,ci,:] is reused across the row. */
VVO_API void vvo_conv3d_same(const float *x, const float *w, float *y, int B, int D, int Ci, int Co, int k, int s) {
    int O, pb;
    same_pad(D, k, s, &O, &pb);
#pragma omp parallel for collapse(3) schedule(static)
    for (int b = 0; b < B; ++b)
        for (int od = 0; od < O; ++od)
            for (int oh = 0; oh < O; ++oh) {
                float *acc = y + (((size_t)b * O + od) * O + oh) * O * Co; /* [O][Co] row block of the output */
                for (size_t i = 0; i < (size_t)O * Co; ++i) acc[i] = 0.f;
                for (int td = 0; td < k; ++td) {
                    int id = s * od - pb + td;
                    if (id < 0 || id >= D) continue;
                    for (int th = 0; th < k; ++th) {
                        int ih = s * oh - pb + th;
                        if (ih < 0 || ih >= D) continue;
                        for (int tw = 0; tw < k; ++tw) {
                            const float *wp = w + (size_t)((td * k + th) * k + tw) * Ci * Co;
                            for (int ci = 0; ci < Ci; ++ci) {
                                const float *wr = wp + (size_t)ci * Co;
                                for (int ow = 0; ow < O; ++ow) {
                                    int iw = s * ow - pb + tw;
                                    if (iw < 0 || iw >= D) continue;
                                    float a = x[((((size_t)b * D + id) * D + ih) * D + iw) * Ci + ci];
                                    float *ar = acc + (size_t)ow * Co;
                                    for (int c = 0; c < Co; ++c) ar[c] += a * wr[c];
                                }
                            }
                        }
                    }
                }
            }
}

/* Conv3DTranspose(padding='same', use_bias=False): autoencoder3D.py:42-45, 129-132.
 * x [B,D,D,D,Ci]; w [k,k,k,Co,Ci] (Keras transposed-kernel layout); y [B,sD,sD,sD,Co].
 * Gradient of the SAME conv of size sD w.r.t. its input:
 * y[o,co] = sum_{i,t : t = o + pb - s*i, 0<=t<k} sum_ci x[i,ci] * w[t,co,ci]. */
VVO_API void vvo_conv3d_transpose_same(const float *x, const float *w, float *y, int B, int D, int Ci, int Co, int k, int s) {
    int N = D * s, O, pb;
    same_pad(N, k, s, &O, &pb); /* O == D */
    /* [t][co][ci] -> [t][ci][co] so the inner loop is a contiguous AXPY (Co > 1) */
    float *wt = (float *)malloc(sizeof(float) * (size_t)k * k * k * Ci * Co);
    for (int t = 0; t < k * k * k; ++t)
        for (int co = 0; co < Co; ++co)
            for (int ci = 0; ci < Ci; ++ci) wt[((size_t)t * Ci + ci) * Co + co] = w[((size_t)t * Co + co) * Ci + ci];
#pragma omp parallel for collapse(3) schedule(static)
    for (int b = 0; b < B; ++b)
        for (int od = 0; od < N; ++od)
            for (int oh = 0; oh < N; ++oh) {
                float *acc = y + (((size_t)b * N + od) * N + oh) * N * Co; /* [N][Co] row block */
                for (size_t i = 0; i < (size_t)N * Co; ++i) acc[i] = 0.f;
                for (int td = 0; td < k; ++td) {
                    int nd = od + pb - td;
                    if (nd < 0 || nd % s) continue;
                    int id = nd / s;
                    if (id >= D) continue;
                    for (int th = 0; th < k; ++th) {
                        int nh = oh + pb - th;
                        if (nh < 0 || nh % s) continue;
                        int ih = nh / s;
                        if (ih >= D) continue;
                        for (int tw = 0; tw < k; ++tw) {
                            int t = (td * k + th) * k + tw;
                            if (Co == 1) {
                                const float *wr = w + (size_t)t * Ci;
                                for (int ow = 0; ow < N; ++ow) {
                                    int nw = ow + pb - tw;
                                    if (nw < 0 || nw % s) continue;
                                    int iw = nw / s;
                                    if (iw >= D) continue;
                                    const float *xp = x + ((((size_t)b * D + id) * D + ih) * D + iw) * Ci;
                                    float d = 0.f;
                                    for (int ci = 0; ci < Ci; ++ci) d += xp[ci] * wr[ci];
                                    acc[ow] += d;
                                }
                            } else {
                                const float *wp = wt + (size_t)t * Ci * Co;
                                for (int ci = 0; ci < Ci; ++ci) {
                                    const float *wr = wp + (size_t)ci * Co;
                                    for (int ow = 0; ow < N; ++ow) {
                                        int nw = ow + pb - tw;
                                        if (nw < 0 || nw % s) continue;
                                        int iw = nw / s;
                                        if (iw >= D) continue;
                                        float a = x[((((size_t)b * D + id) * D + ih) * D + iw) * Ci + ci];
                                        float *ar = acc + (size_t)ow * Co;
                                        for (int c = 0; c < Co; ++c) ar[c] += a * wr[c];
                                    }
                                }
                            }
                        }
                    }
                }
            }
    free(wt);
}

/* act: 0 none, 1 elu (alpha 1), 2 relu, 3 leaky relu (alpha 0.3) -- autoencoder3D.py:33-38 */
static inline float act_f(float v, int act) {
    switch (act) {
        case 1: return v > 0.f ? v : expm1f(v);
        case 2: return v > 0.f ? v : 0.f;
        case 3: return v > 0.f ? v : 0.3f * v;
        default: return v;
    }
}

/* BatchNormalization(training=False) + activation in place, rows x C: autoencoder3D.py:31-38.
 * gamma*(x-mean)/sqrt(var+eps)+beta, eps = 1e-3 (Keras default). */
VVO_API void vvo_bn_act(float *x, long rows, int C, const float *gamma, const float *beta, const float *mean,
                        const float *var, float eps, int act) {
#pragma omp parallel for schedule(static)
    for (long r = 0; r < rows; ++r) {
        float *p = x + (size_t)r * C;
        for (int c = 0; c < C; ++c) {
            float v = gamma[c] * (p[c] - mean[c]) / sqrtf(var[c] + eps) + beta[c];
            p[c] = act_f(v, act);
        }
    }
}

/* Dense(use_bias=True): autoencoder3D.py:59-61.  y[B,Out] = x[B,In] @ W[In,Out] + b */
VVO_API void vvo_dense(const float *x, const float *W, const float *bias, float *y, int B, int In, int Out) {
#pragma omp parallel for schedule(static)
    for (int b = 0; b < B; ++b) {
        float *yp = y + (size_t)b * Out;
        for (int o = 0; o < Out; ++o) yp[o] = 0.f;
        for (int i = 0; i < In; ++i) {
            float a = x[(size_t)b * In + i];
            const float *wr = W + (size_t)i * Out;
            for (int o = 0; o < Out; ++o) yp[o] += a * wr[o];
        }
        if (bias)
            for (int o = 0; o < Out; ++o) yp[o] += bias[o];
    }
}

/* tf.reduce_mean(x, axis=[1,2,3]): autoencoder3D.py:90-91.  x [B,S,C] -> y [B,C] */
VVO_API void vvo_mean_pool(const float *x, float *y, int B, int S, int C) {
    for (int b = 0; b < B; ++b)
        for (int c = 0; c < C; ++c) {
            float a = 0.f;
            for (int s = 0; s < S; ++s) a += x[((size_t)b * S + s) * C + c];
            y[(size_t)b * C + c] = a / (float)S;
        }
}

/* slice + clip + sampling + kl_loss vs N(0,I): src/module/nolbo.py:1464-1470, src/module/function.py:35-38, 84-98.
 * enc_out [B,2L]; eps [B,L]; z [B,L]; kl [B]. */
VVO_API void vvo_reparam_kl(const float *enc_out, const float *eps, float *z, float *kl, int B, int L) {
    for (int b = 0; b < B; ++b) {
        float s = 0.f;
        for (int j = 0; j < L; ++j) {
            float mu = enc_out[(size_t)b * 2 * L + j];
            float lv = enc_out[(size_t)b * 2 * L + L + j];
            lv = lv < -10.f ? -10.f : (lv > 10.f ? 10.f : lv);
            float e = expf(lv);
            z[(size_t)b * L + j] = mu + sqrtf(e) * eps[(size_t)b * L + j];
            s += 0.5f * (0.f - lv) + (e + (mu - 0.f) * (mu - 0.f)) / (2.0f * expf(0.f)) - 0.5f;
        }
        kl[b] = s;
    }
}

/* tf.sigmoid (autoencoder3D.py:136) + binary_loss gamma (function.py:73-82, called with 0.6 at nolbo.py:1497)
 * + voxelPrecisionRecall (function.py:100-115).  logits,target [B,V] -> probs [B,V], bce/tp/fp/fn [B]. */
VVO_API void vvo_sigmoid_bce_counts(const float *logits, const float *target, float *probs, float *bce, float *tp,
                                    float *fp, float *fn, int B, long V, float gamma, float epsilon) {
#pragma omp parallel for schedule(static)
    for (int b = 0; b < B; ++b) {
        double s = 0.0;
        double ntp = 0, nfp = 0, nfn = 0; /* sums of products, as function.py:111-113 (exact for y in {0,1}) */
        const float hi = 1.0f - epsilon; /* 0.99999988 in float32 */
        for (long v = 0; v < V; ++v) {
            float l = logits[(size_t)b * V + v], y = target[(size_t)b * V + v];
            float p = 1.0f / (1.0f + expf(-l));
            probs[(size_t)b * V + v] = p;
            float q = p < epsilon ? epsilon : (p > hi ? hi : p);
            s += -(double)(gamma * y * logf(q) + (1.0f - gamma) * (1.0f - y) * logf(1.0f - q));
            float yh = p >= 0.5f ? 1.f : 0.f;
            ntp += y * yh;
            nfp += (1.f - y) * yh;
            nfn += y * (1.f - yh);
        }
        bce[b] = (float)s;
        tp[b] = (float)ntp;
        fp[b] = (float)nfp;
        fn[b] = (float)nfn;
    }
}

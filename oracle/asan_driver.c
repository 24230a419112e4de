/* TEST INFRASTRUCTURE ONLY: AddressSanitizer / UBSan run of the C restatement (SURVEY.md §5 "sanitizers": GPU ASan is not
 * available on this pool, so the sanitizers run on the CPU build).  Every exported routine of voxvae_oracle.c is driven once on
 * small, odd-sized shapes with exactly-sized heap buffers, so any out-of-bounds index in the restatement trips ASan.
 * Build + run: make -C oracle asan (tests/test_oracle.py::test_c_oracle_under_asan does that). */
#include <stdio.h>
#include <stdlib.h>

#include "voxvae_oracle.c"

static float *buf(size_t n, unsigned seed) {
    float *p = (float *)malloc(n * sizeof(float));
    for (size_t i = 0; i < n; ++i) {
        seed = seed * 1664525u + 1013904223u;
        p[i] = ((seed >> 8) & 0xFFFF) / 65536.0f - 0.5f;
    }
    return p;
}

int main(void) {
    const int B = 3;
    /* Conv3D k4 s2 and s1 (SAME pads 1/1 and 1/2), odd channel counts */
    {
        const int D = 6, Ci = 3, Co = 5;
        float *x = buf((size_t)B * D * D * D * Ci, 1), *w = buf((size_t)64 * Ci * Co, 2);
        float *y2 = buf((size_t)B * 27 * Co, 3), *y1 = buf((size_t)B * D * D * D * Co, 4);
        vvo_conv3d_same(x, w, y2, B, D, Ci, Co, 4, 2);
        vvo_conv3d_same(x, w, y1, B, D, Ci, Co, 4, 1);
        free(x); free(w); free(y2); free(y1);
    }
    /* Conv3DTranspose k4 s2 / s1, Co = 1 (the logit layer) and Co > 1 */
    {
        const int D = 3, Ci = 5;
        for (int Co = 1; Co <= 4; Co += 3) {
            float *x = buf((size_t)B * D * D * D * Ci, 5), *w = buf((size_t)64 * Co * Ci, 6);
            float *y2 = buf((size_t)B * 216 * Co, 7), *y1 = buf((size_t)B * 27 * Co, 8);
            vvo_conv3d_transpose_same(x, w, y2, B, D, Ci, Co, 4, 2);
            vvo_conv3d_transpose_same(x, w, y1, B, D, Ci, Co, 4, 1);
            free(x); free(w); free(y2); free(y1);
        }
    }
    {
        const int C = 7; const long rows = 11;
        float *x = buf((size_t)rows * C, 9), *g = buf(C, 10), *b = buf(C, 11), *m = buf(C, 12), *v = buf(C, 13);
        for (int i = 0; i < C; ++i) v[i] = v[i] * v[i] + 0.5f;
        for (int act = 0; act < 4; ++act) vvo_bn_act(x, rows, C, g, b, m, v, 1e-3f, act);
        free(x); free(g); free(b); free(m); free(v);
    }
    {
        const int In = 9, Out = 6;
        float *x = buf((size_t)B * In, 14), *W = buf((size_t)In * Out, 15), *bias = buf(Out, 16), *y = buf((size_t)B * Out, 17);
        vvo_dense(x, W, bias, y, B, In, Out);
        vvo_dense(x, W, NULL, y, B, In, Out);
        free(x); free(W); free(bias); free(y);
    }
    {
        const int S = 3, C = 4;
        float *x = buf((size_t)B * S * S * S * C, 18), *y = buf((size_t)B * C, 19);
        vvo_mean_pool(x, y, B, S, C);
        free(x); free(y);
    }
    {
        const int L = 5;
        float *e = buf((size_t)B * 2 * L, 20), *eps = buf((size_t)B * L, 21), *z = buf((size_t)B * L, 22), *kl = buf(B, 23);
        for (int i = 0; i < B * 2 * L; ++i) e[i] *= 40.f;         /* exercise the +-10 clip */
        vvo_reparam_kl(e, eps, z, kl, B, L);
        free(e); free(eps); free(z); free(kl);
    }
    {
        const long V = 29;
        float *lg = buf((size_t)B * V, 24), *tg = buf((size_t)B * V, 25), *pr = buf((size_t)B * V, 26);
        float *bce = buf(B, 27), *tp = buf(B, 28), *fp = buf(B, 29), *fn = buf(B, 30);
        for (long i = 0; i < B * V; ++i) { lg[i] *= 60.f; tg[i] = tg[i] > 0.f ? 1.f : 0.f; }   /* saturating logits */
        vvo_sigmoid_bce_counts(lg, tg, pr, bce, tp, fp, fn, B, V, 0.6f, 1e-7f);
        free(lg); free(tg); free(pr); free(bce); free(tp); free(fp); free(fn);
    }
    (void)vvo_num_threads();
    puts("asan driver: ok");
    return 0;
}

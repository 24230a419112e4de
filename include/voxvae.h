/* voxvae.h -- C ABI of libvoxvae.so: the MI355X (gfx950) voxel VAE reconstruction hot path.
 *
 * The reference (bogus2000/anytime-3D-reconstruction) has NO FFI / plugin / operator interface: it is
 * Python on TensorFlow and its seam is the Python API of src/net_core + src/module.  The
 * drop-in for that seam is the Python package in anytime-3d-reconstruction_amd/src/ (same module paths, names,
 * signatures).  This header is the boundary UNDER that package: what a maintainer of the reference would bind
 * (ctypes stub in INTEGRATION.md) to replace the TensorFlow ops each function cites.  Citations are
 * /root/reference paths.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the name ends in _host; tensors are dense, channels-last
 *     (NDHWC), cubic grids with a power-of-two side; `stream` is a hipStream_t passed as void*.
 *   - activations/weights are VV_F32 or VV_BF16 (`dtype`); statistics, losses, logits, probabilities, latents
 *     are always float32.
 *   - functions return 0 (VV_OK) or a negative vv_status; they never throw, never allocate, never
 *     synchronise; scratch comes from the caller (`*_workspace_bytes`).  No global state: thread-safe per stream.
 */
#ifndef VOXVAE_H
#define VOXVAE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum { VV_F32 = 0, VV_BF16 = 1, VV_FP8 = 2 /* OCP e4m3fn, 1 byte; MFMA layers with cin % 128 == 0 only */ } vv_dtype;
typedef enum { VV_ACT_NONE = 0, VV_ACT_ELU = 1, VV_ACT_RELU = 2, VV_ACT_LRELU = 3 } vv_act;
typedef enum {
    VV_OK = 0,
    VV_ERR_NULL = -1,       /* required pointer is NULL */
    VV_ERR_SHAPE = -2,      /* unsupported / inconsistent shape */
    VV_ERR_DTYPE = -3,
    VV_ERR_ALIGN = -4,      /* pointer not 16-byte aligned */
    VV_ERR_WORKSPACE = -5,  /* workspace too small */
    VV_ERR_LAUNCH = -6      /* hipGetLastError() != hipSuccess after launch */
} vv_status;

int vv_abi_version(void);
const char *vv_status_string(int status);
/* hipGetErrorString of the HIP error behind this thread's most recent VV_ERR_LAUNCH. */
const char *vv_last_hip_error(void);

/* ---------------------------------------------------------------------------------------------------------
 * Weight packing: Keras variable layouts -> the K-contiguous [N][K] panels the MFMA kernels stream.
 * One-off per weight update; device to device. */

/* Conv3D kernel [4,4,4,Cin,Cout] (autoencoder3D.py:27) -> packed [Cout][64*Cin], k = tap*Cin + ci. */
int vv_pack_conv_k4(const float *w_keras, void *packed, int cin, int cout, int dtype, void *stream);
/* Conv3DTranspose kernel [4,4,4,Cout,Cin] (autoencoder3D.py:42) stride 2 -> 8 output-parity panels
 * packed [8][Cout][8*Cin]; parity p = (pd,ph,pw), k = (ad,ah,aw)*Cin + ci, tap t = 1 - p + 2a per axis. */
int vv_pack_convT_k4s2(const float *w_keras, void *packed, int cin, int cout, int dtype, void *stream);
/* Final encoder Conv3D k4 s1 SAME (pad 1/2) followed by the spatial mean (autoencoder3D.py:86-91) is linear in
 * its input: packed [Cout][S^3*Cin] with W_eff[i] = (1/S^3) * sum_o w[i - o + 1]  (S = input side). */
int vv_pack_conv_k4s1_meanpool(const float *w_keras, void *packed, int side, int cin, int cout, int dtype, void *stream);
/* The same final encoder conv position by position, for final_pool = 'max' (autoencoder3D.py:92-93: tf.reduce_max is not linear):
 * packed [S^3*Cout][S^3*Cin], row (o, co), column (i, ci) = w[i - o + 1][ci][co] (0 outside the 4 taps); a vv_dense_fwd with this
 * panel gives the conv output [B][S^3][Cout], vv_max_over_positions the pooled [B][Cout]. */
int vv_pack_conv_k4s1_full(const float *w_keras, void *packed, int side, int cin, int cout, int dtype, void *stream);
int vv_max_over_positions(const float *x, float *out, int batch, int npos, int channels, void *stream);
/* y = sigmoid(x), float32, n elements (in place allowed): encoder3D's final_activation 'sigmoid' (autoencoder3D.py:97-99). */
int vv_sigmoid_f32(const float *x, float *y, long n, void *stream);
/* First decoder Conv3DTranspose k4 s1 SAME on the S^3 x Cin seed (autoencoder3D.py:127-128, first loop
 * iteration) as one dense panel: packed [S^3*Cout][S^3*Cin], row (o,co), col (j,ci) = w[o - j + 1][co][ci]. */
int vv_pack_convT_k4s1_dense(const float *w_keras, void *packed, int side, int cin, int cout, int dtype, void *stream);
/* Dense kernel [In,Out] (autoencoder3D.py:59) -> packed [Out][In]. */
int vv_pack_dense(const float *w_keras, void *packed, int in, int out, int dtype, void *stream);
/* BatchNormalization(training=False) folded to y = x*scale + shift (autoencoder3D.py:31,46,62):
 * scale = gamma/sqrt(var+eps); shift = beta + (bias - mean)*scale.  bias may be NULL.  `repeat` tiles the C
 * channel vector (used when a spatial layer is run as one dense panel). */
int vv_fold_bn(const float *gamma, const float *beta, const float *mean, const float *var, const float *bias,
               float eps, float *scale, float *shift, int channels, int repeat, void *stream);

/* ---------------------------------------------------------------------------------------------------------
 * Layers (inference form: conv -> folded BN -> activation).  scale/shift may be NULL (identity). */

/* conv3DEnc with Cin = 1: Conv3D k4 s2 SAME on the occupancy grid + BN + act (autoencoder3D.py:26-39, first
 * loop iteration :84-85) on the same MFMA tile engine, K = the 64 taps gathered from x.  x [B,D,D,D,1] float32;
 * w_packed = vv_pack_conv_k4(cin = 1) = [Cout][64]; y [B,D/2,D/2,D/2,Cout] dtype. */
int vv_conv3d_first_fwd(const float *x, const void *w_packed, const float *scale, const float *shift, void *y,
                        int batch, int side, int cout, int act, int dtype, void *stream);
/* Same with a separate output element type: out_dtype VV_FP8 (e4m3fn) is available for dtype VV_BF16, Cout 64, side >= 32
 * (the hand-over into an fp8 second layer); otherwise out_dtype must equal dtype. */
int vv_conv3d_first_fwd_io(const float *x, const void *w_packed, const float *scale, const float *shift, void *y,
                           int batch, int side, int cout, int act, int dtype, int out_dtype, void *stream);

/* conv3DEnc, Cin % 64 == 0: Conv3D k4 s2 SAME + BN + act as an implicit GEMM on MFMA
 * (autoencoder3D.py:26-39).  x [B,D,D,D,Cin]; w = vv_pack_conv_k4; y [B,D/2,...,Cout]. */
size_t vv_conv3d_k4s2_workspace_bytes(int batch, int side, int cin, int cout, int dtype);
int vv_conv3d_k4s2_fwd(const void *x, const void *w_packed, const float *scale, const float *shift, void *y,
                       int batch, int side, int cin, int cout, int act, int dtype, void *workspace,
                       size_t workspace_bytes, void *stream);

/* conv3DDec stride 2: Conv3DTranspose k4 s2 SAME + BN + act as 8 output-parity implicit GEMMs
 * (autoencoder3D.py:41-54).  x [B,D,D,D,Cin]; w = vv_pack_convT_k4s2; y [B,2D,2D,2D,Cout]. */
size_t vv_convT3d_k4s2_workspace_bytes(int batch, int side, int cin, int cout, int dtype);
int vv_convT3d_k4s2_fwd(const void *x, const void *w_packed, const float *scale, const float *shift, void *y,
                        int batch, int side, int cin, int cout, int act, int dtype, void *workspace,
                        size_t workspace_bytes, void *stream);

/* The two stride-2 layers with separate operand / output element types: dtype VV_FP8 runs the implicit GEMM on
 * v_mfma_f32_32x32x16_fp8_fp8 (x and w_packed in OCP e4m3fn, one byte per element; cin % 128 == 0; per-output-channel
 * weight scales belong in `scale`), out_dtype chooses what the epilogue stores (VV_FP8 for the next fp8 layer, VV_BF16
 * for a bf16 consumer, VV_F32).  dtype VV_BF16 with out_dtype VV_FP8 is the hand-over into an fp8 stretch.  New in this
 * build (BASELINE config 5): the reference is float32 only.  Workspace: the *_workspace_bytes query with `dtype`. */
int vv_conv3d_k4s2_fwd_io(const void *x, const void *w_packed, const float *scale, const float *shift, void *y,
                          int batch, int side, int cin, int cout, int act, int dtype, int out_dtype, void *workspace,
                          size_t workspace_bytes, void *stream);
int vv_convT3d_k4s2_fwd_io(const void *x, const void *w_packed, const float *scale, const float *shift, void *y,
                           int batch, int side, int cin, int cout, int act, int dtype, int out_dtype, void *workspace,
                           size_t workspace_bytes, void *stream);

/* Direct variant of vv_conv3d_k4s2_fwd for the widest encoder layer (bf16, Cin 64 -> Cout 128, side >= 16): per input
 * phase q (x index parity per axis) the layer is a k2 s1 convolution over the phase sub-grid, so a 4x8x8 box of
 * outputs stages 8 phase tiles of 5x9x9 voxels in LDS instead of 64 im2col tiles; same vv_pack_conv_k4 weights.
 * vv_conv3d_k4s2_direct_supported() says whether a shape is covered (callers fall back to vv_conv3d_k4s2_fwd). */
int vv_conv3d_k4s2_direct_supported(int side, int cin, int cout, int dtype);
int vv_conv3d_k4s2_direct_fwd(const void *x, const void *w_packed, const float *scale, const float *shift, void *y,
                              int batch, int side, int cin, int cout, int act, int dtype, void *stream);
/* Same with the output stored as VV_BF16 or VV_FP8 (e4m3fn: the hand-over to an fp8 layer without a conversion pass). */
int vv_conv3d_k4s2_direct_fwd_io(const void *x, const void *w_packed, const float *scale, const float *shift, void *y,
                                 int batch, int side, int cin, int cout, int act, int dtype, int out_dtype, void *stream);
/* fp8 twin of the direct 64 -> 128 layer (conv_direct_fp8.hip): x and w_packed (vv_pack_conv_k4 with dtype VV_FP8, from a
 * kernel already divided by its per-output-channel scale; the scale belongs in `scale`) in OCP e4m3fn, block-scaled K = 64
 * MFMA, y stored as VV_FP8 or VV_BF16.  Input side >= 16. */
int vv_conv3d_k4s2_direct_fp8_supported(int side, int cin, int cout);
int vv_conv3d_k4s2_direct_fp8_fwd(const void *x, const void *w_packed, const float *scale, const float *shift, void *y,
                                  int batch, int side, int cin, int cout, int act, int out_dtype, void *stream);

/* Direct variant of vv_convT3d_k4s2_fwd for the widest decoder layer (bf16, Cin 128 -> Cout 64, side >= 8): the input
 * halo tile of a 4x4x8 block of cells is staged in LDS once and serves all 8 parities x 8 taps; weights come from the
 * MFMA-fragment-ordered panel of vv_pack_convT_k4s2_frag ([8 parity][8 tap][Cin/16][Cout/32][64 lanes][8]).
 * vv_convT3d_k4s2_direct_supported() says whether a shape is covered (callers fall back to vv_convT3d_k4s2_fwd). */
int vv_convT3d_k4s2_direct_supported(int side, int cin, int cout, int dtype);
int vv_pack_convT_k4s2_frag(const float *w_keras, void *packed, int cin, int cout, void *stream);
int vv_convT3d_k4s2_direct_fwd(const void *x, const void *w_frag, const float *scale, const float *shift, void *y,
                               int batch, int side, int cin, int cout, int act, int dtype, void *stream);
/* fp8 twin of the direct transposed layer (convt_direct_fp8.hip): x and w_frag in OCP e4m3fn (w_frag from
 * vv_pack_convT_k4s2_frag_fp8, which converts a float32 Keras kernel already divided by its per-output-channel scale; the
 * scale belongs in `scale`), block-scaled K = 64 MFMA, y stored as VV_BF16 or VV_FP8 (out_dtype; VV_FP8 feeds the fp8 form of
 * vv_convT3d_final_bce_fwd).  Cin 128 -> Cout 64, side >= 8. */
int vv_convT3d_k4s2_direct_fp8_supported(int side, int cin, int cout);
int vv_pack_convT_k4s2_frag_fp8(const float *w_keras, void *packed, int cin, int cout, void *stream);
int vv_convT3d_k4s2_direct_fp8_fwd(const void *x, const void *w_frag, const float *scale, const float *shift, void *y,
                                   int batch, int side, int cin, int cout, int act, int out_dtype, void *stream);

/* conv3DEnc / conv3DDec between the 8^3 and the 4^3 grid (autoencoder3D.py:26-54; the 32^3 model's 128 -> 256 and
 * 256 -> 128 layers, the 64^3 model's 256 -> 512 and 512 -> 256) with four whole samples resident in LDS per workgroup
 * and the taps that fall into the SAME padding skipped at MFMA-tile granularity in d and h (bf16 only).
 *   conv : x [B,8,8,8,Cin] -> y [B,4,4,4,Cout], Cin % 64 == 0, Cout % 64 == 0; w_skip = vv_pack_conv_k4_skip
 *          = [64 taps][Cin/64][Cout][64]
 *   convT: x [B,4,4,4,Cin] -> y [B,8,8,8,Cout], Cin % 64 == 0, Cout % 128 == 0; w_skip = vv_pack_convT_k4s2_skip
 *          = [8 parities][8 taps][Cin/64][Cout][64], tap t = 1 - p + 2a per axis
 * No workspace; batches whose input passes 2 GiB go out as several launches. */
int vv_conv3d_k4s2_skip_supported(int side, int cin, int cout, int dtype);
int vv_convT3d_k4s2_skip_supported(int side, int cin, int cout, int dtype);
int vv_pack_conv_k4_skip(const float *w_keras, void *packed, int cin, int cout, void *stream);
/* Several of the two images above in one call (round 4: the training step needs nine per step and each single call is a 5-8 us launch):
 * kinds[j] = 0: vv_pack_conv_k4_skip, 1: vv_pack_convT_k4s2_skip of (w_keras[j], cin[j], cout[j]) into packed[j]; one launch per kind (and per
 * eight jobs); bit-identical to the single calls.  The arrays are HOST arrays of njobs entries. */
int vv_pack_skip_images(const int *kinds, const float *const *w_keras, void *const *packed, const int *cin, const int *cout, int njobs,
                        void *stream);
int vv_pack_convT_k4s2_skip(const float *w_keras, void *packed, int cin, int cout, void *stream);
int vv_conv3d_k4s2_skip_fwd(const void *x, const void *w_skip, const float *scale, const float *shift, void *y, int batch,
                            int side, int cin, int cout, int act, int dtype, void *stream);
int vv_convT3d_k4s2_skip_fwd(const void *x, const void *w_skip, const float *scale, const float *shift, void *y, int batch,
                             int side, int cin, int cout, int act, int dtype, void *stream);

/* conv3DDec 8^3 x 128 -> 16^3 x 64 (autoencoder3D.py:41-54; the 32^3 model's widest decoder layer) with one WHOLE sample
 * resident in LDS per workgroup (128 KiB, no halo: out-of-grid taps read a zero row), all eight waves on the same output
 * parity sharing each weight chunk through an LDS ring (bf16 only).  w_skip = vv_pack_convT_k4s2_skip's image.  One
 * workgroup per (sample, parity split); the 8 parities are split over 2 / 4 / 8 workgroups until the grid holds two rounds
 * of workgroups per CU (VV_CTW_PS overrides; VV_CTW_SHAPE=32 selects the 32x32x16 MFMA form, default 16x16x32).  Replaces vv_convT3d_k4s2_direct_fwd at this shape; no workspace. */
int vv_convT3d_k4s2_whole_supported(int side, int cin, int cout, int dtype);
int vv_convT3d_k4s2_whole_fwd(const void *x, const void *w_skip, const float *scale, const float *shift, void *y, int batch,
                              int side, int cin, int cout, int act, int dtype, void *stream);

/* conv3DEnc / conv3DDec between the 4^3 and the 2^3 grid (the 32^3 model's 256 -> 512 and 512 -> 256 layers) as a
 * position-major split-K GEMM: rows = the batch at one output position, K = that position's valid (tap, 64-channel chunk)
 * pairs cut into equal shares over the workgroups, 256 x 128 tiles, 3-stage LDS-DMA ring (bf16 only).  Weights in the
 * vv_pack_conv_k4_skip / vv_pack_convT_k4s2_skip layouts.  Workspace = float32 slabs of the positions that are cut into
 * several shares (summed in share order: deterministic).
 *   conv : x [B,4,4,4,Cin] -> y [B,2,2,2,Cout];  convT: x [B,2,2,2,Cin] -> y [B,4,4,4,Cout];  Cin % 64 == 0, Cout % 8 == 0 */
int vv_conv3d_k4s2_pos_supported(int side, int cin, int cout, int dtype);
int vv_convT3d_k4s2_pos_supported(int side, int cin, int cout, int dtype);
size_t vv_conv3d_k4s2_pos_workspace_bytes(int batch, int cin, int cout);
size_t vv_convT3d_k4s2_pos_workspace_bytes(int batch, int cin, int cout);
int vv_conv3d_k4s2_pos_fwd(const void *x, const void *w_skip, const float *scale, const float *shift, void *y, int batch,
                           int side, int cin, int cout, int act, int dtype, void *workspace, size_t workspace_bytes, void *stream);
int vv_convT3d_k4s2_pos_fwd(const void *x, const void *w_skip, const float *scale, const float *shift, void *y, int batch,
                            int side, int cin, int cout, int act, int dtype, void *workspace, size_t workspace_bytes, void *stream);

/* y[M,N] = act((x[M,K] @ w_packed[N,K]^T) * scale[N] + shift[N]): linearTransform (autoencoder3D.py:56-70) and
 * the two layers packed as dense panels above.  K % 8 == 0 (bf16) / % 4 (f32), N % 4 == 0 (tails are masked).
 * out_dtype may differ from dtype (the encoder output is float32). */
size_t vv_dense_workspace_bytes(int m, int n, int k, int dtype);
int vv_dense_fwd(const void *x, const void *w_packed, const float *scale, const float *shift, void *y, int m, int n,
                 int k, int act, int dtype, int out_dtype, void *workspace, size_t workspace_bytes, void *stream);

/* The latent tail of the evaluation path in two launches (latent_tail.hip): encoder tail (the vv_pack_conv_k4s1_meanpool
 * panel w5 [E][K5] applied to the last stride-2 activation h [B][K5]) -> slice | clip | sampling | KL (as vv_reparam_kl_fwd,
 * no dropout) -> linearTransform + BN + act (wd [lin][L], vv_pack_dense) -> first decoder layer as a dense panel
 * (w1 [n1][lin], vv_pack_convT_k4s1_dense) + BN + act.  Outputs: enc_out [B][E] (optional), z [B][L], z_act bf16 (optional),
 * kl [B] (variational), h1 [B][n1] bf16 = the input of the first stride-2 decoder layer.  variational: E = 2L, else E = L and
 * z = enc_out.  bf16 only; L % 32 == 0, lin % 32 == 0, n1 % 16 == 0.  e5_scale: optional per-channel factor on enc_out. */
int vv_latent_tail_supported(int K5, int E, int L, int lin, int n1, int variational, int dtype);
size_t vv_latent_tail_workspace_bytes(int batch, int K5, int E, int n1);
int vv_latent_tail_fwd(const void *h, const void *w5, const float *e5_scale, const float *eps, const void *wd, const float *scale_d,
                       const float *shift_d, const void *w1, const float *scale_1, const float *shift_1, float *enc_out, float *z,
                       void *z_act, float *kl, void *h1, int batch, int K5, int E, int L, int lin, int n1, int variational, int act,
                       int dtype, void *workspace, size_t workspace_bytes, void *stream);

/* The last stride-2 encoder layer (conv3DEnc at 4^3 -> 2^3, autoencoder3D.py:26-39 / :84-85) AND the latent tail above in three
 * launches (posgemm.hip + latent_tail.hip): the convolution leaves only its float32 split-K partial sums, and the encoder-tail
 * kernel sums them in share order, applies that layer's folded BN (scale4 / shift4, may be NULL) + act and rounds to bf16 while it
 * builds its own operand -- the same values, in the same order, as vv_conv3d_k4s2_pos_fwd followed by vv_latent_tail_fwd, without the
 * reduce launch and without the [B][2^3][cout4] activation in memory.  x4 [B][4^3][cin4] bf16, w4_skip = vv_pack_conv_k4_skip's
 * image; K5 = 8 * cout4; the other arguments as vv_latent_tail_fwd.  cout4 % 256 == 0, E % 16 == 0, E <= 128. */
int vv_conv_pos_latent_tail_supported(int cin4, int cout4, int E, int L, int lin, int n1, int variational, int dtype);
size_t vv_conv_pos_latent_tail_workspace_bytes(int batch, int cin4, int cout4, int E);
int vv_conv_pos_latent_tail_fwd(const void *x4, const void *w4_skip, const float *scale4, const float *shift4, int cin4, int cout4,
                                const void *w5, const float *e5_scale, const float *eps, const void *wd, const float *scale_d,
                                const float *shift_d, const void *w1, const float *scale_1, const float *shift_1, float *enc_out,
                                float *z, void *z_act, float *kl, void *h1, int batch, int E, int L, int lin, int n1, int variational,
                                int act, int dtype, void *workspace, size_t workspace_bytes, void *stream);

/* ---------------------------------------------------------------------------------------------------------
 * Latent ops */

/* slice | clip(-10,10) | sampling | Dropout | kl_loss vs N(0,I), fused (nolbo.py:1417-1431, 1464-1470;
 * function.py:35-38, 84-98).  enc_out [B,2L]; eps [B,L] (the tf.random.normal draw, injected);
 * drop_mask [B,L] in {0,1} or NULL, drop_scale = 1/(1-rate); z [B,L] float32; z_act [B,L] = the same values in
 * the activation dtype `act_dtype` that feeds the decoder (may be NULL); kl [B] (may be NULL); mean/logvar [B,L]
 * outputs may be NULL. */
int vv_reparam_kl_fwd(const float *enc_out, const float *eps, const float *drop_mask, float drop_scale, float *z,
                      void *z_act, int act_dtype, float *kl, float *mean, float *logvar, int batch, int latent,
                      void *stream);

/* ---------------------------------------------------------------------------------------------------------
 * Decoder tail + losses, fused: Conv3DTranspose k4 s2 SAME -> 1 channel (logits), tf.sigmoid, binary_loss
 * (gamma, epsilon clip) and voxelPrecisionRecall at p >= 0.5 (autoencoder3D.py:129-136; function.py:73-82,
 * 100-115; nolbo.py:1496-1499).  x [B,D,D,D,Cin] dtype; w_keras [4,4,4,1,Cin] float32; target [B,2D,2D,2D]
 * float32; probs/logits [B,2D,2D,2D] float32 (either may be NULL); stats [B,4] = per-sample (bce,TP,FP,FN). */
/* dtype VV_FP8: x in OCP e4m3fn (side >= 8; the kernel quantises w_keras per tap itself); probabilities, logits and the sums
 * stay float32. */
size_t vv_convT3d_final_bce_workspace_bytes(int batch, int side);
int vv_convT3d_final_bce_fwd(const void *x, const float *w_keras, const float *target, float *probs, float *logits,
                             float *stats, int batch, int side, int cin, float gamma, float epsilon, int dtype,
                             void *workspace, size_t workspace_bytes, void *stream);
/* Same, plus the batch metrics of nolbo.py:1498-1501 in the same reduction launch: metrics4 = (mean bce, mean TP/(TP+FP+1e-10),
 * mean TP/(TP+FN+1e-10), mean IoU) -- what vv_shape_metrics computes from `stats`. */
int vv_convT3d_final_bce_metrics_fwd(const void *x, const float *w_keras, const float *target, float *probs, float *logits,
                                     float *stats, float *metrics4, int batch, int side, int cin, float gamma, float epsilon,
                                     int dtype, void *workspace, size_t workspace_bytes, void *stream);

/* Batch means of nolbo.py:1498-1501: out[0..3] = mean_b bce, mean_b TP/(TP+FP+1e-10), mean_b TP/(TP+FN+1e-10),
 * mean_b TP/max(TP+FP+FN,1) (IoU: not in the reference). */
int vv_shape_metrics(const float *stats, float *out4, int batch, void *stream);

/* ---------------------------------------------------------------------------------------------------------
 * Missing-modality evaluation (nolbo.py:1472-1518; AE: 1277-1322; Pascal: 877-918).  prototypes = the
 * category_vectors array [C,L]; mask [B,L] in {0,1} is the np.random.choice draw of nolbo.py:1475, injected. */

/* z_out = z*mask, then where(z_out == 0, mean_c prototypes, z_out)  (nolbo.py:1477-1482). */
int vv_latent_mask_fill(const float *z, const float *mask, const float *prototypes, int classes, float *z_out,
                        void *z_act, int act_dtype, int batch, int latent, void *stream);
/* argmin[b] = first argmin_c sum_j mask_bj (z_bj - P_cj)^2; mask NULL = all ones (nolbo.py:1489-1493, 1505-1506). */
int vv_nearest_category(const float *z, const float *mask, const float *prototypes, int classes, int *argmin, int batch,
                        int latent, void *stream);
/* z_corr = where(mask == 0, P[argmin] + eps2, z)  (nolbo.py:1507-1510; eps2 = the second normal draw, injected). */
int vv_latent_correct(const float *z, const float *mask, const float *prototypes, const int *argmin, const float *eps2,
                      float *z_corr, void *z_act, int act_dtype, int batch, int latent, void *stream);
/* acc[0] = mean_b [argmin_b == argmax_c onehot_bc]  (nolbo.py:1493-1494). */
int vv_category_accuracy(const int *argmin, const float *onehot, int classes, float *acc, int batch, void *stream);

/* ---------------------------------------------------------------------------------------------------------
 * Stand-alone loss ops with the signatures of src/module/function.py (the fused kernels above are what the
 * model classes use; these serve callers that compose the ops themselves). */

/* binary_loss(xPred probabilities, xTarget, epsilon, gamma, b_range) -> [B]  (function.py:73-82). */
int vv_binary_loss(const float *pred, const float *target, float epsilon, float gamma, float b_range, float *out, int batch,
                   long voxels, void *stream);
/* voxelPrecisionRecall(xTarget, xPred, prob) -> TP,FP,FN [B]  (function.py:100-115). */
int vv_voxel_precision_recall(const float *target, const float *pred, float prob, float *tp, float *fp, float *fn, int batch,
                              long voxels, void *stream);
/* kl_loss(mean, logVar, mean_target, logVar_target) -> [B]  (function.py:84-98). */
int vv_kl_loss(const float *mean, const float *logvar, const float *mean_target, const float *logvar_target, float *out,
               int batch, int latent, void *stream);
/* regulizer_loss (function.py:40-71): pairwise hinge on the scaled L1 distance of latent means,
 * out[i] = sum_j same_class(i,j) * min(sum_l |m_i - m_j| / exp(0.5 lv_i) - dist_in_z_space, 0)^2; class_input may be
 * NULL (every pair counts).  mean, logvar [B,L]; class_input [B,C]; out [B]. */
int vv_regulizer_loss(const float *mean, const float *logvar, const float *class_input, float dist_in_z_space,
                      float *out, int batch, int latent, int class_dim, void *stream);
/* sampling(mu, logVar) = mu + sqrt(exp(logVar))*eps, eps injected  (function.py:35-38). */
int vv_sampling(const float *mu, const float *logvar, const float *eps, float *out, long n, void *stream);

/* ---------------------------------------------------------------------------------------------------------
 * Training path (nolboSingleObject_modelnet_category_{VAE,AE}.fit, nolbo.py:1411-1447 / 1230-1258).  float32.
 * Data gradients reuse the forward kernels: d(Conv3D k4 s2)/d(input) = vv_convT3d_k4s2_fwd with the SAME Keras
 * kernel array packed by vv_pack_convT_k4s2 (read as [4,4,4,Cout_T = Cin, Cin_T = Cout]); d(Conv3DTranspose k4 s2)/
 * d(input) = vv_conv3d_k4s2_fwd with vv_pack_conv_k4 of the same array; Dense: vv_dense_fwd with the Keras
 * [in,out] array as the [N][K] panel. */

/* BatchNormalization(training=True) statistics over x[rows][channels] (autoencoder3D.py:31,46,62): batch mean,
 * biased variance, rstd = 1/sqrt(var+eps), folded scale = gamma*rstd, shift = beta - mean*scale; moving statistics
 * updated in place with `momentum` (Keras 0.99) when the pointers are non-NULL. */
size_t vv_bn_workspace_bytes(long rows, int channels);
int vv_bn_train_stats(const void *x, long rows, int channels, const float *gamma, const float *beta, float eps,
                      float momentum, float *mean, float *var, float *rstd, float *scale, float *shift,
                      float *moving_mean, float *moving_var, int dtype, void *workspace, size_t workspace_bytes,
                      void *stream);
/* Batch statistics from per-block column sums a producer kernel left (round 4): the widest decoder layer's training-mode forward,
 * vv_convT3d_k4s2_whole_stats_fwd (raw bf16 output + partial[(block * 2 + {sum, sum of squares}) * cout + channel], block = sample;
 * vv_convT3d_k4s2_whole_stats_blocks(batch) blocks), and the finalisation of vv_bn_train_stats on such partials: the statistics
 * sweep over the layer's output is not run.  Same results as vv_convT3d_k4s2_whole_fwd + vv_bn_train_stats up to the float32
 * summation order of the column sums (finalised in double). */
int vv_convT3d_k4s2_whole_stats_blocks(int batch);
int vv_convT3d_k4s2_whole_stats_fwd(const void *x, const void *w_skip, void *y, float *stats_partial, size_t stats_bytes, int batch,
                                    int side, int cin, int cout, int dtype, void *stream);
int vv_bn_finalize_stats(const float *partial, int nblocks, long rows, int channels, const float *gamma, const float *beta, float eps,
                         float momentum, float *mean, float *var, float *rstd, float *scale, float *shift, float *moving_mean,
                         float *moving_var, void *stream);
/* y = act(x*scale + shift) */
int vv_bn_act_fwd(const void *x, const float *scale, const float *shift, void *y, long rows, int channels, int act,
                  int dtype, void *stream);
/* Backward of act(BN(x)): given dy = dL/dy, writes dgamma, dbeta [channels] and dx = dL/dx [rows][channels]. */
int vv_bn_act_bwd(const void *x, const void *dy, const float *scale, const float *shift, const float *mean,
                  const float *rstd, float *dgamma, float *dbeta, void *dx, long rows, int channels, int act,
                  int dtype, void *workspace, size_t workspace_bytes, void *stream);

/* Weight gradients as reduction-over-rows GEMMs on the exact-f32 MFMA: dw[m][n] = sum_r a[r][m] * g[r][n].  The operands
 * may be float32 or bf16 independently (*_dtype; bf16 is widened when staged); dw is always float32. */
size_t vv_wgrad_workspace_bytes(long rows, int m, int n);
int vv_wgrad_dense(const void *a, const void *g, float *dw, long rows, int m, int n, int lda, int a_dtype, int g_dtype,
                   void *workspace, size_t workspace_bytes, void *stream);
/* a[r][(t,ci)] = src[b, 2o-1+t, ci] gathered over the half-size grid (zero in the SAME padding); g [B*(side/2)^3][cout].
 * Conv3D k4 s2 (src = layer input, g = dL/d(conv out)): dw = Keras [4,4,4,cin,cout].  Conv3DTranspose k4 s2 (src =
 * dL/d(out) on the big grid, g = layer input): dw = Keras [4,4,4,Cout,Cin] with (cin, cout) := (Cout, Cin).
 * cin == 1 (first conv / last transposed conv) or cin % 64 == 0. */
int vv_wgrad_conv_k4s2(const void *src, const void *g, float *dw, int batch, int side, int cin, int cout, int src_dtype,
                       int g_dtype, void *workspace, size_t workspace_bytes, void *stream);
/* Adjoints of vv_pack_conv_k4s1_meanpool / vv_pack_convT_k4s1_dense: panel gradient -> Keras kernel gradient. */
int vv_unpack_meanpool_grad(const float *dpanel, float *dw, int side, int cin, int cout, void *stream);
int vv_unpack_convT_dense_grad(const float *dpanel, float *dw, int side, int cin, int cout, void *stream);
/* out[cols][rows] = in[rows][cols]^T (panel transposes for the dense-panel data gradients). */
int vv_transpose_f32(const float *in, float *out, int rows, int cols, void *stream);
/* out[c] = sum_r x[r][c] (Dense bias gradient). */
int vv_colsum(const void *x, float *out, long rows, int cols, int dtype, void *stream);

/* dlogit = d(mean_b binary_loss_b)/dlogit through sigmoid and the epsilon clip (function.py:73-82). */
int vv_bce_bwd(const float *probs, const float *target, float *dlogit, int batch, long voxels, float gamma, float epsilon,
               float inv_batch, void *stream);
/* Backward of vv_reparam_kl_fwd for total = mean_b KL + ...: d_enc_out [B,2L] from dz [B,L]. */
int vv_reparam_kl_bwd(const float *enc_out, const float *eps, const float *dz, const float *drop_mask, float drop_scale,
                      float *d_enc_out, int batch, int latent, float inv_batch, void *stream);
/* Keras Adam: m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2; param -= lr_t m / (sqrt(v) + eps),
 * lr_t = lr sqrt(1-b2^t)/(1-b1^t) computed by the caller (nolbo.py:1402, 1441). */
int vv_adam_step(float *param, const float *grad, float *m, float *v, long n, float lr_t, float beta1, float beta2,
                 float epsilon, void *stream);

/* vv_adam_step for every variable in one launch: chunk_table is a device array of nchunks records
 * { float *param; const float *grad; float *m; float *v; long n; } (40 bytes, pointers already offset, n <= 16384). */
int vv_adam_step_multi(const void *chunk_table, int nchunks, float lr_t, float beta1, float beta2, float epsilon,
                       void *stream);

/* dst[i] = (dst type) src[i] between float32 and bf16 (round to nearest even): operand casts of the mixed-precision
 * training step. */
int vv_convert(const void *src, void *dst, long n, int src_dtype, int dst_dtype, void *stream);

/* ---- bit-packed occupancy grids (modelnet_dataset.py:74-91 keeps float32 grids on the host and copies a batch per
 * iteration; 32^3 voxels are 4 KiB as bits).  Voxel v of a sample is bit (v & 7) of byte v >> 3.
 * vv_unpack_bits_gather: out[b][v] = bit v of sample index[b] (index NULL: b) as 0.0f / 1.0f; voxels % 8 == 0.
 * vv_pack_bits: bit = x >= threshold, over n floats (n % 8 == 0). */
int vv_unpack_bits_gather(const void *packed, const int *index, float *out, int batch, long voxels, void *stream);
int vv_pack_bits(const float *x, void *packed, float threshold, long n, void *stream);

#ifdef __cplusplus
}
#endif
#endif

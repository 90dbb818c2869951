/*
 * mmseg_hip.h -- C ABI of libmmseg_hip.so, the gfx950 (MI355X / CDNA4) kernel library behind the
 * DAFNet / MMSDNet training step.
 *
 * The reference (agis85/multimodal_segmentation) has no FFI of its own: its arithmetic lives in Keras 2.1.6 /
 * TensorFlow 1.4 ops reached through the Python layer API.  Each entry point below replaces the TF op(s) that
 * the cited reference line instantiates; the Python host code in multimodal_segmentation_amd/ binds them with
 * ctypes (multimodal_segmentation_amd/_native.py) -- the binding a maintainer of the reference would add is
 * shown in INTEGRATION.md.
 *
 * Conventions
 *   - all tensors are dense fp32, NHWC, in device (HBM) memory; the caller owns every buffer (outputs and
 *     workspaces are allocated by the caller; *_workspace_floats() tell how much); no allocation, no
 *     synchronisation and no host<->device copy happens inside any entry point, so all of them may be
 *     captured into a hipGraph;
 *   - `stream` is a hipStream_t passed as void*; every launch goes to that stream;
 *   - the return value is a hipError_t as int (0 = hipSuccess); invalid geometry returns hipErrorInvalidValue
 *     without launching anything;
 *   - reductions are two-stage and run in a fixed order: results are bitwise reproducible run to run
  *     (the one scatter, mmseg_tps_warp_bwd's d_vol, accumulates in 64-bit fixed point with integer atomics, which is
 *     order-independent as well).
 */
#ifndef MMSEG_HIP_H
#define MMSEG_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

/* activation codes */
#define MMSEG_ACT_NONE 0
#define MMSEG_ACT_RELU 1
#define MMSEG_ACT_LEAKY 2
#define MMSEG_ACT_TANH 3

/* ---- convolution family (csrc/conv.hip): keras Conv2D [+UpSampling2D] [+Concatenate] -----------------------
 * reference: models/unet.py:94-101, utils/model_utils.py:15-22, model_components/segmentor.py:16-24,
 * modality_encoder.py:36-45, decoder.py:28,45-48, layers/spade.py:9-33, layers/stn_spline.py:106-114,
 * models/discriminator.py:24,39.
 * Implicit GEMM on v_mfma_f32_32x32x2_f32.  Logical input [B,H,W,C1+C2] = concat(x1 (optionally stored at
 * H/2 x W/2 and nearest-upsampled: ups=1), x2); kernel w [KH,KW,C1+C2,Cout] (keras HWIO); output [B,Ho,Wo,Cout].
 * transposed=1 selects the fractionally-strided gather used for the data gradient of a strided convolution
 * (tap valid iff (ho + kh - pad) % stride == 0).  y2/nsplit1: optional channel-split of the output
 * (channels [0,nsplit1) -> y, the rest -> y2), used for the data gradient of a two-input convolution. */
int mmseg_conv2d_fwd(const float* x1, const float* x2, const float* w, const float* wt, const float* bias, float* y, float* y2,
                     int B, int H, int W, int C1, int C2, int Ho, int Wo, int Cout, int KH, int KW, int stride,
                     int pad_h, int pad_w, int ups, int transposed, int act, float alpha, int nsplit1, void* stream);
/* precision of the fast-path convolutions -- forward, data gradient, weight gradient (process-wide): 0 = fp32 MFMA (default),
 * 1 = operands rounded to bf16 (RNE) + v_mfma_f32_32x32x16_bf16, 2 = fp16 + v_mfma_f32_32x32x16_f16, both with fp32 accumulation; HBM tensors, weight gradients and everything else stay fp32
 * (BASELINE configs #3 / #5: reduced-precision compute with fp32 master weights and fp32 gradient all-reduce).
 * mmseg_set_conv_precision returns the previous mode. */
int mmseg_set_conv_precision(int mode);
/* Which kernel multiplies 16-bit tensors with channel counts that are multiples of 64 (the UNet / SPADE 3x3 layers of
 * models/unet.py:94-101, layers/spade.py:26-33 in the reduced-precision configurations): 0 = always the 128-wide register-staged
 * kernel, 1 (default) = the 256-pixel direct-to-LDS kernel where the launch has enough tiles, 2 = wherever it applies.  Same
 * 16-bit products and fp32 accumulation, K tiles of another depth: results agree to fp32 rounding.  Returns the previous mode; other
 * values only query. */
int mmseg_conv16_mode(int mode);
int mmseg_get_conv_precision(void);
/* y = act(conv(x) * oscale[c] + bias[c]): convolution + inference-mode BatchNormalization (+ReLU) in one launch (`predict` of the
 * conv blocks of models/unet.py:94-101 and model_components/segmentor.py:16-22); oscale / bias from mmseg_bn_infer_fold */
int mmseg_conv2d_fwd_scaled(const float* x1, const float* x2, const float* w, const float* wt, const float* bias, const float* oscale,
                            float* y, int B, int H, int W, int C1, int C2, int Ho, int Wo, int Cout, int KH, int KW, int stride,
                            int pad_h, int pad_w, int ups, int act, float alpha, void* stream);
/* mmseg_conv2d_fwd / _fwd_scaled with 16-bit tensors in HBM (build-defined reduced-precision storage; mmseg_set_conv_precision(1 | 2)
 * selects bf16 | fp16): io bit 0 = x1, bit 1 = x2 hold 16-bit elements (MFMA fast path only), bit 2 = y / y2 are written 16-bit. */
int mmseg_conv2d_fwd_t(const void* x1, const void* x2, const float* w, const float* wt, const float* bias, void* y, void* y2,
                       int B, int H, int W, int C1, int C2, int Ho, int Wo, int Cout, int KH, int KW, int stride,
                       int pad_h, int pad_w, int ups, int transposed, int act, float alpha, int nsplit1, int io, void* stream);
int mmseg_conv2d_fwd_scaled_t(const void* x1, const void* x2, const float* w, const float* wt, const float* bias, const float* oscale,
                              void* y, int B, int H, int W, int C1, int C2, int Ho, int Wo, int Cout, int KH, int KW, int stride,
                              int pad_h, int pad_w, int ups, int act, float alpha, int io, void* stream);
/* which kernel template the last convolution entry point launched: family * 1000000 + 500000 * flag + (M | K tile) * 1000 + N tile (flag:
 * 16-byte gather of the generic kernels / two inputs of conv_wgrad_tr_kernel); families
 * 1 conv_fast_kernel, 2 conv_fwd_kernel, 3 conv_direct_kernel, 4 conv_fast_batched_kernel, 5 conv_dgrad_s2k4_smallc_kernel,
 * 6 conv_wgrad_tr_kernel, 7 conv_wgrad_fast_kernel, 8 conv_wgrad_kernel, 9 conv_wgrad_c8_kernel (profiling aid, no launch) */
int mmseg_conv2d_last_kernel(void);
/* fast path (Cin % 32 == 0, not transposed, at most 32 taps -- the reference's largest kernel is 5 x 5; more taps fall back to the
 * generic kernel, which reads `w`): K tiles lie inside one tap, gather by buffer loads, weights read from
 * `wt` = the kernel re-laid out as [Cout][K] by mmseg_conv2d_wprep (mode 0 forward, mode 1 data gradient incl. the
 * spatial flip, mode 2 the kernel as it is, in the image's element type); pass wt = NULL to force the generic kernel.  In the reduced-precision modes (mmseg_set_conv_precision 1 | 2) the
 * image holds elements of the 16-bit MFMA operand type (rounded once by the prep launch; it occupies the first half of the same
 * fp32-sized buffer) -- prepare it in the mode the convolution runs in. */
int mmseg_conv2d_fast_path(int C1, int C2, int Cout, int transposed);
int mmseg_conv2d_wprep(const float* w, float* out, int KH, int KW, int Cin, int Cout, int mode, void* stream);
/* the mode 0 / 1 images of n weights in ONE launch (all images of a model after its optimiser step); table: device int64 [n + 1][8] rows
 * {src, dst, KH*KW, Cin, Cout, mode, first block, 0}, row n carries the total block count; blocks per row = taps * ceil(Cin/32) * ceil(Cout/32) */
int mmseg_conv2d_wprep_batch(const long long* table, int n, long total_blocks, void* stream);
/* data gradient of a STRIDED convolution, one launch per parity class (ph, pw) of the input pixels: only the taps
 * kh = ph + s*a, kw = pw + s*b contribute, so each class is a stride-1 convolution over dy with a small sub-kernel and a
 * strided store -- no multiplications by the zeros of a dilated gradient.  wt from mmseg_conv2d_wprep_parity. */
int mmseg_conv2d_parity_taps(int K, int stride, int p);
int mmseg_conv2d_wprep_parity(const float* w, float* out, int KH, int KW, int Cin, int Cout, int stride, int ph, int pw, void* stream);
/* all stride x stride classes in one launch: out = the class images back to back in (ph, pw) raster order (what mmseg_conv2d_dgrad_parity_all reads) */
int mmseg_conv2d_wprep_parity_all(const float* w, float* out, int KH, int KW, int Cin, int Cout, int stride, void* stream);
int mmseg_conv2d_dgrad_parity(const float* dy, const float* wt, float* dx, int B, int Ho, int Wo, int Cout, int H, int W, int Cin,
                              int TH, int TW, int stride, int ph, int pw, void* stream);
/* data gradient of the discriminators' first layer (models/discriminator.py:24: 4x4, stride 2, valid; Cin = 1 or 4, Cout = 64):
 * direct kernel, w in the Keras layout */
int mmseg_conv2d_dgrad_s2k4_smallc(const float* dy, const float* w, float* dx, int B, int H, int W, int Cin, int Ho, int Wo, int Cout,
                                   void* stream);
/* all stride x stride parity classes of the data gradient in one batched launch; wt_all = the classes' sub-kernels from
 * mmseg_conv2d_wprep_parity back to back in (ph, pw) raster order */
int mmseg_conv2d_dgrad_parity_all(const float* dy, const float* wt_all, float* dx, int B, int Ho, int Wo, int Cout, int H, int W,
                                  int Cin, int KH, int KW, int stride, void* stream);
/* Data gradient of a stride-1 convolution with few input channels (segmentor c0 8 -> 64, the SPADE shared convolutions 8 -> 128:
 * model_components/segmentor.py:16, layers/spade.py:29), second half: dx = sum over the taps of the shifted planes of
 * T [B,Ho,Wo,KH*KW*Cin], which the caller computes with ONE 1x1 mmseg_conv2d_fwd over dy (wt = the Keras kernel itself read as
 * [KH*KW*Cin][Cout]). */
int mmseg_conv2d_dgrad_tapsum(const float* T, float* dx, int B, int H, int W, int Ho, int Wo, int Cin, int KH, int KW, int ph, int pw,
                              void* stream);
long mmseg_conv2d_wgrad_workspace(int B, int Ho, int Wo, int Cin, int Cout, int KH, int KW);
/* mmseg_conv2d_wgrad with 16-bit operands in HBM (reduced-precision modes; stride 1 and Wo % 4 == 0 only): io bit 0 = x1 (and x2),
 * bit 2 = dy hold 16-bit elements; dw and the workspace stay fp32 */
int mmseg_conv2d_wgrad_t_supported(int Ho, int Wo, int stride, int C1, int C2, int Cout);
int mmseg_conv2d_wgrad_t(const void* x1, const void* x2, const void* dy, float* dw, float* ws, long ws_floats,
                         int B, int H, int W, int C1, int C2, int Ho, int Wo, int Cout, int KH, int KW, int stride,
                         int pad_h, int pad_w, int ups, int accumulate, int io, void* stream);
/* dW[KH,KW,Cin,Cout] (+)= sum over output pixels of im2col(x)^T * dy (accumulate != 0 adds to dW: gradient arenas);
 * ws: mmseg_conv2d_wgrad_workspace floats */
int mmseg_conv2d_wgrad(const float* x1, const float* x2, const float* dy, float* dw, float* ws, long ws_floats,
                       int B, int H, int W, int C1, int C2, int Ho, int Wo, int Cout, int KH, int KW, int stride,
                       int pad_h, int pad_w, int ups, int accumulate, void* stream);
/* wt[kh][kw][co][ci] = w[KH-1-kh][KW-1-kw][ci][co]: the kernel of the data-gradient convolution */
int mmseg_conv2d_wflip(const float* w, float* wt, int KH, int KW, int Cin, int Cout, void* stream);

/* ---- element-wise and small reductions (csrc/pointwise.hip) --------------------------------------------- */
int mmseg_act_fwd(const float* x, float* y, long n, int act, float alpha, void* stream);
/* dx = dy * act'(.) evaluated from the activation OUTPUT y */
int mmseg_act_bwd(const float* dy, const float* y, float* dx, long n, int act, float alpha, void* stream);
int mmseg_axpby(const float* a, const float* b, float* out, long n, float sa, float sb, void* stream);
int mmseg_fill(float* x, long n, float v, void* stream);
int mmseg_colsum_blocks(long M);
long mmseg_colsum_workspace_floats(long M, int C);
/* out[c] (+)= scale * sum_m x[m][c]; ws: mmseg_colsum_workspace_floats(M, C) floats.  (bias gradients) */
int mmseg_colsum(const float* x, float* out, float* ws, long M, int C, float scale, int accumulate, void* stream);
/* keras MaxPooling2D(2) (models/unet.py:39-51, layers/stn_spline.py:108,111) */
int mmseg_maxpool2_fwd(const float* x, float* y, int B, int H, int W, int C, void* stream);
int mmseg_maxpool2_bwd(const float* x, const float* y, const float* dy, float* dx, int B, int H, int W, int C, void* stream);
/* gradient of keras UpSampling2D(2): dx[B,H,W,C] = 2x2 block sums of dy[B,2H,2W,C] */
int mmseg_upsample2_bwd(const float* dy, float* dx, int B, int H, int W, int C, void* stream);
/* keras UpSampling2D(2) standalone (decoder.py:70-80): y[B,2H,2W,C] */
int mmseg_upsample2_fwd(const float* x, float* y, int B, int H, int W, int C, void* stream);
/* tf.image.resize_nearest_neighbor for an integer down-sampling factor f (layers/spade.py:36-38): y[B,Ho,Wo,C] */
int mmseg_subsample_fwd(const float* x, float* y, int B, int Ho, int Wo, int C, int f, void* stream);
int mmseg_subsample_bwd(const float* dy, float* dx, int B, int Ho, int Wo, int C, int f, void* stream);
/* channel softmax; s (optional) = round-half-even(p): conv_anatomy softmax + layers/rounding.py:23-42 */
int mmseg_softmax_fwd(const float* x, float* p, float* s, long npix, int C, void* stream);
int mmseg_softmax_bwd(const float* dy, const float* p, float* dx, long npix, int C, void* stream);
/* layers/film.py:26-36 fused with the LeakyReLU and residual Add of decoder.py:50-53: y = leaky(x*g+b) [+ res] */
int mmseg_film_fwd(const float* x, const float* gamma, const float* beta, const float* res, float* y, int B, long HW, int C,
                   float alpha, void* stream);
int mmseg_film_bwd_workspace(int B, int C);
int mmseg_film_bwd(const float* du, const float* x, const float* gamma, const float* beta, float* dx, float* dgamma, float* dbeta,
                   float* ws, int B, long HW, int C, float alpha, void* stream);
/* keras Maximum (model_components/anatomy_fuser.py:33); gradient ties go to the first argument (tf.maximum) */
int mmseg_maximum_fwd(const float* a, const float* b, float* y, long n, void* stream);
int mmseg_maximum_bwd(const float* a, const float* b, const float* dy, float* da, float* db, long n, void* stream);
/* Lambda x[..., c0:c0+Cs] (models/dafnet.py:187) */
int mmseg_slice_fwd(const float* x, float* y, long M, int C, int c0, int Cs, void* stream);
int mmseg_slice_bwd(const float* dy, float* dx, long M, int C, int c0, int Cs, void* stream);
/* utils/sdnet_utils.py:9-21 (eps explicit) + costs.py:186-189 */
int mmseg_sampling_kl_fwd(const float* mu, const float* lv, const float* eps, float* z, float* kl, int B, int Z, void* stream);
int mmseg_sampling_kl_bwd(const float* mu, const float* lv, const float* eps, const float* dz, const float* dkl, float* dmu,
                          float* dlv, int B, int Z, void* stream);

/* ---- normalisation (csrc/norm.hip) ------------------------------------------------------------------- */
int mmseg_norm_workspace_floats(int C);
/* keras BatchNormalization, training: batch statistics; updates moving stats in place when non-null */
int mmseg_bn_stats(const float* x, const float* gamma, const float* beta, float* mean, float* invstd, float* scale, float* shift,
                   float* mov_mean, float* mov_var, float* ws, long M, int C, float eps, float momentum, void* stream);
int mmseg_bn_infer_prep(const float* gamma, const float* beta, const float* mov_mean, const float* mov_var, float* scale, float* shift,
                        int C, float eps, void* stream);
/* the same folded behind a convolution with bias conv_bias (may be NULL): scale = gamma / sqrt(var + eps),
 * shift = beta - mean * scale + conv_bias * scale  ->  bn(conv + conv_bias) = conv * scale + shift */
int mmseg_bn_infer_fold(const float* gamma, const float* beta, const float* mov_mean, const float* mov_var, const float* conv_bias,
                        float* scale, float* shift, int C, float eps, void* stream);
int mmseg_bn_apply(const float* x, const float* scale, const float* shift, float* y, long M, int C, int relu, void* stream);
int mmseg_bn_bwd(const float* dy, const float* y, const float* x, const float* gamma, const float* mean, const float* invstd,
                 float* dx, float* dgamma, float* dbeta, float* coef, float* ws, long M, int C, int relu, int accumulate, void* stream);
/* the same without the saved output y: the ReLU mask is recomputed from x with the forward pass's scale / shift (mmseg_bn_stats),
 * bit-identical to the mask of mmseg_bn_apply's output -- 5 tensor passes over HBM instead of 7 */
int mmseg_bn_bwd_x(const float* dy, const float* x, const float* scale, const float* shift, const float* gamma, const float* mean,
                   const float* invstd, float* dx, float* dgamma, float* dbeta, float* coef, float* ws, long M, int C, int relu,
                   int accumulate, void* stream);
/* Synchronised BatchNorm over data-parallel ranks (build-defined option conf.sync_bn; the reference is single device, SURVEY 8e iii):
 * the two halves of mmseg_bn_stats / mmseg_bn_bwd, the caller exchanging [2][C] floats in between (all-gather of (mean, biased
 * variance) in rank order; all-reduce(sum) of (sum g, sum g*xhat)).  stat2 / sums: [2][C]; gathered: [R][2][C]; coef: [3][C]. */
int mmseg_bn_stats_local(const float* x, float* stat2, float* ws, long M, int C, void* stream);
int mmseg_bn_stats_combine(const float* gathered, int R, const float* gamma, const float* beta, float* mean, float* invstd, float* scale,
                           float* shift, float* mov_mean, float* mov_var, long M_total, int C, float eps, float momentum, void* stream);
int mmseg_bn_bwd_sums(const float* dy, const float* y, const float* x, const float* mean, const float* invstd, float* sums, float* ws,
                      long M, int C, int relu, void* stream);
int mmseg_bn_bwd_finish(const float* sums_local, const float* sums_global, const float* gamma, const float* mean, const float* invstd,
                        float* dgamma, float* dbeta, float* coef, int C, long M_total, int accumulate, void* stream);
int mmseg_bn_bwd_apply(const float* dy, const float* y, const float* x, const float* coef, float* dx, long M, int C, int relu, void* stream);
/* ---- reduced-precision STORAGE of the convolutional trunk (csrc/act16.hip; build-defined, BASELINE configs #3 / #5): BatchNorm, 2x2
 *      max pooling and the gradient of nearest x2 up-sampling on tensors stored as fp32 or a 16-bit type, per tensor: element code
 *      h = 0 fp32, 1 bf16, 2 fp16.  Arithmetic, statistics and reductions are fp32 as in the fp32 entry points.  C % 64 == 0 for
 *      the statistics / backward passes.  mmseg_bn_bwd_t: dy, y carry hy; x, dx carry hx; dgamma / dbeta may be NULL. ---- */
int mmseg_bn_stats_t(const void* x, const float* gamma, const float* beta, float* mean, float* invstd, float* scale, float* shift,
                     float* mov_mean, float* mov_var, float* ws, long M, int C, float eps, float momentum, int hx, void* stream);
int mmseg_bn_apply_t(const void* x, const float* scale, const float* shift, void* y, long M, int C, int relu, int hx, int hy, void* stream);
int mmseg_bn_bwd_t(const void* dy, const void* y, const void* x, const float* gamma, const float* mean, const float* invstd, void* dx,
                   float* dgamma, float* dbeta, float* coef, float* ws, long M, int C, int relu, int accumulate, int hx, int hy, void* stream);
int mmseg_maxpool2_fwd_t(const void* x, void* y, int B, int H, int W, int C, int h, void* stream);
int mmseg_maxpool2_bwd_t(const void* x, const void* y, const void* dy, void* dx, int B, int H, int W, int C, int h, void* stream);
int mmseg_upsample2_bwd_t(const void* dy, void* dx, int B, int H, int W, int C, int h, void* stream);
/* dx = gradient of MaxPooling2D(2) + add (add may be NULL): the down-path activation of models/unet.py:39-51 feeds the pooling AND the
 * skip Concatenate (models/unet.py:69-84); both gradients in one pass instead of the autograd engine's separate addition */
int mmseg_maxpool2_bwd_add_t(const void* x, const void* y, const void* dy, const void* add, void* dx, int B, int H, int W, int C, int h,
                             void* stream);
/* out = p0 + ... + p(n-1), 1 <= n <= 8, tensors of `numel` elements with element code h: the sum of the gradients of a tensor with
 * several consumers (keras / TF add them inside the backward graph, e.g. the anatomy s feeding 7 layers in models/dafnet.py:163-222) */
int mmseg_sum_n_t(const void* p0, const void* p1, const void* p2, const void* p3, const void* p4, const void* p5, const void* p6,
                  const void* p7, int n, void* out, long numel, int h, void* stream);
/* out = concatenation of n <= 8 equally sized contiguous parts of `words` 4-byte words along the leading axis; NULL part = zeros
 * (batched calls of per-sample components: the 6 decodings / 4 D_Mask passes of models/dafnet.py:187-215, the pools of
 * model_executors/dafnet_executor.py:524-570) */
int mmseg_cat_words(const void* p0, const void* p1, const void* p2, const void* p3, const void* p4, const void* p5, const void* p6,
                    const void* p7, int n, void* out, long words, void* stream);
/* out[r] = src[idx[r]], rows of `words` 4-byte words (utils/data_utils.py sample(): np.random.choice rows of a fake pool) */
int mmseg_gather_rows(const void* src, const long long* idx, void* out, int rows, long words, int src_rows, void* stream);
/* base_executor.py:83-87 add_residual on the device: out[M][C+1] = masks + background channel */
int mmseg_add_residual(const float* msk, float* out, long M, int C, void* stream);
/* dx = dy * act'(y) of a convolution's fused activation (y = its output), tensors stored with element code h (0 fp32, 1 bf16, 2 fp16);
 * with bias_grad != NULL also the bias gradient bias_grad[c] (+)= sum_m dx[m][c] in the same pass (C % 64 == 0, ws =
 * mmseg_colsum_workspace_floats(M, C) floats) -- one pass instead of mmseg_act_bwd + mmseg_colsum */
int mmseg_act_bwd_bias_t(const void* dy, const void* y, void* dx, float* bias_grad, float* ws, long M, int C, int act, float alpha,
                         int accumulate, int h, void* stream);
/* out[B*H*W][96] (16-bit, code hy) = im2col of x[B,H,W,8] (code hx) for a 3x3 stride-1 'same' convolution: K index = tap * 8 + channel,
 * columns 72..95 zero.  A 1x1 fast-path convolution over it (Cin = 96, weights = the Keras kernel read as [72][Cout] + 24 zero rows)
 * is the reduced-precision form of the SPADE units' 8 -> 128 convolution (layers/spade.py:28) */
int mmseg_im2col8_t(const void* x, void* out, int B, int H, int W, int hx, int hy, void* stream);
/* Conv2D(Cout, 3, padding='same') of an 8-channel tensor in the 16-bit modes in ONE launch: the shared convolution of a SPADE unit (layers/spade.py:27:
 * anatomy -> 128 hidden channels + ReLU) and the segmentor's first layer (model_components/segmentor.py:16).  x fp32 (hx 0) or the mode's 16-bit type, w the
 * Keras kernel [3,3,8,Cout], y fp32 (hy 0) or 16-bit; W % 32 == 0, Cout % 4 == 0 (% 8 for a 16-bit y); act 0 none, 1 ReLU, 2 LeakyReLU(alpha).  Replaces mmseg_im2col8_t + a 1x1 mmseg_conv2d_fwd_t. */
int mmseg_conv8h_fwd_t(const void* x, const float* w, const float* bias, void* y, int B, int H, int W, int Cout, int act, float alpha,
                       int hx, int hy, void* stream);
/* keras_contrib InstanceNormalization(axis=None) fused with SPADE_COND and LeakyReLU (layers/spade.py:7-33,51-54) */
int mmseg_in_workspace_floats(int B);
int mmseg_instnorm_spade_fwd(const float* x, const float* gamma, const float* beta, float* y, float* stat, float* ws, int B,
                             long per_sample, float eps, float act_alpha, void* stream);
int mmseg_instnorm_spade_bwd(const float* dy, const float* x, const float* stat, const float* gamma, const float* beta, float* dx,
                             float* dgamma, float* dbeta, float* dxn, float* ws, int B, long per_sample, float eps, float act_alpha,
                             void* stream);
/* The same with gamma and beta as the two halves of ONE tensor gb [B*H*W][2C] (gamma = channels [0, C), beta = [C, 2C)) -- the
 * output of a SPADE unit's gamma and beta convolutions run as one convolution (layers/spade.py:30-33 of the reference computes
 * Conv2D(f)(a) twice on the same 128-channel tensor); dgb has the layout of gb.  per_sample = H*W*C, C % 4 == 0. */
int mmseg_instnorm_spade_fwd_gb(const float* x, const float* gb, float* y, float* stat, float* ws, int B, long per_sample, int C, float eps,
                                float act_alpha, void* stream);
int mmseg_instnorm_spade_bwd_gb(const float* dy, const float* x, const float* stat, const float* gb, float* dx, float* dgb, float* dxn,
                                float* ws, int B, long per_sample, int C, float eps, float act_alpha, void* stream);
/* the same with gb / dgb stored with element code h and the output y / its gradient dy with element code hy (0 fp32, 1 bf16, 2 fp16):
 * with 16-bit activation storage the fused gamma / beta tensor, the modulated output that feeds the next convolution and their
 * gradients live in HBM in the 16-bit type (x and dx stay fp32) */
int mmseg_instnorm_spade_fwd_gb_t(const float* x, const void* gb, void* y, float* stat, float* ws, int B, long per_sample, int C, float eps,
                                  float act_alpha, int h, int hy, void* stream);
int mmseg_instnorm_spade_bwd_gb_t(const void* dy, const float* x, const float* stat, const void* gb, float* dx, void* dgb, float* dxn,
                                  float* ws, int B, long per_sample, int C, float eps, float act_alpha, int h, int hy, void* stream);
/* out[c] (+)= column sums of a tensor stored with element code h (C % 64 == 0; ws: mmseg_colsum_workspace_floats(M, C) floats) */
int mmseg_colsum_t(const void* x, float* out, float* ws, long M, int C, int accumulate, int h, void* stream);
/* out[m] = (a[m] | b[m]) for M rows of Ca and Cb floats; da[m] += src[m][0:Ca], db[m] += src[m][Ca:]: the fused operand of two
 * convolutions that share their input, and its gradient back into the two parameters */
int mmseg_concat_cols(const float* a, const float* b, float* out, long M, int Ca, int Cb, void* stream);
int mmseg_split_cols_acc(const float* src, float* da, float* db, long M, int Ca, int Cb, void* stream);


/* ---- keras Dense for rows <= 32 (csrc/dense.hip) ----------------------------------------------------- */
long mmseg_dense_workspace_floats(int R, int K, int N);
int mmseg_dense_fwd(const float* x, const float* w, const float* bias, float* y, float* ws, int R, int K, int N, int act,
                    float alpha, void* stream);
int mmseg_dense_dgrad(const float* dy, const float* w, float* dx, int R, int K, int N, void* stream);
int mmseg_dense_wgrad(const float* x, const float* dy, float* dw, int R, int K, int N, int accumulate, void* stream);

/* ---- thin-plate-spline warp (csrc/tps.hip): layers/stn_spline.py:36-67 + interpolate_spline.py + resampler --- */
int mmseg_tps_workspace_floats(int B);
int mmseg_tps_warp_fwd(const float* vol, const float* theta, const float* Mb, float* out, float* loc, int B, int H, int W, int C,
                       void* stream);
long mmseg_tps_scatter_workspace_floats(int B, int H, int W, int C);
/* acc: mmseg_tps_scatter_workspace_floats floats (8-byte aligned) -- 64-bit fixed-point accumulators of the d_vol scatter, which
 * make it independent of the order of the atomics (bitwise reproducible) */
int mmseg_tps_warp_bwd(const float* vol, const float* loc, const float* Mb, const float* dout, float* dvol, float* dtheta, float* dloc,
                       float* ws, float* acc, int B, int H, int W, int C, void* stream);

/* ---- batch gather + affine (rotation) augmentation (csrc/augment.hip): keras ImageDataGenerator(rotation_range=20)
 *      .flow of model_executors/base_executor.py:37-78,103-110 = scipy affine_transform(order=1, mode='nearest') ---- */
int mmseg_affine_gather(const float* data, const int* rows, const float* mat, float* out, int B, int H, int W, int C, int order,
                        void* stream);

/* ---- losses (csrc/loss.hip): costs.py:43-85,129-136, keras mae/mse, costs.ypred ---------------------------- */
int mmseg_segloss_workspace_floats(int B);
int mmseg_segloss_stats_floats(int B);
int mmseg_segloss_coef_floats(int B, int C);
int mmseg_segloss_class_offset(int B);
int mmseg_segloss_stats(const float* pred, const float* target, float* stats, float* ws, int B, long HW, int C, int nm, void* stream);
/* n_pix_global normalises the loss value (pixels of the whole data-parallel batch); n_pix_grad normalises the gradient
 * coefficients (the LOCAL pixel count: the gradient all-reduce takes the mean over ranks).  Equal on one device. */
int mmseg_segloss_finalize(const float* stats, float* loss, float* coef, int B, int C, float n_pix_global, float n_pix_grad,
                           float lambda_bce, void* stream);
int mmseg_segloss_grad(const float* pred, const float* target, const float* coef, float* dpred, int B, long HW, int C, int nm,
                       float scale, int use_bce, void* stream);
/* ---- in-graph per-sample loss terms of the automated-pairing trainers (csrc/pairloss.hip): model_components/balancer.py:33-38
 *      (pair dice), costs.py:24-26 (mae_single_input), costs.py:43-49,88-108,138-143 (combined dice + swapped-argument
 *      per-batch cross-entropy), keras Multiply/Add of models/dafnet.py:290-312 (row dot) ------------------------------ */
int mmseg_pairloss_workspace_floats(int B);
int mmseg_pair_dice_fwd(const float* a, const float* b, float* stats, float* out, int ldo, float* ws, int B, long per_sample,
                        void* stream);
int mmseg_pair_dice_bwd(const float* a, const float* b, const float* stats, const float* g, int ldg, float* da, float* db,
                        int accumulate_a, int B, long per_sample, void* stream);
int mmseg_row_mae_fwd(const float* x, const float* y, float* out, float* ws, int B, long per_sample, void* stream);
int mmseg_row_mae_bwd(const float* x, const float* y, const float* g, float* dy, int B, long per_sample, void* stream);
int mmseg_segpb_stats_floats(int B);
int mmseg_segpb_class_offset(int B);
int mmseg_segpb_stats(const float* pred, const float* target, float* stats, float* ws, int B, long HW, int C, int nm, void* stream);
int mmseg_segpb_loss(const float* stats, float* loss, int B, long HW, int C, float lambda_bce, void* stream);
int mmseg_segpb_classgrad(const float* stats, const float* g, float* A, int B, void* stream);
int mmseg_segpb_grad(const float* target, const float* stats, const float* g, const float* A, float* dpred, int B, long HW, int C,
                     int nm, float lambda_bce, void* stream);
int mmseg_rowdot_fwd(const float* w, const float* l, float* out, int B, int J, void* stream);
int mmseg_rowdot_bwd(const float* w, const float* l, const float* g, float* dw, float* dl, int B, int J, void* stream);

int mmseg_diffloss_workspace_floats(void);
/* mode 0: mean|p-t|, 1: mean (p-t)^2, 2: mean p ; t == NULL -> constant target tconst */
int mmseg_diffloss(const float* p, const float* t, float tconst, long n, int mode, float* loss, float* ws, void* stream);
int mmseg_diffloss_grad(const float* p, const float* t, float tconst, long n, int mode, float scale, float* dp, void* stream);

/* ---- optimiser / regulariser (csrc/optim.hip) ---------------------------------------------------------- */
/* Keras 2.1.6 Adam step over a flat arena (models/dafnet.py:93,114,155,161) */
int mmseg_adam(float* p, const float* g, float* m, float* v, long n, float lr_t, float b1, float b2, float eps, void* stream);
/* the same update with lr_t read from device memory (one float) -- the form recorded when a training step is captured into a hipGraph */
int mmseg_adam_p(float* p, const float* g, float* m, float* v, long n, const float* lr_t, float b1, float b2, float eps, void* stream);
/* layers/spectralnorm.py:199-239 */
long mmseg_spectral_workspace_floats(int K, int N);
int mmseg_spectral_fwd(const float* w, const float* u0, float* loss, float* sgn, float* ws, int K, int N, float alpha, void* stream);
int mmseg_spectral_grad(const float* w, const float* sgn, float scale, long n, float* dw, void* stream);
/* the penalties of up to 4 matrices (the 4 down-sample blocks of one discriminator, models/discriminator.py:24-41) in one batch
 * of launches: loss[i], sgn[i]; ws = the per-matrix workspaces back to back; and their gradients accumulated into dw_i */
int mmseg_spectral_fwd4(const float* w0, const float* w1, const float* w2, const float* w3, const float* u0, const float* u1,
                        const float* u2, const float* u3, float* loss, float* sgn, float* ws, int n, int K0, int N0, int K1, int N1,
                        int K2, int N2, int K3, int N3, float alpha, void* stream);
int mmseg_spectral_grad4(const float* w0, const float* w1, const float* w2, const float* w3, const float* sgn, float* dw0, float* dw1,
                         float* dw2, float* dw3, int n, long n0, long n1, long n2, long n3, float scale, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MMSEG_HIP_H */

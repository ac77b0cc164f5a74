/* segmentron_hip.h — C ABI of libsegmentron_hip.so (gfx950 / MI355X).
 *
 * Drop-in boundary for SegmenTron's dense-convolution hot path.  The reference has NO native
 * boundary for this path: all arithmetic is delegated to torch.nn / torch.nn.functional
 * (ATen -> MIOpen/oneDNN), see SURVEY.md §1 "Op layer" and §8(b).  Each entry point below is what
 * a binding for the corresponding reference call site would call instead of the ATen op; the
 * reference-side binding (ctypes stub) is shown in INTEGRATION.md.  The reference's only native
 * code (segmentron/modules/csrc/vision.cpp:6-11, CCNet criss-cross attention) is out of scope.
 *
 * Conventions
 *   - plain pointers + sizes, no torch types; every pointer is DEVICE memory unless noted
 *   - activations are NHWC: element (n,h,w,c) of a tensor with row pitch `ld` (elements) lives at
 *     base + ((n*H + h)*W + w)*ld + c ; `ld >= C` lets an op read/write a channel slice of a wider
 *     (concatenation) buffer.  C and ld must be multiples of 16 bytes / sizeof(element).
 *   - dtype: 0 = float32, 1 = bfloat16 (element type of activations / packed weights);
 *     BatchNorm parameters, statistics and all accumulation are float32 / float64
 *   - pro_mode ("prologue"): the producer's BatchNorm(+ReLU) applied on the fly to an input:
 *     0 none, 1 relu(x), 2 x*scale[c]+shift[c], 3 relu(x*scale[c]+shift[c]); +4 = ReLU6 clamp
 *     (5 = relu6(x), 7 = relu6(x*scale[c]+shift[c]))
 *   - `stream` is a hipStream_t; all launches are asynchronous on it; no host synchronisation,
 *     no allocation; re-entrant and graph-capturable; no process-wide state.
 *   - return value 0 = ok; otherwise seg_last_error() (thread-local) describes the failure.
 *     Never aborts.
 */
#ifndef SEGMENTRON_HIP_H
#define SEGMENTRON_HIP_H
#ifdef __cplusplus
extern "C" {
#endif

const char* seg_last_error(void);
int seg_version(void);

/* ---- nn.Conv2d, groups=1 (1x1 any stride, dense KxK): implicit GEMM on MFMA ----------------
 * Replaces F.conv2d for segmentron/modules/basic.py:42,69; segmentron/modules/module.py:45,57;
 * segmentron/models/backbones/xception.py:21,77,81; segmentron/models/deeplabv3_plus.py:64.
 * y[n,ho,wo,o] = bias[o] + sum_{kh,kw,c} act(x[n,ho*stride-pad+kh*dil, wo*stride-pad+kw*dil, c])
 *                                        * w[o][(kh*KW+kw)*C + c]
 * w is packed [O][KH*KW*C] in `dtype`.  stat_partial (nullable): [seg_conv_gemm_tiles_m][2][O]
 * fp32 per-tile (sum, sum of squares) of the pre-bias fp32 results, for the following BatchNorm.
 * out_s != 1 scatters output pixel (n,ho,wo) to row ((n*out_H + ho*out_s)*out_W + wo*out_s)
 * (data-gradient of a strided 1x1 conv; the caller zero-fills y first).
 * The same entry point computes data gradients: pass dy as x and the transposed/flipped weights.
 * tconv = 1: transposed-stride gather — x is dy [N,Hi,Wi,C=O_fwd], (Ho,Wo) the size of dx, w packed
 * [C_fwd][KH*KW*O_fwd] (not flipped): dx[h,w] = sum dy[(h+pad-kh*dil)/stride, ...] W (exact
 * divisions only) = data gradient of a strided KxK convolution (resnet.py:52-53, hrnet.py:195-204).
 * ep_x (nullable, addressed like y with pitch ldep) fuses the BatchNorm-backward correction of a
 * folded layer into the store: y = acc - ep_c0[o] - ep_c1[o]*ep_x[row][o] (see seg_fold_*). */
int seg_conv_gemm_fwd(int dtype, const void* x, long ldx, int N, int Hi, int Wi, int C,
                      const void* w, int O, int KH, int KW, int stride, int pad, int dil,
                      int pro_mode, const float* pro_scale, const float* pro_shift,
                      const float* bias, void* y, long ldy, int Ho, int Wo, int out_H, int out_W,
                      int out_s, float* stat_partial, const void* ep_x, long ldep,
                      const float* ep_c0, const float* ep_c1, int tconv, void* stream);
int seg_conv_gemm_tiles_m(int N, int Ho, int Wo);

/* Weight gradient of the same convolution (autograd's conv2d backward wrt weight):
 * partial[s][o][k] for s < splits (fp32); sum over s with seg_colsum gives dW[O][KH*KW*C].
 * The forward prologue is re-applied to x (the activated input is never stored). */
int seg_conv_gemm_wgrad(int dtype, const void* x, long ldx, int N, int Hi, int Wi, int C,
                        const void* dy, long lddy, int Ho, int Wo, int O, int KH, int KW,
                        int stride, int pad, int dil, int pro_mode, const float* pro_scale,
                        const float* pro_shift, float* partial, int splits, void* stream);
/* rows of the [rows][2][O] statistics buffer seg_conv_gemm_fwd writes for this geometry */
int seg_conv_gemm_stat_rows(int dtype, int N, int Ho, int Wo, int C, int O, int KH, int KW,
                            int stride, int pad, int dil, int tconv, int has_bias, int pro_mode);
/* splits to allocate `partial` for (same geometry arguments as the seg_conv_gemm_wgrad call:
 * plain 1x1 convolutions run on the direct-to-LDS kernel, which wants ~one block per CU; the
 * 3x3 stride-1 stems with 32 input channels on persistent blocks, one partial each) */
int seg_conv_gemm_wgrad_splits(int dtype, int N, int Ho, int Wo, int C, int O, int KH, int KW,
                               int stride, int pad, int dil, int pro_mode);

/* ---- nn.Conv2d, groups=C, 3x3, padding=dilation (depthwise) --------------------------------
 * Replaces segmentron/modules/basic.py:38-40 (SeparableConv2d.depthwise), :152-153.
 * w9c: fp32 taps; w_layout 0 = [9][C] tap-major, bit 0 = torch's own [C,1,3,3] (no repacking,
 * stride 1 / dilation <= 2 only), bit 1 = taps read reversed (stride-1 data gradient = forward
 * correlation with flipped taps).  mode 0: forward (x -> y).  mode 1: data gradient
 * (x = dy with geometry N,Hi,Wi; y = dx with geometry Ho,Wo; same w9c, stride, dil).
 * stat_partial (forward only, nullable): [grid_y][2][C].  grid_y from seg_dwconv_grid_y. */
int seg_dwconv3x3(int dtype, int mode, const void* x, long ldx, int N, int Hi, int Wi, int C,
                  const float* w9c, int w_layout, int stride, int dil, int pro_mode,
                  const float* pro_scale, const float* pro_shift, void* y, long ldy, int Ho, int Wo,
                  float* stat_partial, int grid_y, void* stream);
/* partial rows (= persistent blocks per channel block) of one depthwise launch.
 * kind: 0 forward / data gradient, 1 fused backward, 2 weight gradient */
int seg_dwconv_grid_y(int dtype, int C, int N, int Ho, int Wo, int stride, int dil, int kind);
/* partial: fp32 [grid_y][9][C]; column-sum gives dW[9][C];
 * seg_dwconv3x3_wgrad_finalize reduces it straight into torch's [C,1,3,3] layout. */
int seg_dwconv3x3_wgrad_finalize(const float* partial, int R, int C, float* dw_c9, void* stream);
int seg_dwconv3x3_wgrad(int dtype, const void* x, long ldx, int N, int Hi, int Wi, int C,
                        const void* dy, long lddy, int Ho, int Wo, int stride, int dil,
                        int pro_mode, const float* pro_scale, const float* pro_shift,
                        float* partial, int grid_y, void* stream);

/* Fused stride-1 backward of the depthwise conv in ONE pass over (dy, x):
 *   g[p]  = relu_mask(x[p]) * sum_k dy[p - d_k] * w9c[k]        (gradient wrt act(x), masked)
 *   partial_w  [grid_y][9][C] : weight-gradient partials (sum rows -> dW[9][C])
 *   partial_bn [grid_y][2][C] : (sum g, sum g*x_raw) for seg_bn_bwd_finalize_p (nullable)
 * x is the forward input (raw tensor + prologue), w9c the FORWARD taps (w_layout as above, bit 1
 * unused).  grid_y from seg_dwconv_grid_y(dtype, C, N, H, W, 1, dil, 1).  dil <= 2: LDS-tiled
 * kernel with a tile software pipeline (csrc/dwconv_tiled.hip); wider dilations: strip kernel. */
int seg_dwconv3x3_bwd_fused(int dtype, const void* dy, long lddy, const void* x, long ldx, int N,
                            int H, int W, int C, const float* w9c, int w_layout, int dil,
                            int pro_mode, const float* pro_scale, const float* pro_shift, void* g,
                            long ldg, float* partial_w, float* partial_bn, int grid_y,
                            void* stream);
/* The same with `res` ([N,H,W,C], pitch ldr, element type of g) ADDED to the masked data gradient
 * in the store path: g = relu_mask(x) * dgrad + res — the second gradient of a forked activation
 * (an Xception block input feeds the residual sum and the first separable conv,
 * segmentron/models/backbones/xception.py:36-42), which autograd would add in a separate
 * element-wise pass.  LDS-tiled kernel only: stride 1, dilation 1 (seg_..._add_ok(dil) != 0). */
int seg_dwconv3x3_bwd_fused_add_ok(int dil);
int seg_dwconv3x3_bwd_fused_add(int dtype, const void* dy, long lddy, const void* x, long ldx, int N,
                                int H, int W, int C, const float* w9c, int w_layout, int pro_mode,
                                const float* pro_scale, const float* pro_shift, const void* res,
                                long ldr, void* g, long ldg, float* partial_w, float* partial_bn,
                                int grid_y, void* stream);

/* ---- nn.BatchNorm2d / nn.SyncBatchNorm (train + eval, forward + backward) -------------------
 * Replaces F.batch_norm behind every `bn*` module (segmentron/modules/basic.py:41,43,70;
 * segmentron/modules/module.py:46,54,58; xception.py:22,78,82) and torch's SyncBatchNorm
 * collectives (tools/train.py:76): the caller all-reduces the 2C doubles between seg_colsum
 * and the finalize call.  eps / momentum are read at call time. */
/* out[l] = sum_r in[r][l]  (in: fp32 [R][L]); ws: >= 64*L doubles (needed when R > 128). */
int seg_colsum(const float* in, long R, int L, double* out_d, float* out_f, double* ws,
               void* stream);
/* float64 column sums with the local element count appended (out_d[L] = count): the
 * SyncBatchNorm forward message of one BatchNorm, assembled by one launch. */
int seg_colsum_count(const float* in, long R, int L, double* out_d, double count, double* ws,
                     void* stream);
/* sums = [sum x (C), sum x^2 (C)] over `count` samples.  Writes mean, invstd (biased var),
 * scale = gamma*invstd, shift = beta - mean*scale; updates running stats (nullable) with the
 * unbiased variance and `momentum`.  mean_offset (nullable, [C]) is added to the batch mean for the
 * running_mean update only: the per-channel constant a folded convolution (seg_fold_weights) leaves
 * out of its stored output because the following BatchNorm cancels it.
 * count_dev (nullable, here and in the backward finalizes): a DEVICE double that overrides
 * `count` — the SyncBatchNorm caller all-reduces [sums | local count] as one message and passes
 * &buf[2C], so unequal per-rank shards are exact without a host read-back. */
int seg_bn_finalize(const double* sums, double count, const double* count_dev,
                    const float* gamma, const float* beta, float eps, float momentum, float* running_mean, float* running_var,
                    float* mean, float* invstd, float* scale, float* shift, int C,
                    const float* mean_offset, void* stream);
/* Single-process BatchNorm: the same directly from the [R][2][C] fp32 partial rows the conv /
 * reduce kernels emit (column sum fused in, fp64).  ws: >= 128*C doubles, used when R > 1024. */
int seg_bn_finalize_p(const float* partial, long R, double count, const float* gamma,
                      const float* beta, float eps, float momentum, float* running_mean,
                      float* running_var, float* mean, float* invstd, float* scale, float* shift,
                      int C, const float* mean_offset, double* ws, void* stream);
/* The same for a BatchNorm over FEW samples (M = N*H*W <= 4096 rows of the stored NHWC tensor y,
 * row pitch ldy): two-pass statistics (mean, then sum (x - mean)^2) straight from y — the
 * single-pass partial sums lose mean^2 / var digits, catastrophic for the 2-sample BatchNorm of
 * the ASPP image-pooling branch (module.py:52-64) and PSP's pyramid bins (module.py:89-97). */
int seg_bn_finalize_small(int dtype, const void* y, long ldy, long M, int C, const float* gamma,
                          const float* beta, float eps, float momentum, float* running_mean,
                          float* running_var, float* mean, float* invstd, float* scale,
                          float* shift, const float* mean_offset, void* stream);
/* ... under SyncBatchNorm (torch.nn.SyncBatchNorm / the reference's NaiveSyncBatchNorm,
 * segmentron/modules/batch_norm.py:135-176): every rank takes ITS rows' two-pass moments and the
 * ranks merge them as float64 sums (n*mean_r, M2_r + n*mean_r^2, n).  seg_bn_moments_small writes
 * those 2C + 1 doubles for a torch.distributed all-reduce followed by seg_bn_finalize;
 * seg_bn_finalize_small_sync merges them through the peer mailbox inside the launch (count_out =
 * the global element count, as seg_bn_finalize_p_sync). */
int seg_bn_moments_small(int dtype, const void* y, long ldy, long M, int C, double* moments,
                         void* stream);
int seg_bn_finalize_small_sync(void* p2p, int dtype, const void* y, long ldy, long M, int C,
                               const float* gamma, const float* beta, float eps, float momentum,
                               float* running_mean, float* running_var, float* mean, float* invstd,
                               float* scale, float* shift, const float* mean_offset,
                               double* count_out, void* stream);
/* y = xs[0] + ... + xs[n-1] (2 <= n <= 8 NHWC operands of M rows x C channels, row pitches lds[],
 * host arrays of device pointers / pitches), fp32 accumulation in index order, one rounding: the
 * gradient of an activation with several consumers (torch autograd: n-1 `add` launches). */
int seg_sum_n(int dtype, int n, const void* const* xs, const long* lds, void* y, long ldy, long M,
              int C, void* stream);
/* eval mode: scale/shift from running statistics. */
int seg_bn_eval_affine(const float* gamma, const float* beta, const float* rm, const float* rv,
                       float eps, float* scale, float* shift, int C, void* stream);
/* ... of n BatchNorms at once (host arrays of device pointers; out[j]: 2*C[j] floats = scale row,
 * shift row; C[j] <= 2048; gamma[j] / beta[j] nullable): one launch per 48 jobs instead of one per
 * BatchNorm of an evaluation-mode forward. */
int seg_bn_eval_affine_multi(int n, const float* const* gamma, const float* const* beta,
                             const float* const* rm, const float* const* rv, const float* eps,
                             float* const* out, const int* C, void* stream);
/* Materialise: y = post_relu?( act_x(x) * chan_mul[n][c] + act_r(r) )  (r, chan_mul nullable).
 * Covers BN+ReLU materialisation, the residual adds of xception.py:40,42 / resnet.py:38,78 and
 * nn.Dropout2d (segmentron/modules/module.py:60; chan_mul = mask/(1-p), rows_per_n = H*W) and
 * nn.Dropout (module.py:21, pspnet.py:52; elem_mul = per-element mask/(1-p) of the element type). */
int seg_bn_apply(int dtype, const void* x, long ldx, int mode_x, const float* sx, const float* tx,
                 const void* r, long ldr, int mode_r, const float* sr, const float* tr,
                 const float* chan_mul, long rows_per_n, const void* elem_mul, long ldm,
                 int post_relu, void* y, long ldy, long M, int C, void* stream);
/* Backward of "act(x) consumed with gradient g": g' = g * chan_mul * relu_mask.
 * reduce  : partial[grid_y][2][C] = (sum g', sum g'*x) per block
 * finalize: dgamma, dbeta and the coefficients c0,c1 of  dx = scale*g' - c0 - c1*x
 * apply   : dx (may alias g).  With mode lacking the affine bit: dx = g' (ReLU backward). */
int seg_bn_bwd_grid_y(int dtype, int C, long M);
int seg_bn_bwd_reduce(int dtype, const void* g, long ldg, const void* x, long ldx, int mode,
                      const float* scale, const float* shift, const float* chan_mul,
                      long rows_per_n, const void* elem_mul, long ldm, long M, int C,
                      float* partial, int grid_y, void* stream);
/* BatchNorm backward of a SMALL tensor (M = N*H*W <= 4096 rows; same layers as
 * seg_bn_finalize_small) in ONE launch with float64 arithmetic: g' = g * chan_mul * elem_mul *
 * relu_mask, dx = gamma*invstd*(g' - mean(g') - xhat*mean(g'*xhat)) (training = 1) or
 * scale*g' (training = 0), dgamma = sum g'*xhat, dbeta = sum g'.  dx may alias g.  Replaces
 * seg_bn_bwd_reduce + seg_bn_bwd_finalize_p + seg_bn_bwd_apply there: with a handful of samples
 * per channel dx is the remainder of cancelling terms, which fp32 `scale*g' - c0 - c1*x` loses. */
int seg_bn_bwd_small(int dtype, const void* g, long ldg, const void* x, long ldx, void* dx,
                     long lddx, long M, int C, int mode, const float* scale, const float* shift,
                     const float* mean, const float* invstd, const float* gamma,
                     const float* chan_mul, long rows_per_n, const void* elem_mul, long ldm,
                     double count, int training, float* dgamma, float* dbeta, void* stream);
int seg_bn_bwd_finalize(const double* sums, double count, const double* count_dev,
                        const float* mean, const float* invstd, const float* gamma, float* dgamma, float* dbeta, float* c0, float* c1,
                        int C, void* stream);
/* seg_bn_bwd_finalize with dgamma / dbeta multiplied by grad_scale (SyncBatchNorm: the sums are
 * global and the data-parallel gradient averaging divides by the world size once more — torch's
 * SyncBatchNorm returns local sums there, torch/nn/modules/_functions.py:150-170). */
int seg_bn_bwd_finalize_s(const double* sums, double count, const double* count_dev,
                          const float* mean, const float* invstd, const float* gamma, float* dgamma,
                          float* dbeta, float* c0, float* c1, int C, double grad_scale, void* stream);
int seg_bn_bwd_finalize_p(const float* partial, long R, double count, const float* mean,
                          const float* invstd, const float* gamma, float* dgamma, float* dbeta,
                          float* c0, float* c1, int C, double* ws, void* stream);
/* Both reductions behind a fused depthwise backward in one launch: seg_bn_bwd_finalize_p on
 * partial_bn [Rb][2][C] (Rb <= 1024) and seg_dwconv3x3_wgrad_finalize on partial_w [Rw][9][C]
 * (replaces two launches per depthwise layer and step; the reference's counterpart is autograd's
 * batch_norm_backward + the depthwise weight-gradient reduction inside cudnn/MIOpen,
 * segmentron/modules/basic.py:38-44). */
int seg_dw_bwd_finalize(const float* partial_bn, int Rb, double count, const float* mean,
                        const float* invstd, const float* gamma, float* dgamma, float* dbeta,
                        float* c0, float* c1, const float* partial_w, int Rw, float* dw_c9, int C,
                        void* stream);
int seg_bn_bwd_apply(int dtype, const void* g, long ldg, const void* x, long ldx, int mode,
                     const float* scale, const float* shift, const float* c0, const float* c1,
                     const float* chan_mul, long rows_per_n, const void* elem_mul, long ldm,
                     void* dx, long lddx, long M, int C, void* stream);

/* ---- nn.GroupNorm(min(32, C), C): the 'GN' choice of cfg.MODEL.BN_TYPE -------------------------
 * Replaces the nn.GroupNorm modules segmentron/modules/batch_norm.py:105-108,129 builds (F.group_norm
 * forward + autograd backward).  Group statistics are per SAMPLE: z = x*a[n][c] + b[n][c] — a
 * per-(sample, channel) affine, materialised (it cannot ride in a per-channel prologue).
 *   seg_gn_moments      partial[n][chunk][2][C] = (sum u, sum u*v) over the chunk's pixels; v null:
 *                       v = u (sum x, sum x^2).  chunks = seg_gn_chunks(HW).
 *   seg_gn_fwd_finalize mean_rstd[n][g][2] and coef[n][3][C] = (gamma*rstd, 0, beta - mean*gamma*rstd)
 *                       (gamma / beta nullable: affine=False), float64 sums, biased variance
 *   seg_gn_bwd_finalize from the moments of (u, v) = (dz, x): coef[n][3][C] of
 *                       dx = k1*dz + k2*x + k3, and contrib[n][2][C] = per-sample terms of
 *                       (dgamma, dbeta) — sum over n with seg_colsum
 *   seg_gn_affine       out = coef[n][0][c]*u + coef[n][1][c]*v + coef[n][2][c]   (v nullable) */
int seg_gn_chunks(long HW);
int seg_gn_moments(int dtype, const void* u, long ldu, const void* v, long ldv, int N, long HW,
                   int C, float* partial, void* stream);
int seg_gn_fwd_finalize(const float* partial, int N, long HW, int C, int G, const float* gamma,
                        const float* beta, double eps, float* mean_rstd, float* coef,
                        void* stream);
int seg_gn_bwd_finalize(const float* partial, int N, long HW, int C, int G,
                        const float* mean_rstd, const float* gamma, float* coef, float* contrib,
                        void* stream);
int seg_gn_affine(int dtype, const void* u, long ldu, const void* v, long ldv, const float* coef,
                  void* out, long ldo, int N, long HW, int C, void* stream);

/* ---- linear BatchNorm folded into a 1x1 convolution -------------------------------------------
 * `relu_first` SeparableConv2d (segmentron/modules/basic.py:46-50) has no non-linearity between
 * bn_depth and the pointwise conv: W (s.*x + t) = (W diag(s)) x + W t.
 * fold_weights : Wp[o][c] = W[o][c]*scale[c] (dtype), WpT = its transpose [C][O] (nullable),
 *                bprime[o] = sum_c W[o][c]*shift[c] (nullable).  W is fp32 [O][C].
 * fold_bwd_*   : from dWp = dY^T x_raw (fp32 [O][C]) and db = colsum(dY) (nullable = 0):
 *                dW, then dgamma/dbeta of the folded BatchNorm and the coefficients of
 *                dx_raw = dY Wp - c0 - c1 x_raw.  dsdt = [ds (C), dt (C)] is all-reduced by the
 *                caller between the two calls under SyncBN. */
int seg_fold_weights(int dtype, const float* W, const float* scale, const float* shift, void* Wp,
                     void* WpT, float* bprime, int O, int C, void* stream);
/* dWp: the weight-gradient split partials [splits][O*C] as written by seg_conv_gemm_wgrad (summed
 * here); dsdt: [seg_fold_bwd_rows(O)][2][C] partial (ds, dt) rows, summed by the finalize. */
int seg_fold_bwd_rows(int O);
int seg_fold_bwd_reduce(const float* W, const float* dWp, int splits, const float* scale,
                        const float* shift, const float* db, float* dW, float* dsdt, int O, int C,
                        void* stream);
int seg_fold_bwd_finalize(const float* dsdt, int rows, double count, const double* count_dev,
                          const float* mean, const float* invstd, const float* gamma, const float* scale, float* dgamma, float* dbeta,
                          float* c0, float* c1, int C, void* stream);
/* the same with dgamma / dbeta multiplied by grad_scale (SyncBatchNorm: 1 / world size) */
int seg_fold_bwd_finalize_s(const float* dsdt, int rows, double count, const double* count_dev,
                            const float* mean, const float* invstd, const float* gamma,
                            const float* scale, float* dgamma, float* dbeta, float* c0, float* c1,
                            int C, double grad_scale, void* stream);

/* ---- pooling --------------------------------------------------------------------------------
 * nn.MaxPool2d(k, stride, pad) (segmentron/models/backbones/resnet.py:119) on a deferred
 * activation; idx = one byte per output element (winning tap), consumed by the backward gather. */
int seg_maxpool_fwd(int dtype, const void* x, long ldx, int N, int Hi, int Wi, int C, int k,
                    int stride, int pad, int pro_mode, const float* pro_scale,
                    const float* pro_shift, void* y, long ldy, int Ho, int Wo, void* idx,
                    void* stream);
int seg_maxpool_bwd(int dtype, void* gx, long ldgx, int N, int Hi, int Wi, int C, int k, int stride,
                    int pad, const void* gy, long ldgy, int Ho, int Wo, const void* idx,
                    void* stream);
/* nn.AdaptiveAvgPool2d(o) (segmentron/modules/module.py:52,89), ATen bins
 * [floor(i*H/o), ceil((i+1)*H/o)).  partial: fp32 [chunks][N*o*o][C] bin SUMS (reduce over chunks
 * with seg_colsum, divide by the bin area). */
int seg_adaptive_avgpool_chunks(int H, int W, int o);
int seg_adaptive_avgpool_partial(int dtype, const void* x, long ldx, int N, int H, int W, int C,
                                 int o, float* partial, int chunks, void* stream);
int seg_adaptive_avgpool_bwd(int dtype, void* gx, long ldgx, int N, int H, int W, int C, int o,
                             const void* gy, long ldgy, void* stream);

/* ---- F.interpolate(mode='bilinear') ---------------------------------------------------------
 * Replaces segmentron/models/deeplabv3_plus.py:39,44,71; segmentron/modules/module.py:64,96;
 * segmentron/models/segbase.py:83.  NHWC -> NHWC with optional prologue and per-(n,c) multiplier. */
int seg_bilinear_fwd(int dtype, const void* x, long ldx, int N, int Hi, int Wi, int C,
                     int pro_mode, const float* pro_scale, const float* pro_shift,
                     const float* chan_mul, void* y, long ldy, int Ho, int Wo, int align_corners,
                     void* stream);
int seg_bilinear_bwd(int dtype, void* gx, long ldgx, int N, int Hi, int Wi, int C, const void* gy,
                     long ldgy, int Ho, int Wo, int align_corners, void* stream);
/* nn.Upsample(scale_factor=2^shift, mode='nearest') fused with the running sum of HRNet's
 * cross-resolution fuse (segmentron/models/backbones/hrnet.py:186,215-229):
 * y = post_relu?( act_x(x) + act_r(r[n, h>>shift, w>>shift]) ); backward of r = block sums. */
int seg_nearest_add(int dtype, const void* x, long ldx, int mode_x, const float* sx,
                    const float* tx, const void* r, long ldr, int mode_r, const float* sr,
                    const float* tr, int shift, int post_relu, void* y, long ldy, int N, int H,
                    int W, int C, void* stream);
int seg_nearest_sum_bwd(int dtype, const void* g, long ldg, int N, int H, int W, int C, int shift,
                        void* gr, long ldgr, void* stream);
/* Model boundary: NHWC logits (C valid channels, pitch ldx) -> NCHW float32 [N,C,Ho,Wo] and back. */
int seg_upsample_to_nchw(int dtype, const void* x, long ldx, int N, int Hi, int Wi, int C,
                         float* out, int Ho, int Wo, int align_corners, void* stream);
int seg_upsample_to_nchw_bwd(int dtype, void* gx, long ldgx, int N, int Hi, int Wi, int C,
                             const float* gy, int Ho, int Wo, int align_corners, void* stream);
/* Input boundary: NCHW float32 image [N,Cin,H,W] -> NHWC [N,H,W,16/sizeof(elem)] zero-padded. */
int seg_nchw_to_nhwc_pad(int dtype, const float* x, int N, int Cin, int H, int W, void* y,
                         void* stream);

/* ---- fused bilinear upsample -> log-softmax -> NLL (ignore_index), mean over valid pixels ------
 * Replaces F.interpolate(mode='bilinear', align_corners) of the head's logits
 * (segmentron/models/deeplabv3_plus.py:44, pspnet.py:34, fcn.py:28) + F.cross_entropy
 * (segmentron/solver/loss.py:16-46: nn.CrossEntropyLoss(ignore_index=-1)) without materialising
 * the [N, C, H, W] float32 logits.  lo: [N, Hi, Wi, ld] logits in `dtype` (C <= 32 classes, ld a
 * vector-padded pitch); target: int64 [N, H, W]; loss_out: float32[2] = (mean loss, 1 / number
 * of valid pixels); ws: >= 2 * seg_upsample_ce_blocks(N, H, W) doubles.  Backward: dlo
 * [N, Hi, Wi, lddlo] in `dtype` (channels >= C written as zeros) = grad_out[0] * dLoss/dlo,
 * up-sampling factors up to 8.1 (output-stride-4 and -8 heads).  Deterministic (fixed-order float64 / gather reductions). */
int seg_upsample_ce_blocks(int N, int H, int W);
int seg_upsample_ce_fwd(int dtype, const void* lo, long ld, int N, int Hi, int Wi, int C,
                        const long* target, int H, int W, long ignore_index, int align_corners,
                        double* ws, float* loss_out, void* stream);
int seg_upsample_ce_bwd(int dtype, const void* lo, long ld, int N, int Hi, int Wi, int C,
                        const long* target, int H, int W, long ignore_index, int align_corners,
                        const float* loss_out, const float* grad_out, void* dlo, long lddlo,
                        void* stream);

/* ---- stride-2 depthwise 3x3 (padding 1, dilation 1): fused backward --------------------------
 * One pass over dy [N,(H+1)/2,(W+1)/2,C] and x [N,H,W,C] (+ prologue): g = masked data gradient,
 * partial_w [grid_y][9][C] (reduce with seg_dwconv3x3_wgrad_finalize), partial_bn [grid_y][2][C]
 * (nullable) = (sum g, sum g*x_raw).  w_c9: torch's [C,1,3,3] fp32.  grid_y from
 * seg_dwconv3x3_s2_grid_y. */
int seg_dwconv3x3_s2_grid_y(int C, int N, int H, int W);
int seg_dwconv3x3_s2_bwd_fused(int dtype, const void* dy, long lddy, const void* x, long ldx, int N,
                               int H, int W, int C, const float* w_c9, int pro_mode,
                               const float* pro_scale, const float* pro_shift, void* g, long ldg,
                               float* partial_w, float* partial_bn, int grid_y, void* stream);

/* ---- torch.optim.SGD(momentum, weight_decay) step over many tensors -------------------------
 * Replaces the optimizer the reference builds in segmentron/solver/optimizer.py:45-50 (dampening
 * 0, no Nesterov): d = g + wd*p; m = first ? d : momentum*m + d; p -= lr*m, fp32.
 * params / grads / bufs: HOST arrays of ntensors device pointers; numel / group: host arrays;
 * lr_dev / wd_dev: DEVICE float arrays indexed by parameter group (a captured graph of the step
 * follows the LR schedule: the host rewrites these floats between replays). */
int seg_sgd_multi_tensor(int ntensors, const void* const* params, const void* const* grads,
                         const void* const* bufs, const long* numel, const int* group,
                         const float* lr_dev, const float* wd_dev, float momentum, int first,
                         void* stream);
/* Multi-tensor weight packing (r04): njobs jobs "fp32 [O][C] contiguous -> dtype [O][C]
 * (transpose = 0) or [C][O] (transpose = 1)" in as few launches as the 32-job / 1024-tile
 * argument block allows: the GEMM operands of the non-folded 1x1 convolutions, re-made once per
 * optimizer step (torch: one cast / transposing copy per tensor and direction).  srcs / dsts / O /
 * C / transpose: HOST arrays. */
int seg_pack_multi(int dtype, int njobs, const void* const* srcs, const void* const* dsts,
                   const int* O, const int* C, const int* transpose, void* stream);

/* ---- pixAcc / mIoU counters ------------------------------------------------------------------
 * Replaces segmentron/utils/score.py:83-113 (batch_pix_accuracy: argmax of the logits TRUNCATED
 * to integers; batch_intersection_union: argmax + three torch.histc on the CPU).  counters:
 * int64 [2 + 3*nclass] = correct, labelled, inter[nclass], pred[nclass], lab[nclass], ACCUMULATED
 * (zero them to start an evaluation).  _nchw takes fp32 NCHW logits; _upsample takes the
 * network's low-resolution NHWC logits and applies the final bilinear resize on the fly with
 * seg_upsample_to_nchw's arithmetic. */
int seg_metric_update_nchw(const float* logits, int N, int C, int H, int W, const long* target,
                           int nclass, long* counters, void* stream);
int seg_metric_update_upsample(int dtype, const void* lo, long ld, int N, int Hi, int Wi, int C,
                               const long* target, int H, int W, int align_corners, int nclass,
                               long* counters, void* stream);

/* ---- criss-cross attention (CCNet) ------------------------------------------------------------
 * Replaces the reference's CUDA extension segmentron/modules/csrc/criss_cross_attention/
 * ca_cuda.cu:8-177 (+ the softmax of cc_attention.py:66), NHWC.  Entry z of pixel (y, x):
 * z < W -> (y, z); else i = z - W, (i < y ? i : i + 1, x).  L = W + H - 1 <= 512.
 * att / de: fp32 [N, H, W, L].
 *   seg_cca_attention     : att = softmax_z(q[p] . k[partner(p, z)])
 *   seg_cca_map           : out = gamma * sum_z w b[partner] (+ res); transposed = 1 sums over
 *                           the pixels attending TO p (dv, dk); raw (nullable) = the plain sum
 *   seg_cca_attention_bwd : de = softmax backward of dA = scale * dout[p] . v[partner(p, z)] */
int seg_cca_attention(int dtype, const void* q, long ldq, const void* k, long ldk, int N, int H,
                      int W, int C, float* att, void* stream);
int seg_cca_attention_bwd(int dtype, const void* dout, long lddo, const void* v, long ldv, int N,
                          int H, int W, int C, const float* att, const float* scale, float* de,
                          void* stream);
int seg_cca_map(int dtype, const float* wt, const void* b, long ldb, int N, int H, int W, int C,
                int transposed, const float* gamma, const void* res, long ldres, void* out,
                long ldo, void* raw, long ldraw, void* stream);

/* ---- DANet position / channel attention (segmentron/modules/module.py:100-162).  The two
 * torch.bmm of each module are GEMMs on the convolution entry points (seg_conv_gemm_fwd = NT,
 * seg_conv_gemm_wgrad = TN); the softmax between them (nn.Softmax(dim=-1), module.py:110,141):
 *   seg_row_softmax     : a[r][j] = softmax_j(sign * e[r][j]) for j < L, 0 for L <= j < Lp
 *   seg_row_softmax_bwd : de[r][j] = sign * a[r][j] * (g[r][j] - sum_j a[r][j] g[r][j])
 * e / a / g / de: row-major [R][pitch], element type DT_F32 (0) or DT_BF16 (1) given per operand;
 * sign = -1 is CAM's softmax(max(energy) - energy) (shift invariance). */
int seg_row_softmax(const void* e, int dt_in, long lde, void* a, int dt_out, long lda, long R,
                    int L, int Lp, float sign, void* stream);
int seg_row_softmax_bwd(const void* a, int dt_a, long lda, const void* g, int dt_g, long ldg,
                        void* de, int dt_out, long ldde, long R, int L, int Lp, float sign,
                        void* stream);
/* ---- SyncBatchNorm statistics exchange: one-hop xGMI peer writes ----------------------------
 * Replaces, for the data-parallel path of tools/train.py:73-79 (convert_sync_batchnorm +
 * DistributedDataParallel), the two collectives per BatchNorm that torch's SyncBatchNorm issues
 * (torch/nn/modules/_functions.py:49-74 all_gather of mean/invstd/count, :140 all_reduce of the
 * backward sums): an in-place float64 SUM over the ranks of one node, every rank writing its
 * vector into a mailbox on each peer (csrc/p2p.hip).  One process per GPU.
 *   seg_p2p_create       : this rank's mailbox (uncached device memory), slot_bytes % 128 == 0
 *                          bounds one message; *handle_out is an opaque object of THIS process
 *   seg_p2p_ipc_handle   : the 64-byte hipIpcMemHandle_t of the mailbox, to be sent to the peers
 *   seg_p2p_connect      : handles = [world][64] bytes in rank order (own entry ignored)
 *   seg_p2p_all_reduce_f64 / _f32: buf[n] <- sum over ranks (added in rank order: bit-identical
 *                          on every rank), on `stream` (capturable); every rank must issue the
 *                          same sequence of calls
 *   seg_p2p_status       : synchronises; 3 = a peer failed to publish within the time limit (the
 *                          kernel stops waiting instead of hanging, and this and every later
 *                          exchange returns NaN — the error word is never reset)
 *   seg_p2p_set_timeout  : bound of one in-kernel wait in seconds (default 600)                 */
int seg_p2p_create(int rank, int world, long slot_bytes, void** handle_out);
int seg_p2p_ipc_handle(void* handle, void* out64);
int seg_p2p_connect(void* handle, const void* handles);
int seg_p2p_all_reduce_f64(void* handle, void* buf, int n, void* stream);
int seg_p2p_all_reduce_f32(void* handle, void* buf, int n, void* stream);
int seg_p2p_status(void* handle);
int seg_p2p_set_timeout(void* handle, double seconds);
int seg_p2p_destroy(void* handle);
/* The BatchNorm finalize steps with the exchange INSIDE the kernel (column sums of this rank's
 * partial rows -> peer writes -> finalize): a SyncBatchNorm then costs the same single launch per
 * direction as a plain BatchNorm.  p2p = handle of seg_p2p_create; Rb <= 1024; ws (>= 128 C
 * doubles) is needed for R > 1024 as in seg_bn_finalize_p; count_out /
 * count_dev: the GLOBAL element count (float64, device) produced by the forward step and read by
 * the backward ones; grad_scale multiplies dgamma / dbeta (1 / world, see seg_bn_bwd_finalize_s). */
int seg_bn_finalize_p_sync(void* p2p, const float* partial, long R, double local_count,
                           const float* gamma, const float* beta, float eps, float momentum,
                           float* running_mean, float* running_var, float* mean, float* invstd,
                           float* scale, float* shift, int C, const float* mean_offset,
                           double* count_out, double* ws, void* stream);
int seg_bn_bwd_finalize_p_sync(void* p2p, const float* partial, long R, const double* count_dev,
                               const float* mean, const float* invstd, const float* gamma,
                               float* dgamma, float* dbeta, float* c0, float* c1, int C,
                               double grad_scale, double* ws, void* stream);
int seg_dw_bwd_finalize_sync(void* p2p, const float* partial_bn, int Rb, const double* count_dev,
                             const float* mean, const float* invstd, const float* gamma,
                             float* dgamma, float* dbeta, float* c0, float* c1,
                             const float* partial_w, int Rw, float* dw_c9, int C,
                             double grad_scale, void* stream);
int seg_fold_bwd_finalize_sync(void* p2p, const float* dsdt, int rows, const double* count_dev,
                               const float* mean, const float* invstd, const float* gamma,
                               const float* scale, float* dgamma, float* dbeta, float* c0,
                               float* c1, int C, double grad_scale, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SEGMENTRON_HIP_H */

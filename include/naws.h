/*
 * naws.h — C ABI of the MI355X-native NA-fWebSOD hot path (libnaws_hip.so).
 *
 * Every entry point is `extern "C"`, takes plain device pointers + sizes + a
 * `void* stream` (a hipStream_t; NULL = the null stream), allocates nothing,
 * launches asynchronously on that stream and returns 0 (NAWS_OK) or a
 * negative NAWS_ERR_* code.  Shape / argument violations return an error
 * where the reference op's CAFFE_ENFORCE* would throw (sites cited per
 * function).  The two stateful reference ops (Stat: cur_iter_/init_;
 * ACMWeightDecayMomentumSGDUpdate: iter_count_) take the state as explicit
 * caller-owned arguments.  Entry points are re-entrant across host threads,
 * streams and devices (the kernels run on the calling thread's current
 * device, which must own the pointers).  What the library keeps process-wide,
 * all of it listed here: (1) the per-thread error code behind
 * naws_last_hip_error(); (2) a per-(kernel, device) record of which kernels
 * have had their dynamic-LDS limit raised - a cache of an idempotent driver
 * call, see naws_launch_state_reset(); (3) the A/B knobs of
 * naws_set_variant() (accumulation order at most).  It reads no environment
 * variables.
 *
 * All tensors are dense fp32 unless noted.  "ref:" citations are relative to
 * the upstream repository root (shenyunhang/NA-fWebSOD).
 */
#ifndef NAWS_H_
#define NAWS_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NAWS_OK 0
#define NAWS_ERR_SHAPE (-1)       /* a dimension is inconsistent / non-positive */
#define NAWS_ERR_ARG (-2)         /* an argument value is out of range */
#define NAWS_ERR_NULL (-3)        /* a required pointer is NULL */
#define NAWS_ERR_LAUNCH (-4)      /* hipGetLastError() != hipSuccess after launch */
#define NAWS_ERR_UNSUPPORTED (-5) /* valid in the reference, not built here */

#define NAWS_LAYOUT_NCHW 0
#define NAWS_LAYOUT_NHWC 1

/* Library identification; also used by the CPU-side "does it load" test. */
const char* naws_version(void);
/* hipError_t of the most recent failed launch in this thread (0 if none). */
int naws_last_hip_error(void);

/* ------------------------------------------------------------------------ *
 * a-1  VGG-16 conv body   ref: detectron/modeling/VGG16.py:9-48 (Caffe2
 *      Conv / Relu / MaxPool built-ins, pytorch v1.3.0 caffe2/operators).
 * Activations are NHWC inside the body (channel-contiguous rows feed the
 * implicit-GEMM MFMA kernel); the two boundary transposes are explicit.
 * ------------------------------------------------------------------------ */

/* conv1_1: X NCHW [N,3,H,W] -> Y NHWC [N,H,W,Cout]; 3x3, pad 1, stride 1,
 * optional fused ReLU.  Wt is the reference blob layout [Cout,3,3,3] (OIHW). */
int naws_conv3x3_c3_nchw_to_nhwc_fwd(const float* X, const float* Wt, const float* bias,
                                     int N, int H, int W, int Cout, int relu,
                                     float* Y, void* stream);

/* Repack a reference conv weight blob [Cout,Cin,3,3] (OIHW) into the
 * K-contiguous implicit-GEMM operand [Cout][kh][kw][Cin]. */
int naws_conv3x3_pack_weight(const float* W_oihw, int Cout, int Cin, float* W_packed,
                             void* stream);

/* 3x3 conv, stride 1, pad == dilation (1 or 2), NHWC in/out, MFMA implicit
 * GEMM (M = N*H*W pixels, N = Cout, K = 9*Cin), bias + optional ReLU fused.
 * Requires Cin % 32 == 0 and Cout % 32 == 0 (true for every VGG layer but
 * conv1_1).  Wp is the packed weight from naws_conv3x3_pack_weight. */
int naws_conv3x3_nhwc_fwd(const float* X, const float* Wp, const float* bias,
                          int N, int H, int W, int Cin, int Cout, int dilation, int relu,
                          float* Y, void* stream);

/* Winograd F(2x2,3x3) form of the same convolution for the deep layers (2.25x fewer MFMA
 * flops; fp32, within ~1e-6 relative of the direct sum).  U = weight_transform(W_oihw) is
 * [16][Cout][Cin]; workspace holds naws_winograd_workspace_floats(...) floats.  Dilation d
 * runs the d*d (y%d, x%d) sub-grids as independent dense convolutions.  Cin, Cout % 4 == 0. */
int64_t naws_winograd_workspace_floats(int N, int H, int W, int Cin, int Cout, int dilation);
int naws_winograd_weight_transform(const float* W_oihw, int Cout, int Cin, float* U,
                                   void* stream);
int naws_conv3x3_winograd_nhwc_fwd(const float* X, const float* U, const float* bias,
                                   int N, int H, int W, int Cin, int Cout, int dilation, int relu,
                                   float* workspace, float* Y, void* stream);

/* MaxPool kernel 2, pad 0, stride 1 or 2, floor output size (Caffe2 legacy
 * pooling rule: Ho = (H - 2) / stride + 1).  NHWC in/out. */
int naws_maxpool2x2_nhwc_fwd(const float* X, int N, int H, int W, int C, int stride,
                             float* Y, void* stream);

int naws_nchw_to_nhwc(const float* X, int N, int C, int H, int W, float* Y, void* stream);
int naws_nhwc_to_nchw(const float* X, int N, int H, int W, int C, float* Y, void* stream);

/* ------------------------------------------------------------------------ *
 * a-2 / a-3  RoIPoolF (+ RoIFeatureBoost)
 *   ref: detectron/modeling/detector.py:268-331 (op emission),
 *        detectron/ops/roi_loop_pool_op.cu:31-101 (in-tree statement of the
 *        Caffe2 RoIPoolF arithmetic; :72 holds the original empty-bin rule),
 *        detectron/ops/roi_feature_boost_op.cc:8-35.
 * X: [N,C,H,W] (layout NCHW) or [N,H,W,C] (layout NHWC).  rois: [R,5] =
 * (batch_idx, x1, y1, x2, y2) in input-image pixels.  Y: [R,C,ph,pw].
 * argmax (int32, same shape as Y, value h*W+w or -1) may be NULL.
 * boost may be NULL; otherwise [R] and Y[r,...] *= boost[r] (fused a-3).
 * Errors: R<0, C/H/W/ph/pw<=0 -> SHAPE; layout unknown -> ARG.
 * ------------------------------------------------------------------------ */
int naws_roi_pool_f_fwd(const float* X, int layout, int N, int C, int H, int W,
                        const float* rois, int R, const float* boost,
                        int pooled_h, int pooled_w, float spatial_scale,
                        float* Y, int32_t* argmax, void* stream);

/* Y[r,f] = X[r,f] * S[r]   (in place allowed)  ref: roi_feature_boost_op.cc:8-35 */
int naws_roi_feature_boost_fwd(const float* X, const float* S, int R, int F, float* Y,
                               void* stream);
/* dX[r,f] = dY[r,f] * S[r]                     ref: roi_feature_boost_op.cc:37-66 */
int naws_roi_feature_boost_bwd(const float* dY, const float* S, int R, int F, float* dX,
                               void* stream);

/* ------------------------------------------------------------------------ *
 * a-7  RoIIoU   ref: detectron/ops/roi_iou_op.cu:27-62 (kernel), :65-84 (op).
 * rois [R,5] -> J [R,R]; coordinates truncated to int, diagonal forced to 1.
 * ------------------------------------------------------------------------ */
int naws_roi_iou_fwd(const float* rois, int R, float* J, void* stream);

/* ------------------------------------------------------------------------ *
 * a-4  FC layers as fp32 MFMA GEMMs (v_mfma_f32_32x32x2_f32)
 *   ref: detectron/modeling/wsl_heads.py:654-681, webly_heads.py:463-502
 *        (Caffe2 FC: Y = X W^T + b with W [out,in]; Relu; Dropout scale
 *        1/(1-ratio)).
 * ------------------------------------------------------------------------ */

/* Epilogues of naws_gemm_f32 */
#define NAWS_EPI_NONE 0          /* C = acc                                   */
#define NAWS_EPI_BIAS 1          /* C = acc + bias[n]                         */
#define NAWS_EPI_BIAS_RELU 2     /* C = max(acc + bias[n], 0)                 */
#define NAWS_EPI_BIAS_RELU_DROP 3 /* C = max(acc+bias[n],0) * keep(m,n)/(1-p) */
#define NAWS_EPI_GATE_POS 4      /* C = aux[m,n] > 0 ? acc * alpha : 0        */

/* General row-major fp32 GEMM:  C[M,N] (+)= op(A)[M,K] * op(B)[K,N].
 *   transA == 0: A is [M,K] (lda >= K);  transA == 1: A is stored [K,M] (lda >= M)
 *   transB == 0: B is [K,N] (ldb >= N);  transB == 1: B is stored [N,K] (ldb >= K)
 * `batch` independent problems with element strides strideA/B/C (bias and
 * aux stride by strideBias / strideC).  accumulate != 0 adds to C.
 * Dropout keep(m,n) is a counter-based hash of (seed, m*N+n); ratio in [0,1).
 * Alignment: every leading dimension and pointer offset must be a multiple
 * of 4 floats (16 B) -> NAWS_ERR_ARG otherwise. */
int naws_gemm_f32(int transA, int transB, int M, int N, int K,
                  const float* A, int lda, const float* B, int ldb, float* C, int ldc,
                  int batch, int64_t strideA, int64_t strideB, int64_t strideC,
                  int epilogue, const float* bias, int64_t strideBias,
                  const float* aux, int ldaux, float alpha,
                  float drop_ratio, uint64_t seed, int accumulate, void* stream);
/* The same product for a SMALL output with a LONG inner dimension (fc8: logits = H7 W8^T, an
 * [R x 2C] block from K = 4096; dW8 = dL^T H7, [2C x 4096] from K = R): K is cut into `ksplit`
 * slices so that more workgroups share the walk, the partial products go to `workspace`
 * (naws_gemm_f32_splitk_workspace_floats floats, caller-owned) and a second pass adds them in
 * slice order - deterministic - and applies the epilogue (NAWS_EPI_NONE or NAWS_EPI_BIAS only).
 * ldc / C need no alignment here.  ref: Caffe2 FC / FCGradient of fc8c, fc8d
 * (detectron/modeling/wsl_heads.py:213-227). */
int64_t naws_gemm_f32_splitk_workspace_floats(int M, int N, int batch, int ksplit);
int naws_gemm_f32_splitk(int transA, int transB, int M, int N, int K, const float* A, int lda,
                         const float* B, int ldb, float* C, int ldc, int batch, int64_t strideA,
                         int64_t strideB, int64_t strideC, int epilogue, const float* bias,
                         int64_t strideBias, int ksplit, float* workspace, void* stream);

/* The dropout keep-mask the GEMM epilogue applies, materialised (tests and
 * the op-level Dropout API): mask[i] = keep(seed, i) ? 1 : 0, i in [0,n). */
int naws_dropout_mask(uint64_t seed, float drop_ratio, int64_t n, float* mask, void* stream);

/* db[n] (+)= sum_m dY[m,n]   (FC bias gradient) */
int naws_colsum_f32(const float* dY, int M, int N, int ld, float* db, int accumulate,
                    void* stream);

/* ------------------------------------------------------------------------ *
 * a-5 / a-6  WSDDN two-stream outputs, both branches, per-image segments
 *   ref: detectron/modeling/wsl_heads.py:23-56 (add_wsl_outputs),
 *        detectron/modeling/webly_heads.py:32-74 (add_webly_outputs),
 *        detectron/modeling/wsl_heads.py:213-227 (add_cls_pred).
 * Inputs: four logit matrices [Rt,C] with row stride ld (floats).  Branch 0
 * ("clean") uses (fc8c, fc8d); branch 1 ("noise") uses (fc8c + noisy_fc8c,
 * fc8d + noisy_fc8d).  noisy_* may both be NULL -> only branch 0 is built.
 * seg_off: int32 DEVICE array [nseg+1]; rows seg_off[s]..seg_off[s+1]-1 are
 * image s (the reference is nseg == 1).  Softmax over classes is per row;
 * softmax over proposals and the ReduceSum are per segment.
 * Outputs (branch-major): alpha_cls, alpha_det, rois_pred [nb,Rt,C];
 * cls_prob [nb,nseg,C];  nb = 1 or 2.
 * ------------------------------------------------------------------------ */
int naws_wsddn_outputs_fwd(const float* fc8c, const float* fc8d,
                           const float* noisy_fc8c, const float* noisy_fc8d, int ld,
                           const int32_t* seg_off, int nseg, int Rt, int C,
                           float* alpha_cls, float* alpha_det, float* rois_pred,
                           float* cls_prob, void* stream);

/* a-11 (head part): backward of the above through Mul, both Softmaxes,
 * ReduceSum and the residual Add.
 * d_cls_prob [nb,nseg,C] -> d_fc8c, d_fc8d, d_noisy_fc8c, d_noisy_fc8d
 * [Rt,C] with row stride ldd.  d_fc8c/d_fc8d receive clean + noise
 * gradients (fan-in of the Add, webly_heads.py:57-61). */
int naws_wsddn_outputs_bwd(const float* alpha_cls, const float* alpha_det,
                           const float* rois_pred, const float* cls_prob,
                           const float* d_cls_prob, const int32_t* seg_off, int nseg,
                           int Rt, int C, int nb,
                           float* d_fc8c, float* d_fc8d, float* d_noisy_fc8c,
                           float* d_noisy_fc8d, int ldd, void* stream);

/* ------------------------------------------------------------------------ *
 * a-8  Spatial (IoU-graph) entropy gate
 *   ref: detectron/modeling/webly_heads.py:265-391 (add_spatial_entropy_weight)
 * rois [Rt,5], rois_pred [Rt,C] (branch 0), cls_prob [nseg,C], labels_oh
 * [nseg,C] -> class_weight, class_weight_noise [nseg,C]; also exports
 * hatE_sum and hatE_sum_norm [nseg,C] (inputs of the Stat ops).  J is never
 * materialised: IoU is recomputed per (r,j) pair inside the J@E product.
 * workspace: fp32, at least naws_entropy_gate_workspace_floats(Rt,C,nseg,max_seg).
 * ------------------------------------------------------------------------ */
int64_t naws_entropy_gate_workspace_floats(int Rt, int C, int nseg, int max_seg_len);
int naws_entropy_gate_fwd(const float* rois, const float* rois_pred, const float* cls_prob,
                          const float* labels_oh, const int32_t* seg_off, int nseg,
                          int Rt, int C, int max_seg_len, float* workspace,
                          float* class_weight, float* class_weight_noise,
                          float* hatE_sum, float* hatE_sum_norm, void* stream);

/* ------------------------------------------------------------------------ *
 * a-9  (Weighted)CrossEntropyWithLogits forward / gradient
 *   ref: detectron/ops/cross_entropy_wsl_op.cc:7-45 (CE fwd), :47-85 (CE
 *        grad), :87-132 (WCE fwd), :134-180 (WCE grad); thresholds
 *        cross_entropy_wsl_op.h:90-110 (1e-20 / 1e+4).
 * X, L, W: [N,C] (W NULL -> unweighted op).  Y: scalar.  The loss is
 * accumulated serially in index order by one lane, as the CPU op does.
 * nprob independent problems laid out back to back (X,L,W,dX: [nprob,N,C];
 * Y,dY: [nprob]) run in one launch; the reference op is nprob == 1.
 * Errors: N<=0 or C<=0 -> SHAPE (ENFORCE sites .cc:16-17, :98-104).
 * ------------------------------------------------------------------------ */
int naws_weighted_ce_fwd(const float* X, const float* L, const float* W, int N, int C,
                         int is_mean, int nprob, float* Y, void* stream);
int naws_weighted_ce_bwd(const float* X, const float* L, const float* W, const float* dY,
                         int N, int C, int is_mean, int nprob, float* dX, void* stream);
/* The same for nprob problems whose labels repeat with period lab_period (L: [lab_period,N,C];
 * problem p is scored against L[p % lab_period]) - the loss tail of the noise-aware head runs
 * 2 branches x nseg images against ONE labels_oh per image (webly_heads.py:167-197: both
 * add_cross_entropy_loss calls name the same 'labels_oh' blob).  bwd: dY NULL -> every problem's
 * upstream gradient is dy_const (the loss seed 1.0 of utils/blob.py:167-173). */
int naws_weighted_ce_shared_fwd(const float* X, const float* L, const float* W, int N, int C,
                                int is_mean, int nprob, int lab_period, float* Y, void* stream);
int naws_weighted_ce_shared_bwd(const float* X, const float* L, const float* W, const float* dY,
                                float dy_const, int N, int C, int is_mean, int nprob, int lab_period,
                                float* dX, void* stream);

/* ------------------------------------------------------------------------ *
 * a-12  ACMWeightDecayMomentumSGDUpdate (fused), one launch per arena
 *   ref: detectron/ops/acm_weightdecay_momentum_sgd_op.h:48-112,
 *        detectron/ops/acm_weightdecay_momentum_sgd_op_gpu.cu:7-33,
 *        detectron/modeling/optimizer_wsl.py:96-137 (per-param wd / lr_mult).
 * Parameters live in one contiguous arena of `total` floats cut into `nseg`
 * segments; seg_end[i] (int64 DEVICE array, exclusive prefix ends),
 * seg_lr_mult[i], seg_wd[i] (fp32 DEVICE arrays).  lr is a DEVICE scalar.
 * iter_count is the op's caller-owned state BEFORE this call (the op zeroes
 * momentum and acmgrad when it is 0, .h:62-69).  Semantics per element:
 *   acm += g; if ((iter_count+1) % iter_size == 0) { acm *= 1/(iter_size*
 *   gpu_num); acm += wd*p; m = lr*lr_mult*acm + mom*m; p -= m (or the
 *   nesterov form); acm = 0 }.  The grad blob is read-only (the op's
 *   update value goes to the acmgrad output, which is then cleared,
 *   .h:94-108).  acmgrad may be NULL when iter_size == 1: the buffer is then
 *   identically zero between calls and the 3-read/2-write form is exact.
 * ------------------------------------------------------------------------ */
int naws_acm_sgd_update(const float* grad, float* momentum_buf, const float* lr, float* param,
                        float* acmgrad, int64_t total, const int64_t* seg_end,
                        const float* seg_lr_mult, const float* seg_wd, int nseg,
                        float momentum, int nesterov, int iter_size, int gpu_num,
                        int64_t iter_count, void* stream);
/* The same update, additionally reporting max|updated parameter| per matrix row of up to 4
 * regions of the arena (for the fp16x2 re-split of the weights that follows: naws_split_f16x2_dual
 * then needs no maxima pass).  rm_table_host: HOST int64 [n_rm][4] = {first element, end element,
 * row length, first index in rowmax}, n_rm <= 4, starts and row lengths multiples of 256 floats
 * (passed to the kernel by value).  rowmax: uint32 bit patterns, atomically max-ed into (the
 * caller zeroes them); untouched on iterations that only accumulate. */
int naws_acm_sgd_update_rowmax(const float* grad, float* momentum_buf, const float* lr,
                               float* param, float* acmgrad, int64_t total, const int64_t* seg_end,
                               const float* seg_lr_mult, const float* seg_wd, int nseg,
                               float momentum, int nesterov, int iter_size, int gpu_num,
                               int64_t iter_count, uint32_t* rowmax, const int64_t* rm_table_host,
                               int n_rm, void* stream);

/* The same update for iter_size == 1 (acmgrad NULL), additionally writing the fp16x2 operand planes
 * of up to 4 weight matrices of the arena itself (the head GEMMs read fc6_w / fc7_w as row-scaled
 * f16 hi / lo planes, naws_split_f16x2): no re-split pass over the weights after the update.
 * A row's scale is taken from 2 x bound[row], bound = max|w| of the row BEFORE this update (what
 * the previous call reported in rowmax, or naws_split_f16x2's maxima) - an upper bound is all a
 * scale needs.  rowmax (zeroed by the caller, != bound) receives max|w| AFTER the update,
 * inv_scale the 1/scale of the planes just written.  If any |w_new| exceeds its 2 x bound (or is
 * NaN), *overflow = max(*overflow, overflow_tag): queue naws_split_f16x2_rows_if(...,
 * overflow, overflow_tag) behind this call and the planes are redone from the exact maxima, on
 * the device's own decision.  regions: HOST array, ascending and disjoint; rows and
 * rows_per_batch multiples of 32, cols of 256, one (lr_mult, weight decay) run per region.
 * rows < rows_per_batch describes a block of rows of ONE batch item of a larger matrix: planes then
 * points at the block's first row inside that matrix's planes (colmax must be NULL).
 * Parameters, momentum: bit-identical to naws_acm_sgd_update.
 *   ref: detectron/ops/acm_weightdecay_momentum_sgd_op.h:72-109. */
typedef struct naws_sgd_plane_region {
  int64_t start;           /* first arena element of the matrix */
  int32_t rows, cols;      /* row-major, contiguous (ld = cols) */
  int32_t rows_per_batch;  /* planes: f16 [2][rows / rows_per_batch][cols / 16][rows_per_batch][16] */
  int32_t reserved;
  void* planes;            /* NULL: the matrix is left untouched - parameters, momentum and planes
                            * were already updated by naws_gemm_f32_f16x2_nt_xk_sgd */
  int64_t plane_stride;    /* elements between the hi and the lo plane */
  const uint32_t* bound;   /* [rows] */
  uint32_t* rowmax;        /* [rows] */
  float* inv_scale;        /* [rows] */
  uint32_t* colmax;        /* [rows / rows_per_batch][cols], nullable, NAWS_PLANES_F16X2 only: receives
                            * max|w| per column and batch item after the update (atomic max of bit
                            * patterns; caller zeroes) - the maxima naws_split_f16x2_dual needs for
                            * the TRANSPOSED operand planes (fc7_w^T of the dgrad) */
} naws_sgd_plane_region;
int naws_acm_sgd_update_f16x2(const float* grad, float* momentum_buf, const float* lr, float* param,
                              int64_t total, const int64_t* seg_end, const float* seg_lr_mult,
                              const float* seg_wd, int nseg, float momentum, int nesterov,
                              int gpu_num, int64_t iter_count, const naws_sgd_plane_region* regions,
                              int n_regions, int32_t* overflow, int32_t overflow_tag, void* stream);
/* The same with the plane format of the other 16-bit arithmetic plans: NAWS_PLANES_BF16X3 = the
 * exact three-plane bf16 split (naws_split_bf16x3: planes [3][batch][cols/16][rows_per_batch][16]),
 * NAWS_PLANES_BF16 = one plane of the weights rounded to bf16 (naws_to_bf16_slab; cols % 64 == 0).
 * Neither carries a scale: bound / rowmax / inv_scale of the regions and the overflow word are
 * unused (may be NULL).  NAWS_PLANES_F16X2 is naws_acm_sgd_update_f16x2. */
#define NAWS_PLANES_F16X2 0
#define NAWS_PLANES_BF16X3 1
#define NAWS_PLANES_BF16 2
int naws_acm_sgd_update_planes(int format, const float* grad, float* momentum_buf, const float* lr,
                               float* param, int64_t total, const int64_t* seg_end,
                               const float* seg_lr_mult, const float* seg_wd, int nseg,
                               float momentum, int nesterov, int gpu_num, int64_t iter_count,
                               const naws_sgd_plane_region* regions, int n_regions,
                               int32_t* overflow, int32_t overflow_tag, void* stream);
/* naws_split_f16x2's row-scaled form from given maxima (rowmax [batch][rows] bit patterns ->
 * planes [2][batch][kpad/16][rows][16], inv_scale [batch][rows]), run only if *cond == cond_value
 * when the kernel executes (cond NULL: always): the fallback of naws_acm_sgd_update_f16x2. */
int naws_split_f16x2_rows_if(const float* X, int batch, int rows, int cols, int ld, int64_t strideX,
                             const uint32_t* rowmax, void* P, float* inv_scale, int kpad,
                             const int32_t* cond, int32_t cond_value, void* stream);
/* The same for rows [row0, row0 + rows) of an unbatched [plane_rows, cols] matrix (X = its row 0;
 * rowmax, inv_scale [plane_rows]; planes [2][kpad/16][plane_rows][16]): one row block of a weight
 * matrix - an fc6_w piece of the pipelined N > 1 update, the rows a rank does not own under
 * NAWS.SHARDED_UPDATE. */
int naws_split_f16x2_row_range_if(const float* X, int plane_rows, int row0, int rows, int cols,
                                  int ld, const uint32_t* rowmax, void* P, float* inv_scale, int kpad,
                                  const int32_t* cond, int32_t cond_value, void* stream);

/* ------------------------------------------------------------------------ *
 * a-14  Stat accumulate   ref: detectron/ops/stat_op.cu:14-20, :24-78
 * AI += I*L; AL += L  (n elements).  init != 0 zeroes AI/AL first (the op's
 * init_ flag).  Printing AI/AL is host-side (ops.Stat).
 * ------------------------------------------------------------------------ */
int naws_stat_accumulate(const float* I, const float* L, int n, int init, float* AI,
                         float* AL, void* stream);

/* ------------------------------------------------------------------------ *
 * Small Caffe2 built-ins used op-by-op by the graph executor
 * ------------------------------------------------------------------------ */
#define NAWS_UN_LOG 0
#define NAWS_UN_SCALE 1      /* y = x * a              */
#define NAWS_UN_REPLACE_NAN 2 /* y = isnan(x) ? a : x   */
#define NAWS_UN_LEAKY_RELU 3 /* y = x >= 0 ? x : a*x   */
#define NAWS_UN_CLIP 4       /* y = min(max(x,a),b)    */
#define NAWS_UN_RELU 5
int naws_unary_f32(int op, const float* X, int64_t n, float a, float b, float* Y, void* stream);

#define NAWS_BIN_ADD 0
#define NAWS_BIN_SUB 1
#define NAWS_BIN_MUL 2
#define NAWS_BIN_DIV 3
#define NAWS_BIN_GATE_POS 4 /* Y = B > 0 ? A : 0   (ReluGradient: A = dY, B = Y) */
/* Y[r,c] = A[ra,ca] op B[rb,cb], numpy-style broadcasting of two 2-D
 * operands: each of (rowsA, colsA, rowsB, colsB) is either 1 or the output
 * extent. */
int naws_binary_f32(int op, const float* A, int rowsA, int colsA, const float* B, int rowsB,
                    int colsB, float* Y, int rows, int cols, void* stream);

/* Softmax along axis 1 of a [rows, cols] matrix (max-subtracted). */
int naws_softmax_rows_fwd(const float* X, int rows, int cols, float* Y, void* stream);
int naws_softmax_rows_bwd(const float* Y, const float* dY, int rows, int cols, float* dX,
                          void* stream);
int naws_transpose2d_f32(const float* X, int rows, int cols, float* Y, void* stream);
/* Y[c] = sum_r X[r,c]  (ReduceSum axes=[0]) */
int naws_reduce_sum_axis0(const float* X, int rows, int cols, float* Y, void* stream);

/* ---- bf16 MFMA option (BASELINE.json configs[3]: "bf16 MFMA conv/fc with fp32 loss") ----------
 * Replaces the same Caffe2 FC / FCGradient / Conv operators as naws_gemm_f32 and
 * naws_conv3x3_nhwc_fwd (reference: detectron/modeling/wsl_heads.py:33-47 model.FC calls,
 * detectron/modeling/VGG16.py:37-130 model.Conv calls), computing in
 * v_mfma_f32_32x32x16_bf16 with fp32 accumulation and fp32 results.
 *
 * C[M,N] (+)= A[M,K] * B[N,K]^T ; both operands K-contiguous.  Each operand is read either as
 * fp32 (rounded to bf16, round-to-nearest-even, inside the kernel) or as bf16 (uint16 storage).
 * lda/ldb/strideA/strideB are in elements of the operand's own type and, like K, must be
 * multiples of 8; A and B must be 16-byte aligned.  Epilogue arguments as naws_gemm_f32. */
int naws_gemm_bf16_nt(int M, int N, int K, const void* A, int a_is_bf16, int lda, const void* B,
                      int b_is_bf16, int ldb, float* C, int ldc, int batch, int64_t strideA,
                      int64_t strideB, int64_t strideC, int epilogue, const float* bias,
                      int64_t strideBias, const float* aux, int ldaux, float alpha,
                      float drop_ratio, uint64_t seed, int accumulate, void* stream);
/* 3x3 / stride 1 / pad = dilation convolution on NHWC fp32 activations with the packed weight of
 * naws_conv3x3_pack_weight, bf16 MFMA inner product (Cin % 32 == 0). */
int naws_conv3x3_nhwc_bf16_fwd(const float* X, const float* Wp, const float* bias, int N, int H,
                               int W, int Cin, int Cout, int dilation, int relu, float* Y,
                               void* stream);
/* The same operator on the wave-private halo-tile kernel (round 4; the bf16 plan's conv body):
 * W1 = naws_to_bf16_slab of the packed weight viewed [Cout][9 * Cin] (bf16 [9 * Cin / 16][Cout][16]);
 * Cin % 16 == 0, 9 * Cin % 64 == 0, Cout % 64 == 0, dilation 1 or 2.  pool2: the 2x2 / stride-2
 * max-pool that follows (reference: VGG16.py:14-36 pool1..pool3) taken in the epilogue, Y is then
 * [N][H/2][W/2][Cout] (dilation 1 only). */
int naws_conv3x3_nhwc_bf16_wp_fwd(const float* X, const void* W1, const float* bias, int N, int H,
                                  int W, int Cin, int Cout, int dilation, int relu, int pool2,
                                  float* Y, void* stream);
/* Y[b][c][r] = bf16(X[b][r][c]) for r < rows, 0 for rows <= r < rows_pad: the K-contiguous bf16
 * copies the backward FC GEMMs (dX = dY W, dW = dY^T X) take as operands. */
int naws_transpose_to_bf16(const float* X, int batch, int rows, int cols, int ld, int rows_pad,
                           void* Y, void* stream);

/* ---- fp32 GEMM on the bf16 matrix cores ("3 x bf16" split, fp32-accurate) ----------------------
 * Same operators as naws_gemm_f32 (Caffe2 FC / FCGradient of fc6 / fc7; reference
 * detectron/modeling/wsl_heads.py:674-679, webly_heads.py:490-498).  An fp32 value is exactly
 * a1 + a2 + a3 with three bf16 values; the GEMM keeps the six products a_p*b_q with p+q <= 4
 * (dropped terms < 2^-26 |ab|) and accumulates them in fp32 inside v_mfma_f32_32x32x16_bf16.
 *
 * naws_split_bf16x3: X fp32 [batch][rows][ld] -> P bf16 [3][batch][kpad/16][outer][16]
 * ("K-slab-major": the 16-deep K-step of any run of rows is contiguous), where
 *   transpose == 0: outer = rows, K = cols;   transpose != 0: outer = cols, K = rows
 * (the K-contiguous operand of dW = dY^T X).  kpad = K rounded up to 16, the pad is zero-filled.
 * X == P[0] + P[1] + P[2] exactly for |x| >= 2^-109 (below, residues are denormal and flushed).
 * naws_gemm_f32x3_nt: C[M,N] (+)= A[M,K] B[N,K]^T on such planes; slabA/slabB = elements
 * between K slabs (outer * 16; A3/B3 may point at a row offset inside a larger operand),
 * planeA/planeB = elements between planes, K % 16 == 0.  Epilogue arguments as naws_gemm_f32. */
int naws_split_bf16x3(const float* X, int batch, int rows, int cols, int ld, int64_t strideX,
                      int transpose, int kpad, void* P, void* stream);
int naws_gemm_f32x3_nt(int M, int N, int K, const void* A3, int64_t slabA, int64_t planeA,
                       const void* B3, int64_t slabB, int64_t planeB, float* C, int ldc, int batch,
                       int64_t strideA, int64_t strideB, int64_t strideC, int epilogue,
                       const float* bias, int64_t strideBias, const float* aux, int ldaux,
                       float alpha, float drop_ratio, uint64_t seed, int accumulate, void* stream);
/* ---- fp32 GEMM on the f16 matrix cores ("2 x f16" split with per-row scaling) -------------------
 * Same operators again (Caffe2 FC / FCGradient of fc6 / fc7; wsl_heads.py:674-679,
 * webly_heads.py:490-498), half the MFMA work of the bf16 split.  Every row of an NT operand
 * (= one output row or column) is scaled by s = 2^(14 - floor(log2 max|x|)) and written as
 * hi = f16(x s), lo = f16(x s - hi):  |x s - hi - lo| <= max(2^-22 |x s|, 2^-25), i.e. >= 22
 * significand bits for every element within 2^-18 of its row's maximum and an absolute floor of
 * 2^-39 of the row maximum below that.  The GEMM keeps hi*hi + hi*lo + lo*hi (dropped: lo*lo
 * < 2^-22 |ab|), accumulates in fp32 inside v_mfma_f32_32x32x16_f16 and multiplies the
 * accumulator by 1/(s_row s_col) (exact).  s is capped at 2^101: rows whose maximum is below
 * 2^-87 keep an absolute floor of 2^-126.  NaN / inf propagate as NaN.
 *
 * naws_split_f16x2: X fp32 [batch][rows][ld] -> P f16 [2][batch][kpad/16][outer][16] (layout and
 * transpose as naws_split_bf16x3, kpad = K rounded up to 32) and scales fp32 [2][batch][outer]:
 * [1] = 1/s per outer index (the GEMM's scaleA / scaleB), [0] = scratch.
 * naws_gemm_f32_f16x2_nt: C[M,N] (+)= A B^T; scaleA[M], scaleB[N] (strideScale* between batch
 * items), K % 32 == 0, everything else as naws_gemm_f32x3_nt. */
int naws_split_f16x2(const float* X, int batch, int rows, int cols, int ld, int64_t strideX,
                     int transpose, int kpad, void* P, float* scales, void* stream);
int naws_gemm_f32_f16x2_nt(int M, int N, int K, const void* A2, int64_t slabA, int64_t planeA,
                           const float* scaleA, const void* B2, int64_t slabB, int64_t planeB,
                           const float* scaleB, float* C, int ldc, int batch, int64_t strideA,
                           int64_t strideB, int64_t strideC, int64_t strideScaleA,
                           int64_t strideScaleB, int epilogue, const float* bias,
                           int64_t strideBias, const float* aux, int ldaux, float alpha,
                           float drop_ratio, uint64_t seed, int accumulate, void* stream);
/* The producer reports the maxima, the consumer's split reads X once.
 * naws_gemm_f32_f16x2_nt_amax / naws_gemm_f32_amax: as naws_gemm_f32_f16x2_nt / naws_gemm_f32 (same
 * reference operators: Caffe2 FC / FCGradient, wsl_heads.py:674-679, webly_heads.py:490-498), and
 * the epilogue additionally folds |C| of the values it stores into
 *   rowmax [batch][nseg][M]  max over the columns of segment n / rowmax_seg_cols (0 = one segment;
 *                            otherwise a multiple of 256), nullable
 *   colmax [batch][N]        max over rows of |C[m][n] * colmax_rowmul[m]| (colmax_rowmul nullable)
 * as uint32 bit patterns by atomic max: the caller zeroes them.  NaNs are skipped as fmaxf skips
 * them, so the maxima equal those of naws_split_f16x2's own pass bit for bit.
 * naws_split_f16x2_dual: X fp32 [batch][rows][ld] -> Pn f16 [2][batch][kpad_n/16][rows][16] scaled
 * per row from rowmax [batch][rows] and / or Pt f16 [2][batch][kpad_t/16][cols][16] = planes of
 * (diag(rowmul) X)^T scaled per column from colmax [batch][cols]; scales_n / scales_t as
 * naws_split_f16x2's scales ([1] receives 1/scale; [0] untouched).  Pn or Pt may be null. */
int naws_gemm_f32_f16x2_nt_amax(int M, int N, int K, const void* A2, int64_t slabA, int64_t planeA,
                                const float* scaleA, const void* B2, int64_t slabB, int64_t planeB,
                                const float* scaleB, float* C, int ldc, int batch, int64_t strideA,
                                int64_t strideB, int64_t strideC, int64_t strideScaleA,
                                int64_t strideScaleB, int epilogue, const float* bias,
                                int64_t strideBias, const float* aux, int ldaux, float alpha,
                                float drop_ratio, uint64_t seed, int accumulate, uint32_t* rowmax,
                                int rowmax_seg_cols, uint32_t* colmax, const float* colmax_rowmul,
                                void* stream);
/* One COLUMN RANGE [drop_col0, drop_col0 + N) of a drop_ld-wide unbatched product: B2, scaleB, bias,
 * colmax and C point at the range's first weight row / output column, rowmax at the words of the
 * rowmax segment the range lies in; the Dropout counters are those of the full-width launch
 * (element (m, n) draws counter m * drop_ld + drop_col0 + n), so the ranges of one activation can be
 * produced by separate launches with bit-identical results.  Used by the pipelined N > 1 step: fc6
 * forward is cut along the weight rows and each piece starts as soon as ITS rows of fc6_w have
 * been exchanged and updated (reference: the FC op of wsl_heads.py:674-679 after the per-blob
 * all-reduce + update of optimizer_wsl.py:52-137 - same values, earlier start). */
int naws_gemm_f32_f16x2_nt_cols(int M, int N, int K, const void* A2, int64_t slabA, int64_t planeA,
                                const float* scaleA, const void* B2, int64_t slabB, int64_t planeB,
                                const float* scaleB, float* C, int ldc, int epilogue,
                                const float* bias, float drop_ratio, uint64_t seed, uint32_t* rowmax,
                                int rowmax_seg_cols, uint32_t* colmax, int drop_ld, int drop_col0,
                                void* stream);
int naws_gemm_f32_amax(int transA, int transB, int M, int N, int K, const float* A, int lda,
                       const float* B, int ldb, float* C, int ldc, int batch, int64_t strideA,
                       int64_t strideB, int64_t strideC, int epilogue, const float* bias,
                       int64_t strideBias, const float* aux, int ldaux, float alpha,
                       float drop_ratio, uint64_t seed, int accumulate, uint32_t* rowmax,
                       int rowmax_seg_cols, uint32_t* colmax, const float* colmax_rowmul,
                       void* stream);
int naws_split_f16x2_dual(const float* X, int batch, int rows, int cols, int ld, int64_t strideX,
                          const uint32_t* rowmax, const uint32_t* colmax, const float* rowmul,
                          void* Pn, float* scales_n, int kpad_n, void* Pt, float* scales_t,
                          int kpad_t, void* stream);
/* Winograd F(2x2,3x3) convolution (as naws_conv3x3_winograd_nhwc_fwd: reference
 * detectron/modeling/VGG16.py:24-46, conv3_x .. conv5_x) with the 16 batched GEMMs in the 2 x f16
 * split: the input transform scales by one power of two per tensor (|B^T d B| <= 4 max|x|) and
 * writes the f16 hi / lo planes itself.  U2 / scaleU: naws_split_f16x2 of the transformed weight
 * viewed as batch 16 of [Cout][Cin] (scaleU = its scales[1], [16][Cout]).  Cin % 32 == 0.
 * amax_in (nullable): device word holding the bit pattern of an upper bound of max|X| (e.g. the
 * amax_out of the layer that produced X, also valid across a max-pool); null = measured here.
 * amax_out (nullable, != amax_in): device word that receives the bit pattern of max|Y|. */
int64_t naws_winograd_f16x2_workspace_floats(int N, int H, int W, int Cin, int Cout, int dilation);
int naws_conv3x3_winograd_nhwc_f16x2_fwd(const float* X, const void* U2, const float* scaleU,
                                         const float* bias, int N, int H, int W, int Cin, int Cout,
                                         int dilation, int relu, float* workspace, float* Y,
                                         const uint32_t* amax_in, uint32_t* amax_out, void* stream);
/* Winograd F(4x4,3x3) form of the same operator (csrc/winograd4.hip; reference
 * detectron/modeling/VGG16.py:33-46, conv4_x / conv5_x): 36 batched GEMMs over 6x6 input tiles,
 * 4x fewer MFMA flops than the direct sum and 0.56 of F(2x2)'s transform-domain bytes.  The
 * transforms run in fp32; the input transform scales by one power of two per tensor
 * (interpolation points 0, 1, -1, 2, -1/2, inf: |B^T d B| <= 196 max|x|) and writes the f16 hi / lo
 * planes itself.
 * naws_winograd4_weight_transform: U[36][Cout][Cin] = G g G^T of W_oihw, taken in double and
 * rounded once.  U2 / scaleU: naws_split_f16x2 of U viewed as batch 36 of [Cout][Cin] (scaleU = its
 * scales[1], [36][Cout]).  Cin % 32 == 0, Cout % 4 == 0.  amax_in (required): device word holding
 * the bit pattern of an upper bound of max|X| (naws_amax_f32, or the amax_out of the producing
 * layer).  amax_out (nullable, != amax_in): receives the bit pattern of max|Y|.  workspace:
 * naws_winograd4_f16x2_workspace_floats(...) floats, 16-byte aligned. */
int naws_winograd4_weight_transform(const float* W_oihw, int Cout, int Cin, float* U, void* stream);
int64_t naws_winograd4_f16x2_workspace_floats(int N, int H, int W, int Cin, int Cout, int dilation);
int naws_conv3x3_winograd4_nhwc_f16x2_fwd(const float* X, const void* U2, const float* scaleU,
                                          const float* bias, int N, int H, int W, int Cin, int Cout,
                                          int dilation, int relu, float* workspace, float* Y,
                                          const uint32_t* amax_in, uint32_t* amax_out, void* stream);
/* 3x3 / stride 1 / pad 1 convolution of the shallow VGG layers (conv1_2 .. conv2_2; reference
 * detectron/modeling/VGG16.py:12-22) in the 2 x f16 split: as naws_conv3x3_nhwc_f32x3_fwd's
 * halo-tile kernel with 3 MFMA terms.  W2 / scaleW: naws_split_f16x2 of the packed weight
 * (naws_conv3x3_pack_weight) viewed [Cout][9*Cin].  The activations are scaled by the power of
 * two derived from the bound (*amax_in) * in_mul + in_add >= max|X| (amax_in: bit pattern of a
 * float on the device, e.g. the amax_out of the producing layer with in_mul = 1, in_add = 0, or
 * naws_amax_f32 of the network input with the producing layer's weight L1 norm / bias maximum).
 * amax_out (nullable, != amax_in) receives the bit pattern of max|Y| (atomic max: the word is
 * zeroed by the call unless amax_out_zeroed != 0 says the caller already did).
 * pool2 != 0: Y is [N][H/2][W/2][Cout] = MaxPool 2x2 / stride 2 (VGG16.py pool1..pool3) of the
 * layer's output, taken in the epilogue (identical values: max commutes with the monotone
 * bias / ReLU epilogue); the full-resolution output is not written.
 * Cin % 16 == 0, Cout % 32 == 0, Cout <= 128 or a multiple of 128. */
int naws_conv3x3_nhwc_f16x2_fwd(const float* X, const void* W2, const float* scaleW,
                                const float* bias, int N, int H, int W, int Cin, int Cout,
                                int dilation, int relu, float* Y, const uint32_t* amax_in,
                                float in_mul, float in_add, uint32_t* amax_out, int amax_out_zeroed,
                                int pool2, void* stream);
/* out[0] = bit pattern of max|X[0..n)| (non-negative floats order like unsigned words). */
int naws_amax_f32(const float* X, int64_t n, uint32_t* out, void* stream);
/* RoIPoolF + RoIFeatureBoost (as naws_roi_pool_f_fwd, NHWC; detectron/ops/roi_pool_f_op.cu:14-127,
 * roi_feature_boost_op.cc:8-66) written straight as the fp16x2 operand of the fc6 GEMM: planes
 * f16 [2][K/16][R][16] with K = C*pooled_h*pooled_w holding hi / lo of Y[r][:] * s_r, scales fp32
 * [2][R] with [1] = 1/s_r.  s_r comes from the bound max|Y[r]| <= max|X[n]| * |boost[r]|:
 * amax_words[n] = bit pattern of an upper bound of max|X| of image n (n_words entries, the last one
 * is used for higher batch indices), e.g. the amax_out of the conv body's last layer.
 * C % 64 == 0, pooled_h*pooled_w <= 256, K % 32 == 0. */
int naws_roi_pool_f_f16x2_fwd(const float* X, int N, int C, int H, int W, const float* rois, int R,
                              const float* boost, int pooled_h, int pooled_w, float spatial_scale,
                              const uint32_t* amax_words, int n_words, void* planes, float* scales,
                              void* stream);
/* Hierarchical forms of RoIPoolF (+ boost) on NHWC features (same operator, same values bit for
 * bit: reference detectron/modeling/detector.py:319-329, detectron/ops/roi_loop_pool_op.cu:31-101):
 * the bin maxima are taken over precomputed 2x2 / 4x4 block maxima (max is idempotent), which cuts
 * the window gather ~8x.  workspace: naws_roi_pool_workspace_floats(N, C, H, W) floats (the two
 * block-maxima maps, built by the call).  C % 64 == 0; no argmax.
 * naws_roi_pool_f_nhwc_hier_fwd writes Y fp32 [R][C][ph][pw]; naws_roi_pool_f_f16x2_hier_fwd the
 * fp16x2 operand planes exactly as naws_roi_pool_f_f16x2_fwd. */
int64_t naws_roi_pool_workspace_floats(int N, int C, int H, int W);
int naws_roi_pool_f_nhwc_hier_fwd(const float* X, int N, int C, int H, int W, const float* rois,
                                  int R, const float* boost, int pooled_h, int pooled_w,
                                  float spatial_scale, float* workspace, float* Y, void* stream);
int naws_roi_pool_f_f16x2_hier_fwd(const float* X, int N, int C, int H, int W, const float* rois,
                                   int R, const float* boost, int pooled_h, int pooled_w,
                                   float spatial_scale, const uint32_t* amax_words, int n_words,
                                   float* workspace, void* planes, float* scales, void* stream);
/* The two halves of naws_roi_pool_f_f16x2_hier_fwd as separate calls: naws_roi_maxmaps_fwd builds
 * the 2x2 / 4x4 block-maxima maps M2 / M4 (each [N][H][W][C] floats, 16-byte aligned, C % 4 == 0)
 * of N images - e.g. one image at a time on that image's own stream -, and
 * naws_roi_pool_f_f16x2_mapped_fwd pools over maps that already exist.  Same operator and values
 * (detectron/modeling/detector.py:319-329, detectron/ops/roi_loop_pool_op.cu:31-101). */
int naws_roi_maxmaps_fwd(const float* X, int N, int C, int H, int W, float* M2, float* M4,
                         void* stream);
int naws_roi_pool_f_f16x2_mapped_fwd(const float* X, int N, int C, int H, int W, const float* rois,
                                     int R, const float* boost, int pooled_h, int pooled_w,
                                     float spatial_scale, const uint32_t* amax_words, int n_words,
                                     const float* M2, const float* M4, void* planes, float* scales,
                                     void* stream);
/* ... for rois [r_first, r_first + count) of R_total only (rois, boost, planes, scales: those of all
 * R_total rois): one image's proposals, pooled on that image's stream at the tail of its conv
 * chain.  Launches over disjoint ranges write disjoint rows of the planes; together they equal
 * the one launch bit for bit. */
int naws_roi_pool_f_f16x2_mapped_range_fwd(const float* X, int N, int C, int H, int W,
                                           const float* rois, int R_total, int r_first, int count,
                                           const float* boost, int pooled_h, int pooled_w,
                                           float spatial_scale, const uint32_t* amax_words,
                                           int n_words, const float* M2, const float* M4,
                                           void* planes, float* scales, void* stream);
/* Q f16 [2][Rpad/16][K][16] = transposition of P f16 [2][K/16][R][16] (Rpad = R rounded up to 32,
 * rows >= R zero): the K(=rois)-contiguous form of the same scaled matrix, B operand of
 * fc6's dW = dY^T X with a scale vector of ones, provided dY is split by
 * naws_split_f16x2_kscaled with rowmul = 1/s_r. */
int naws_f16_planes_transpose(const void* P, int R, int K, int Rpad, void* Q, void* stream);
/* The bf16 plan's forms (round 4): RoIPoolF (+ boost) over existing block-maxima maps written as
 * fc6's forward operand - ONE plane P[K/16][R][16] of the pooled features rounded to bf16, the
 * layout of naws_to_bf16_slab (K = C * ph * pw, K % 64 == 0; reference operator: RoILoopPool +
 * RoIFeatureBoost, detectron/ops/roi_loop_pool_op.cu:31-101) - and its transposition
 * Q[Rpad/16][K][16] (Rpad = R rounded up to 64, rows >= R zero), the operand of dW = dY^T X. */
int naws_roi_pool_f_bf16_slab_mapped_fwd(const float* X, int N, int C, int H, int W, const float* rois,
                                         int R, const float* boost, int pooled_h, int pooled_w,
                                         float spatial_scale, const float* M2, const float* M4,
                                         void* P, void* stream);
int naws_bf16_slab_transpose(const void* P, int R, int K, int Rpad, void* Q, void* stream);
/* naws_split_f16x2 of diag(rowmul) X (rowmul[rows], shared by the batch items; nullable). */
int naws_split_f16x2_kscaled(const float* X, int batch, int rows, int cols, int ld, int64_t strideX,
                             int transpose, int kpad, void* P, float* scales, const float* rowmul,
                             void* stream);
/* bf16 plan on the same pipeline: one plane (operands rounded to bf16), 64-deep K-steps.
 * naws_to_bf16_slab: as naws_split_bf16x3 with a single plane, P[batch][kpad/16][outer][16],
 * kpad = K rounded up to 64.  naws_gemm_bf16_slab_nt: C (+)= A B^T on such operands, K % 64 == 0;
 * replaces naws_gemm_bf16_nt for fc6 / fc7 of the bf16 plan (same reference operators). */
int naws_to_bf16_slab(const float* X, int batch, int rows, int cols, int ld, int64_t strideX,
                      int transpose, int kpad, void* P, void* stream);
int naws_gemm_bf16_slab_nt(int M, int N, int K, const void* A, int64_t slabA, const void* B,
                           int64_t slabB, float* C, int ldc, int batch, int64_t strideA,
                           int64_t strideB, int64_t strideC, int epilogue, const float* bias,
                           int64_t strideBias, const float* aux, int ldaux, float alpha,
                           float drop_ratio, uint64_t seed, int accumulate, void* stream);
/* naws_gemm_bf16_slab_nt with the ACM SGD update of `param` in place of the store (round 4; the
 * bf16 plan's fc6_w gradient at one process - the reference adds its all-reduce ops only for
 * NUM_GPUS > 1, optimizer_wsl.py:52-72, and the update operator is
 * acm_weightdecay_momentum_sgd_op.h:48-112): the product A B^T [M x N] is the gradient of
 * param [M][ldp] and is never written; param / momentum_buf are updated in place exactly as
 * naws_acm_sgd_update_planes (format NAWS_PLANES_BF16) would from that gradient, and the updated
 * rows are rounded into P, param's bf16 operand plane [N/16][prows][16] (P points at this block's
 * first row).  N % 16 == 0, K % 64 == 0, ldp % 4 == 0. */
int naws_gemm_bf16_slab_nt_sgd(int M, int N, int K, const void* A, int64_t slabA, const void* B,
                               int64_t slabB, float* momentum_buf, float* param, int ldp,
                               const float* lr, float lr_mult, float weight_decay, float momentum,
                               int nesterov, int gpu_num, int64_t iter_count, void* P, int prows,
                               void* stream);
/* 3x3 / stride 1 / pad = dilation convolution, NHWC fp32 in and out, as an fp32x3 implicit GEMM
 * (same operator as naws_conv3x3_nhwc_fwd: Caffe2 Conv + Relu, reference
 * detectron/modeling/VGG16.py:37-130).  W3 = naws_split_bf16x3 (transpose = 0) of the packed
 * [Cout][3][3][Cin] weight viewed as [Cout][9*Cin].  Cin % 16 == 0. */
int naws_conv3x3_nhwc_f32x3_fwd(const float* X, const void* W3, const float* bias, int N, int H,
                                int W, int Cin, int Cout, int dilation, int relu, float* Y,
                                void* stream);
/* The same convolution (dilation 1) followed by the reference's 2x2 / stride-2 MaxPool (pool1 ..
 * pool3, detectron/modeling/VGG16.py:14-30), taken in the kernel's epilogue: Y is
 * [N][H/2][W/2][Cout], bit-identical to naws_conv3x3_nhwc_f32x3_fwd + naws_maxpool2x2_nhwc_fwd.
 * Cout % 64 == 0, Cout <= 256. */
int naws_conv3x3_nhwc_f32x3_pool_fwd(const float* X, const void* W3, const float* bias, int N, int H,
                                     int W, int Cin, int Cout, int relu, float* Y, void* stream);
/* Winograd F(2x2,3x3) convolution (as naws_conv3x3_winograd_nhwc_fwd) with the 16 batched GEMMs
 * in the fp32x3 split; U3 = naws_split_bf16x3 (batch 16, transpose 0) of the transformed weight
 * U[16][Cout][Cin].  Cin % 16 == 0. */
int64_t naws_winograd_f32x3_workspace_floats(int N, int H, int W, int Cin, int Cout, int dilation);
int naws_conv3x3_winograd_nhwc_f32x3_fwd(const float* X, const void* U3, const float* bias, int N,
                                         int H, int W, int Cin, int Cout, int dilation, int relu,
                                         float* workspace, float* Y, void* stream);
/* ---- inference post-processing ------------------------------------------------
 * Greedy NMS for `batch` independent box lists (= the classes of one image) in one call.
 * replaces: detectron/utils/cython_nms.pyx:36-87 `nms`, called per class from
 * detectron/core/test_wsl.py:803-863.  boxes[batch][n_max][4] = (x1,y1,x2,y2) of each list's
 * candidates ALREADY in visiting order (descending score), counts[batch] = candidates per list.
 * keep[batch][n_max] <- 1 where the box survives (0 beyond counts[b]).  Same fp32 arithmetic and
 * comparison (suppress when IoU >= thresh, +1 pixel areas) as the reference loop.
 * workspace: naws_nms_workspace_bytes(batch, n_max) bytes, 8-byte aligned.  n_max <= 16384. */
int64_t naws_nms_workspace_bytes(int batch, int n_max);
/* Soft-NMS for `batch` independent detection lists (= the classes of one image) in one call.
 * replaces: detectron/utils/cython_nms.pyx:98-203 `soft_nms` (boxes.py:321-338), the
 * TEST.SOFT_NMS branch of core/test_wsl.py:826-834.  dets[batch][n_max][5] = (x1,y1,x2,y2,score)
 * in the caller's order (the reference passes them in proposal order), counts[batch].  method:
 * 0 hard, 1 linear, 2 gaussian; overlap_thresh = Nt, score_thresh = the discard threshold.
 * out_dets[batch][n_max][5] / keep[batch][n_max] (original indices) / out_counts[batch]: the
 * surviving detections with their decayed scores IN THE REFERENCE'S OUTPUT ORDER (its in-place
 * swap / overwrite-by-the-last-box sequence is replayed).  n_max <= 5111 (the list lives in LDS). */
int naws_soft_nms_fwd(const float* dets, const int32_t* counts, int batch, int n_max, float sigma,
                      float overlap_thresh, float score_thresh, int method, float* out_dets,
                      int32_t* keep, int32_t* out_counts, void* stream);
int naws_nms_sorted_fwd(const float* boxes, const int32_t* counts, int batch, int n_max,
                        float thresh, void* workspace, int32_t* keep, void* stream);
/* ---- loader image preparation --------------------------------------------------
 * One decoded image (uint8 H x W x 3, BGR, on the device) -> its slot of the NCHW batch blob:
 * optional horizontal flip, crop (taken on the flipped image, rows crop_y0.., cols crop_x0..),
 * optional HSV saturation / exposure jitter (distort != 0: cv2 8-bit BGR2HSV, S = min(saturation*S,
 * 255), V = min(exposure*V, 255), uint8 truncation, HSV2BGR - minibatch_wsl.py:127-138),
 * float32 (v - means[c]) / stds[c], bilinear resize by im_scale with cv2.resize(INTER_LINEAR)
 * semantics to out_h x out_w (= cvRound(crop_h * im_scale), cvRound(crop_w * im_scale), computed
 * by the caller), written to out[c * plane_stride + y * row_stride + x].
 * replaces: detectron/roi_data/minibatch_wsl.py:121-157, detectron/utils/blob.py:67-131.
 * means3 / stds3 are HOST pointers (3 floats each). */
int naws_prep_image_fwd(const uint8_t* im_bgr_hwc, int H, int W, int flip, int crop_y0,
                        int crop_x0, int crop_h, int crop_w, const float* means3,
                        const float* stds3, double im_scale, int distort, float saturation,
                        float exposure, int out_h, int out_w, int64_t plane_stride,
                        int row_stride, float* out, void* stream);
/* ---- adjacent loss op -----------------------------------------------------------
 * MinEntropyLoss / MinEntropyLossGradient (cfg.WSL.MIN_ENTROPY_LOSS), detectron/ops/
 * min_entropy_loss_op.cc:7-98: X [N,C] probabilities, L [1,C] labels (B must be 1, :30),
 * Y[0] = -sum_{n, L[c]>=0.5} p log p / count with p = max(X, 1e-20);
 * dX = min(dY[0]/count * (-1 - log p), 1e4) on those entries, 0 elsewhere. */
int naws_min_entropy_loss_fwd(const float* X, const float* L, int N, int C, float* Y, void* stream);
int naws_min_entropy_loss_bwd(const float* X, const float* L, const float* dY, int N, int C,
                              float* dX, void* stream);

/* ------------------------------------------------------------------------ *
 * f-4  OICR refinement operators (WSL.OICR: wsl_heads.py:134-156, :512-560) and the mining gate's
 *      RoIEntropy (webly_heads.py:219-262).
 *
 * naws_roi_label_fwd — RoILabel, ref: detectron/ops/roi_label_op.cc:10-123, schema :133-145.
 *   S fp32 [n][cs] scores (cs == c, or c + 1 with a background column 0), U fp32 [n][n] IoU,
 *   L fp32 [c] image labels, CW fp32 [c] class weights (nullable) -> RL int32 [n], RW fp32 [n];
 *   stats fp32 [4] += {fg rois, bg rois, fg weight, bg weight} (the op's display counters).
 *   Errors: cs not in {c, c+1} -> SHAPE (ENFORCE :19); num_pos / num_neg < n -> UNSUPPORTED (a
 *   binding cap makes the result depend on the reference's time-seeded shuffle, :62-70).
 *   workspace: naws_roi_label_workspace_bytes(n, c, top_k) bytes.
 * naws_softmax_with_loss_n_fwd / _bwd — SoftmaxWithLossN(+Gradient), label mode, ref:
 *   detectron/ops/softmax_with_loss_n_op.cc:152-263 / :265-357.  X fp32 [N][D], T int32 [N], W fp32
 *   [N] (nullable) -> P fp32 [N][D], loss fp32 [1] = scale * sum(-w log P[i][T_i]) / sum(w) (0 when
 *   sum(w) == 0).  bwd: dX = (P - onehot) * w * scale / total * dloss[0], total = #{w > 1e-12} with
 *   weights, N without.  A label outside [0, D) (the ENFORCE at :192) makes the loss NaN.
 *   workspace: naws_softmax_with_loss_n_workspace_floats(N) floats.
 * naws_roi_entropy_fwd — RoIEntropy, ref: detectron/ops/roi_entropy_op.cu:24-112.  S fp32 [n]
 *   scores, C fp32 [n] class ids -> E fp32 [num_classes] = 1 - normalised entropy per class; mean
 *   (nullable) is the op's running accumulator (zeroed first when init != 0).
 * ------------------------------------------------------------------------ */
int64_t naws_roi_label_workspace_bytes(int n, int c, int top_k);
int naws_roi_label_fwd(const float* S, const float* U, const float* L, const float* CW, int n, int cs,
                       int c, float fg_thresh, float bg_thresh_hi, float bg_thresh_lo, int top_k,
                       int num_pos, int num_neg, void* workspace, int32_t* RL, float* RW,
                       float* stats, void* stream);
int64_t naws_softmax_with_loss_n_workspace_floats(int N);
int naws_softmax_with_loss_n_fwd(const float* X, const int32_t* T, const float* W, int N, int D,
                                 float scale, float* workspace, float* P, float* loss, void* stream);
int naws_softmax_with_loss_n_bwd(const int32_t* T, const float* W, const float* P,
                                 const float* dloss, int N, int D, float scale, float* workspace,
                                 float* dX, void* stream);
int naws_roi_entropy_fwd(const float* S, const float* C, int n, int num_classes, int rm_bg, float* E,
                         float* mean, int init, void* stream);

/* ------------------------------------------------------------------------ *
 * f-2  Inference post-processing on the GPU (BASELINE configs[4]: multi-scale TTA).
 *
 * naws_roi_dedup_fwd — per TTA pass p (one workgroup each): project the image boxes into the pass's
 *   input frame, de-duplicate them on the DEDUP_BOXES grid and emit the network's roi blobs.
 *   ref: detectron/core/test_wsl.py:998-1026 (float64 product, float32 blob), :293-301 (flip),
 *   :125-133 (hash + np.unique), :1058 (obn + 1).
 *   boxes fp32 [n][4] image coordinates, obn_scores fp32 [n]; passes: DEVICE array of npass
 *   naws_dedup_pass records.  Outputs per pass (stride n): rois_out fp32 [npass][n][5] (the unique
 *   rois in np.unique's ascending-hash order, column 0 = batch_index), obn_out fp32 [npass][n],
 *   index_out int32 [npass][n] (first occurrence of each unique roi), inv_out int32 [npass][n]
 *   (scores_unique[inv] = scores in proposal order), count_out int32 [npass].  n <= 16384
 *   (NAWS_ERR_UNSUPPORTED beyond).  Caller's contract (the pass records live on the device, the
 *   entry cannot check it): every projected coordinate c satisfies |c| * dedup_boxes < 562, so
 *   that the hash stays below 2^49 inside the 64-bit sort key (hash << 14 | index); the reference
 *   default 1/16 allows 8992 px.  naws_hip.ops / core/test_wsl.py fall back to the host path
 *   otherwise.
 * naws_tta_accumulate — acc[n][k] (=, when first) += scores[inv_index[r]][:] (inv_index nullable =
 *   identity): the scatter-back of :173-176 fused with the running sum of np.mean (:260-261).
 * naws_tta_finish — acc = float32(double(acc) / npass) (numpy's mean division).
 * naws_det_limit_fwd — the image-wide DETECTIONS_PER_IM cut of :849-863 after the per-class NMS
 *   (naws_nms_sorted_fwd): scores fp32 [R][K] (K = C + 1, column 0 background), keep uint8 [C][R];
 *   limit == 0 keeps everything.  Outputs (class ascending, row ascending - the reference's
 *   order): out_count int32 [1] (may exceed cap: then only cap triples were written), out_cls /
 *   out_row int32 [cap], out_score fp32 [cap].
 * ------------------------------------------------------------------------ */
typedef struct naws_dedup_pass {
  double im_scale;      /* target_scale / short side (capped by max size), as the host computes it */
  float im_width;       /* width of the (un-scaled) image: used when flip != 0 */
  int32_t flip;         /* horizontally mirrored pass */
  float batch_index;    /* column 0 of the emitted rois */
} naws_dedup_pass;
int naws_roi_dedup_fwd(const float* boxes, const float* obn_scores, int n, int npass,
                       const void* passes, float dedup_boxes, float* rois_out, float* obn_out,
                       int32_t* index_out, int32_t* inv_out, int32_t* count_out, void* stream);
int naws_tta_accumulate(const float* scores, const int32_t* inv_index, int n, int k, int first,
                        float* acc, void* stream);
int naws_tta_finish(float* acc, int64_t total, int npass, void* stream);
int naws_det_limit_fwd(const float* scores, const uint8_t* keep, int C, int R, int K, int limit,
                       int cap, int32_t* out_count, int32_t* out_cls, int32_t* out_row,
                       float* out_score, void* stream);

/* fc6 weight gradient reading the pooled features in their FORWARD operand layout (no
 * naws_f16_planes_transpose copy): C [M x N] (ld ldc) = sum_k A[k][m] X[k][n], A2 = planes
 * [2][K/16][M][16] (a transposing naws_split_f16x2* of the output gradient, K % 32 == 0, zero
 * beyond the xrows valid rows) with per-row factors scaleA; X2 = planes [2][N/16][xrows][16] as
 * naws_roi_pool_f_f16x2_fwd / naws_split_f16x2 write them, slabX = elements between 16-column
 * blocks, scaleX per column (NULL: ones).  Replaces: Caffe2 FCGradient's dW for fc6 (reference
 * detectron/modeling/wsl_heads.py:674-679); same accumulation order as naws_gemm_f32_f16x2_nt on
 * the transposed copy - bit-identical results. */
int naws_gemm_f32_f16x2_nt_xk(int M, int N, int K, const void* A2, int64_t slabA, int64_t planeA,
                              const float* scaleA, const void* X2, int64_t slabX, int64_t planeX,
                              int xrows, const float* scaleX, float* C, int ldc, void* stream);

/* naws_gemm_f32_f16x2_nt_xk with the ACM SGD update of the [M x N] parameter block `param`
 * (ld ldp; `momentum_buf` its momentum, same indexing) in the epilogue INSTEAD of the store: the
 * gradient never reaches memory.  For a run without a gradient exchange - one process; the
 * reference adds its all-reduce ops only for NUM_GPUS > 1 (detectron/modeling/
 * optimizer_wsl.py:52-72) - and ITER_SIZE 1.  Element arithmetic (lr[0] * lr_mult, weight_decay,
 * 1 / gpu_num, iter_count == 0 starts the momentum at zero), the block's fp16x2 operand planes
 * (`planes`: hi at planes, lo at planes + plane_stride, [N/16][plane_rows][16], plane_rows >= M),
 * bound / rowmax / inv_scale / overflow exactly as naws_acm_sgd_update_f16x2 treats a region:
 * bit-identical parameters, momentum and planes.  Hand that call the same region with planes =
 * NULL for the rest of the arena.
 *   ref: Caffe2 FCGradient dW (detectron/modeling/wsl_heads.py:674-679) +
 *        detectron/ops/acm_weightdecay_momentum_sgd_op.h:72-109. */
int naws_gemm_f32_f16x2_nt_xk_sgd(int M, int N, int K, const void* A2, int64_t slabA, int64_t planeA,
                                  const float* scaleA, const void* X2, int64_t slabX,
                                  int64_t planeX, int xrows, const float* scaleX, float* param,
                                  float* momentum_buf, int ldp, const float* lr, float lr_mult,
                                  float weight_decay, float momentum, int nesterov, int gpu_num,
                                  int64_t iter_count, void* planes, int64_t plane_stride,
                                  int plane_rows, const uint32_t* bound, uint32_t* rowmax,
                                  float* inv_scale, int32_t* overflow, int32_t overflow_tag,
                                  void* stream);

/* ---- process-wide state and tuning (no reference counterpart) ------------------------------ */
/* Kernels that need more than 64 KB of dynamic LDS have that limit raised once per (kernel,
 * device) pair; the library remembers which pairs are done.  After hipDeviceReset() - which
 * drops the attribute - call this to make it forget.  Always safe to call. */
int naws_launch_state_reset(void);
/* Select a tile / pipeline form for the A/B tools (na-fwebsod_amd/tools/ab_*.py): knob in
 * {"gemm", "x3", "h2", "conv_ring", "conv_bn", "roi_nw", "wino"}; a form may change the
 * fp32 accumulation order (last bits), never the arithmetic.
 * Unknown knob: NAWS_ERR_ARG.  The library never reads the environment. */
int naws_set_variant(const char* knob, int value);
/* A stream for background work (the parameter update that runs beside the next iteration's conv
 * body): priority 0 = normal, > 0 = lower than normal, < 0 = higher (clamped to the device's
 * range); cu_mask (mask_words 32-bit words, bit i = compute unit i in the runtime's numbering;
 * NULL / 0 = all) confines its kernels to those compute units so an HBM-bound kernel does not take
 * wave slots from an MFMA-bound one.  The handle is a hipStream_t; release it with
 * naws_stream_destroy.  No reference counterpart (Caffe2 runs one stream per GPU). */
int naws_stream_create(int priority, const uint32_t* cu_mask, int mask_words, void** stream);
/* MEASUREMENT AID, not on the product path (bench.py --emulate-exchange): `cus` workgroups copy
 * `bytes` (a multiple of 16) from src to dst at an aggregate pace of gbytes_per_sec - what an RCCL
 * ring all-reduce does to a rank's compute units and HBM for the duration the xGMI links allow,
 * so that a one-GPU box can show what the exchange costs the kernels it runs beside.  Replaces
 * nothing in the reference (its exchange is NCCLAllreduce, detectron/modeling/optimizer_wsl.py:52-72). */
int naws_emulate_exchange(const void* src, void* dst, int64_t bytes, int cus, float gbytes_per_sec,
                          void* stream);
int naws_stream_destroy(void* stream);

#ifdef __cplusplus
}
#endif
#endif /* NAWS_H_ */

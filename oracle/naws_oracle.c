/*
 * naws_oracle.c — CPU restatement of the NA-fWebSOD hot-path operators.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is shipped or measured as
 * the product: only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library, and only as the checker / the timed
 * CPU baseline.
 *
 * PARITY PINNING (see oracle/README.md): the reference's own tests hold no
 * golden vectors for this path and the reference's C++/CUDA ops
 * cannot be built here (they need the un-vendored Caffe2 v1.3.0 headers), so
 * the operator semantics below are "parity unpinned" by the reference; they are
 * pinned instead by hand-derived known-answer tests (tests/test_oracle_kat.py),
 * by the one reference output recorded in SURVEY.md §8c (WeightedCrossEntropy
 * = 0.206650317), and — for the pure-Python reference pieces (lr policy, roi
 * projection, roi sampling) — by fixtures generated from the imported
 * reference (tests/golden/make_golden_from_reference.py).
 *
 * Every function cites the reference file:line it follows (paths relative to
 * the upstream repository root).  Plain scalar C, single thread, fp32
 * arithmetic in the reference's order of operations.  Build: oracle/Makefile
 * (gcc -O2 -ffp-contract=off so no FMA is introduced).
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define IMAX(a, b) ((a) > (b) ? (a) : (b))
#define IMIN(a, b) ((a) < (b) ? (a) : (b))

/* ---------------------------------------------------------------------------
 * RoIPoolF forward.  ref: detectron/ops/roi_loop_pool_op.cu:31-101 is the
 * in-tree copy of the Caffe2 kernel; we follow it WITHOUT the inner-rectangle
 * skip (:80-85, RoILoopPool only) and WITH the original empty-bin rule kept in
 * the comment at :72 (maxval = is_empty ? 0 : -FLT_MAX), i.e. Caffe2
 * modules/detectron/roi_pool_f_op.cu.  X is NCHW; rois [R,5]; Y [R,C,PH,PW];
 * argmax may be NULL.
 * ------------------------------------------------------------------------- */
void oracle_roi_pool_f(const float* X, int N, int C, int H, int W, const float* rois, int R,
                       int PH, int PW, float spatial_scale, float* Y, int32_t* argmax) {
  (void)N;
  for (int n = 0; n < R; ++n) {
    const float* roi = rois + (size_t)n * 5;
    int roi_batch_ind = (int)roi[0];
    int roi_start_w = (int)roundf(roi[1] * spatial_scale);
    int roi_start_h = (int)roundf(roi[2] * spatial_scale);
    int roi_end_w = (int)roundf(roi[3] * spatial_scale);
    int roi_end_h = (int)roundf(roi[4] * spatial_scale);
    int roi_width = IMAX(roi_end_w - roi_start_w + 1, 1);
    int roi_height = IMAX(roi_end_h - roi_start_h + 1, 1);
    float bin_size_h = (float)roi_height / (float)PH;
    float bin_size_w = (float)roi_width / (float)PW;
    for (int c = 0; c < C; ++c) {
      const float* xc = X + ((size_t)roi_batch_ind * C + c) * H * W;
      for (int ph = 0; ph < PH; ++ph) {
        for (int pw = 0; pw < PW; ++pw) {
          int hstart = (int)floorf((float)ph * bin_size_h);
          int wstart = (int)floorf((float)pw * bin_size_w);
          int hend = (int)ceilf((float)(ph + 1) * bin_size_h);
          int wend = (int)ceilf((float)(pw + 1) * bin_size_w);
          hstart = IMIN(IMAX(hstart + roi_start_h, 0), H);
          hend = IMIN(IMAX(hend + roi_start_h, 0), H);
          wstart = IMIN(IMAX(wstart + roi_start_w, 0), W);
          wend = IMIN(IMAX(wend + roi_start_w, 0), W);
          int is_empty = (hend <= hstart) || (wend <= wstart);
          float maxval = is_empty ? 0.f : -FLT_MAX;
          int maxidx = -1;
          for (int h = hstart; h < hend; ++h)
            for (int w = wstart; w < wend; ++w) {
              int idx = h * W + w;
              if (xc[idx] > maxval) {
                maxval = xc[idx];
                maxidx = idx;
              }
            }
          size_t o = (((size_t)n * C + c) * PH + ph) * PW + pw;
          Y[o] = maxval;
          if (argmax) argmax[o] = maxidx;
        }
      }
    }
  }
}

/* RoIFeatureBoost fwd (and, identically, its gradient).
 * ref: detectron/ops/roi_feature_boost_op.cc:8-35 (:37-66 gradient). */
void oracle_roi_feature_boost(const float* X, const float* S, int R, int F, float* Y) {
  for (int b = 0; b < R; ++b)
    for (int f = 0; f < F; ++f) Y[(size_t)b * F + f] = X[(size_t)b * F + f] * S[b];
}

/* RoIIoU.  ref: detectron/ops/roi_iou_op.cu:27-62.  idx = j*n + i. */
static float iou_pair(const float* Rd, int i, int j) {
  int ixmin = (int)Rd[i * 5 + 1], iymin = (int)Rd[i * 5 + 2];
  int ixmax = (int)Rd[i * 5 + 3], iymax = (int)Rd[i * 5 + 4];
  int jxmin = (int)Rd[j * 5 + 1], jymin = (int)Rd[j * 5 + 2];
  int jxmax = (int)Rd[j * 5 + 3], jymax = (int)Rd[j * 5 + 4];
  int xmin = IMAX(ixmin, jxmin), ymin = IMAX(iymin, jymin);
  int xmax = IMIN(ixmax, jxmax), ymax = IMIN(iymax, jymax);
  int w = (int)fmax(xmax - xmin + 1., 0.);
  int h = (int)fmax(ymax - ymin + 1., 0.);
  float inters = (float)(w * h);
  float uni = (float)((ixmax - ixmin + 1.) * (iymax - iymin + 1.) +
                      (jxmax - jxmin + 1.) * (jymax - jymin + 1.) - inters);
  return inters / uni;
}
void oracle_roi_iou(const float* Rd, int n, float* J) {
  for (int j = 0; j < n; ++j)
    for (int i = 0; i < n; ++i) J[(size_t)j * n + i] = (i == j) ? 1.0f : iou_pair(Rd, i, j);
}

/* (Weighted)CrossEntropyWithLogits forward.  W == NULL -> unweighted op.
 * ref: detectron/ops/cross_entropy_wsl_op.cc:7-45 (CE), :87-132 (WCE);
 * thresholds cross_entropy_wsl_op.h:90 (1e-20).  `log` of a float argument
 * resolves to the double overload in the reference's translation unit; the
 * term is then rounded to float by `loss -=`. */
void oracle_wce_fwd(const float* X, const float* L, const float* W, int N, int C, int is_mean,
                    float* Y) {
  float norm = is_mean ? (float)C : 1.f;
  float loss = 0.f;
  for (int i = 0; i < N * C; ++i) {
    float prob = fmaxf(X[i], 1e-20f);
    float one_prob = fmaxf(1 - X[i], 1e-20f);
    if (W)
      loss -= (L[i] * log(prob) + (1 - L[i]) * log(one_prob)) * W[i];
    else
      loss -= (L[i] * log(prob) + (1 - L[i]) * log(one_prob));
  }
  float y = loss / norm;
  Y[0] = y * (float)(1.0 / N);
}

/* ref: detectron/ops/cross_entropy_wsl_op.cc:47-85 (CE grad), :134-180 (WCE
 * grad); the 1e4 clamp is applied BEFORE the weight (.cc:170-173). */
void oracle_wce_bwd(const float* X, const float* L, const float* W, const float* dY, int N, int C,
                    int is_mean, float* dX) {
  float norm = is_mean ? (float)C : 1.f;
  for (int i = 0; i < N * C; ++i) {
    float grad = dY[0];
    float prob = fmaxf(X[i], 1e-20f);
    float one_prob = fmaxf(1 - X[i], 1e-20f);
    float v = fminf(grad * (-1 * L[i] / prob - (-1) * (1 - L[i]) / one_prob) / norm, 1e4f);
    if (W) v = v * W[i];
    dX[i] = v;
  }
  for (int i = 0; i < N * C; ++i) dX[i] = dX[i] * (float)(1.0 / N);
}

/* ACMWeightDecayMomentumSGDUpdate, one parameter blob, one call.
 * ref: detectron/ops/acm_weightdecay_momentum_sgd_op.h:48-112 and :9-33
 * (momentum_sgd_update_mult).  iter_count is the op state before the call;
 * returns the state after it. */
int64_t oracle_acm_sgd(const float* g, float* m, const float* lr, float* p, float* acm, int64_t n,
                       float momentum, int nesterov, float weight_decay, int iter_size,
                       int gpu_num, float lr_mult, int64_t iter_count) {
  if (iter_count == 0) {
    memset(acm, 0, sizeof(float) * (size_t)n);
    memset(m, 0, sizeof(float) * (size_t)n);
  }
  for (int64_t i = 0; i < n; ++i) acm[i] = g[i] + acm[i];
  iter_count += 1;
  if (iter_count % iter_size == 0) {
    float scale = (float)(1.0 / (iter_size * gpu_num));
    for (int64_t i = 0; i < n; ++i) acm[i] = acm[i] * scale;
    for (int64_t i = 0; i < n; ++i) acm[i] = acm[i] + weight_decay * p[i];
    float LR = lr[0] * lr_mult;
    for (int64_t i = 0; i < n; ++i) {
      float ng;
      if (!nesterov) {
        float adj = LR * acm[i] + momentum * m[i];
        m[i] = adj;
        ng = adj;
      } else {
        float mi = m[i];
        float mi_new = momentum * mi + LR * acm[i];
        m[i] = mi_new;
        ng = (1 + momentum) * mi_new - momentum * mi;
      }
      /* the op writes ng into the acmgrad output and updates param with it;
       * the net passes grad in place as OUTPUT_GRAD but the kernel's ng
       * pointer is OUTPUT_ACMGRAD (.h:94-101), which is then zeroed. */
      p[i] -= ng;
      acm[i] = ng;
    }
    memset(acm, 0, sizeof(float) * (size_t)n);
  }
  return iter_count;
}

/* Stat accumulate.  ref: detectron/ops/stat_op.cu:14-20, :37-52. */
void oracle_stat(const float* I, const float* L, int n, int init, float* AI, float* AL) {
  if (init) {
    memset(AI, 0, sizeof(float) * (size_t)n);
    memset(AL, 0, sizeof(float) * (size_t)n);
  }
  for (int i = 0; i < n; ++i) {
    AI[i] = I[i] * L[i] + AI[i];
    AL[i] = L[i] + AL[i];
  }
}

/* Caffe2 Softmax over axis 1 of [rows, cols]: subtract the row max, exp,
 * divide by the row sum (caffe2/operators/softmax_shared.cc, v1.3.0 —
 * third-party, restated from its published algorithm). */
static void softmax_rows(const float* X, int rows, int cols, float* Y) {
  for (int r = 0; r < rows; ++r) {
    const float* x = X + (size_t)r * cols;
    float* y = Y + (size_t)r * cols;
    float m = x[0];
    for (int c = 1; c < cols; ++c) m = fmaxf(m, x[c]);
    float s = 0.f;
    for (int c = 0; c < cols; ++c) {
      y[c] = expf(x[c] - m);
      s += y[c];
    }
    for (int c = 0; c < cols; ++c) y[c] = y[c] / s;
  }
}
static void softmax_rows_grad(const float* Y, const float* dY, int rows, int cols, float* dX) {
  for (int r = 0; r < rows; ++r) {
    const float* y = Y + (size_t)r * cols;
    const float* dy = dY + (size_t)r * cols;
    float d = 0.f;
    for (int c = 0; c < cols; ++c) d += y[c] * dy[c];
    for (int c = 0; c < cols; ++c) dX[(size_t)r * cols + c] = y[c] * (dy[c] - d);
  }
}
static void transpose2d(const float* X, int rows, int cols, float* Y) {
  for (int r = 0; r < rows; ++r)
    for (int c = 0; c < cols; ++c) Y[(size_t)c * rows + r] = X[(size_t)r * cols + c];
}

/* WSDDN outputs for ONE image (the reference's IMS_PER_BATCH == 1 graph).
 * ref: detectron/modeling/wsl_heads.py:51-55 (Softmax, Transpose, Softmax,
 * Transpose, Mul), :227 (ReduceSum axes=[0]); webly_heads.py:57-74 (the
 * residual Add and the same chain for the noise branch).
 * fc8c/fc8d/noisy_* are dense [R,C]; noisy_* NULL -> clean branch only. */
void oracle_wsddn_outputs_fwd(const float* fc8c, const float* fc8d, const float* noisy_fc8c,
                              const float* noisy_fc8d, int R, int C, float* alpha_cls,
                              float* alpha_det, float* rois_pred, float* cls_prob) {
  size_t n = (size_t)R * C;
  float* zc = (float*)malloc(sizeof(float) * n);
  float* zd = (float*)malloc(sizeof(float) * n);
  float* t0 = (float*)malloc(sizeof(float) * n);
  float* t1 = (float*)malloc(sizeof(float) * n);
  for (size_t i = 0; i < n; ++i) {
    zc[i] = noisy_fc8c ? fc8c[i] + noisy_fc8c[i] : fc8c[i];
    zd[i] = noisy_fc8d ? fc8d[i] + noisy_fc8d[i] : fc8d[i];
  }
  softmax_rows(zc, R, C, alpha_cls);
  transpose2d(zd, R, C, t0);          /* [C,R] */
  softmax_rows(t0, C, R, t1);
  transpose2d(t1, C, R, alpha_det);   /* [R,C] */
  for (size_t i = 0; i < n; ++i) rois_pred[i] = alpha_cls[i] * alpha_det[i];
  for (int c = 0; c < C; ++c) {
    float s = 0.f;
    for (int r = 0; r < R; ++r) s += rois_pred[(size_t)r * C + c];
    cls_prob[c] = s;
  }
  free(zc); free(zd); free(t0); free(t1);
}

/* Backward of the chain above for one image and one branch: d_cls_prob [C]
 * -> dzc, dzd [R,C] (gradients w.r.t. the branch's summed logits).
 * Follows the Caffe2 auto-generated gradient ops in reverse order:
 * ReduceSumGradient (broadcast), MulGradient, Transpose, SoftmaxGradient,
 * Transpose, SoftmaxGradient. */
void oracle_wsddn_outputs_bwd(const float* alpha_cls, const float* alpha_det,
                              const float* d_cls_prob, int R, int C, float* dzc, float* dzd) {
  size_t n = (size_t)R * C;
  float* dac = (float*)malloc(sizeof(float) * n);
  float* dad = (float*)malloc(sizeof(float) * n);
  float* t0 = (float*)malloc(sizeof(float) * n);
  float* t1 = (float*)malloc(sizeof(float) * n);
  float* t2 = (float*)malloc(sizeof(float) * n);
  for (int r = 0; r < R; ++r)
    for (int c = 0; c < C; ++c) {
      float g = d_cls_prob[c];
      dac[(size_t)r * C + c] = g * alpha_det[(size_t)r * C + c];
      dad[(size_t)r * C + c] = g * alpha_cls[(size_t)r * C + c];
    }
  softmax_rows_grad(alpha_cls, dac, R, C, dzc);
  transpose2d(alpha_det, R, C, t0);
  transpose2d(dad, R, C, t1);
  softmax_rows_grad(t0, t1, C, R, t2);
  transpose2d(t2, C, R, dzd);
  free(dac); free(dad); free(t0); free(t1); free(t2);
}

/* Spatial entropy gate for ONE image.
 * ref: detectron/modeling/webly_heads.py:265-391 (add_spatial_entropy_weight):
 *   J = RoIIoU(rois); E = ReplaceNaN(-(p*log p)); D = LeakyRelu(J @ E, 0.01);
 *   G = E / D; hatE = E * G; hatE_sum = sum_r hatE;
 *   norm = (log N - log y) * y; v = Clip(hatE_sum / norm, 0, 1);
 *   w_noise = v * (1 - labels_oh); w = 1 - w_noise.
 * Caffe2 defaults: ReplaceNaN value 0, LeakyRelu alpha 0.01; Clip leaves NaN.
 * MatMul accumulates in j order here (the reference's sgemm order is
 * unspecified). */
void oracle_entropy_gate(const float* rois, const float* rois_pred, const float* cls_prob,
                         const float* labels_oh, int R, int C, float* class_weight,
                         float* class_weight_noise, float* hatE_sum, float* hatE_sum_norm) {
  size_t n = (size_t)R * C;
  float* J = (float*)malloc(sizeof(float) * (size_t)R * R);
  float* E = (float*)malloc(sizeof(float) * n);
  oracle_roi_iou(rois, R, J);
  for (size_t i = 0; i < n; ++i) {
    float lg = logf(rois_pred[i]);
    float e = (rois_pred[i] * lg) * -1.0f;
    E[i] = isnan(e) ? 0.f : e;
  }
  float logN = logf((float)R);
  for (int c = 0; c < C; ++c) {
    float s = 0.f;
    for (int r = 0; r < R; ++r) {
      float d = 0.f;
      for (int j = 0; j < R; ++j) d += J[(size_t)r * R + j] * E[(size_t)j * C + c];
      d = d >= 0.f ? d : 0.01f * d;
      float g = E[(size_t)r * C + c] / d;
      s += E[(size_t)r * C + c] * g;
    }
    float y = cls_prob[c];
    float norm = (logN - logf(y)) * y;
    float v = s / norm;
    v = (v < 0.f) ? 0.f : v;
    v = (v > 1.f) ? 1.f : v;
    float bg = 1.0f - labels_oh[c];
    hatE_sum[c] = s;
    hatE_sum_norm[c] = v;
    class_weight_noise[c] = v * bg;
    class_weight[c] = 1.0f - class_weight_noise[c];
  }
  free(J); free(E);
}

/* Greedy NMS.  ref: detectron/utils/cython_nms.pyx:36-87 (`nms`): areas with the +1 pixel
 * convention, boxes visited in the given score order, box j suppressed when
 * inter / (area_i + area_j - inter) >= thresh (fp32 arithmetic, in this operation order).
 * `order` is the visiting order (descending score; the reference takes scores.argsort()[::-1]);
 * suppressed[n] is written (1 = suppressed); returns the number kept. */
int oracle_nms(const float* dets /* [n][5] x1 y1 x2 y2 score */, const int64_t* order, int n,
               float thresh, int32_t* suppressed) {
  int kept = 0;
  for (int i = 0; i < n; ++i) suppressed[i] = 0;
  for (int _i = 0; _i < n; ++_i) {
    const int i = (int)order[_i];
    if (suppressed[i]) continue;
    ++kept;
    const float ix1 = dets[i * 5 + 0], iy1 = dets[i * 5 + 1];
    const float ix2 = dets[i * 5 + 2], iy2 = dets[i * 5 + 3];
    const float iarea = (ix2 - ix1 + 1) * (iy2 - iy1 + 1);
    for (int _j = _i + 1; _j < n; ++_j) {
      const int j = (int)order[_j];
      if (suppressed[j]) continue;
      const float jarea = (dets[j * 5 + 2] - dets[j * 5 + 0] + 1) * (dets[j * 5 + 3] - dets[j * 5 + 1] + 1);
      const float xx1 = ix1 >= dets[j * 5 + 0] ? ix1 : dets[j * 5 + 0];
      const float yy1 = iy1 >= dets[j * 5 + 1] ? iy1 : dets[j * 5 + 1];
      const float xx2 = ix2 <= dets[j * 5 + 2] ? ix2 : dets[j * 5 + 2];
      const float yy2 = iy2 <= dets[j * 5 + 3] ? iy2 : dets[j * 5 + 3];
      float w = xx2 - xx1 + 1, h = yy2 - yy1 + 1;
      w = 0.0f >= w ? 0.0f : w;
      h = 0.0f >= h ? 0.0f : h;
      const float inter = w * h;
      const float ovr = inter / (iarea + jarea - inter);
      if (ovr >= thresh) suppressed[j] = 1;
    }
  }
  return kept;
}

/* Soft-NMS.  ref: detectron/utils/cython_nms.pyx:98-203 (`soft_nms`), statement for statement:
 * in place on a copy of the [n][5] detections; position i takes the first maximum of positions
 * i..N-1 (swap), every later box overlapping it (iw > 0 and ih > 0, +1 pixel convention) has its
 * score multiplied by weight (method 1 linear: 1 - ov if ov > Nt; 2 gaussian: exp(-ov*ov/sigma),
 * np.exp = double exp of the float quotient stored back to float; else hard: 0 if ov > Nt) and,
 * when the new score is below `threshold`, is overwritten by box N-1 (N shrinks, the position is
 * examined again).  All cdef variables are C float; `ua = float(...)` is a double conversion of a
 * float expression stored to a float.  Outputs boxes[:N] and inds[:N]; returns N. */
int oracle_soft_nms(const float* boxes_in, int n, float sigma, float Nt, float threshold,
                    int method, float* boxes /* [n][5] */, int64_t* inds /* [n] */) {
  unsigned int N = (unsigned int)n;
  for (int i = 0; i < n * 5; ++i) boxes[i] = boxes_in[i];
  for (int i = 0; i < n; ++i) inds[i] = i;
  const unsigned int N0 = N;                 /* `for i in range(N)`: the bound is taken once */
  for (unsigned int i = 0; i < N0; ++i) {
    float maxscore = boxes[i * 5 + 4];
    unsigned int maxpos = i;
    float tx1 = boxes[i * 5 + 0], ty1 = boxes[i * 5 + 1], tx2 = boxes[i * 5 + 2];
    float ty2 = boxes[i * 5 + 3], ts = boxes[i * 5 + 4];
    int64_t ti = inds[i];
    unsigned int pos = i + 1;
    while (pos < N) {                        /* get max box */
      if (maxscore < boxes[pos * 5 + 4]) { maxscore = boxes[pos * 5 + 4]; maxpos = pos; }
      pos = pos + 1;
    }
    for (int c = 0; c < 5; ++c) boxes[i * 5 + c] = boxes[maxpos * 5 + c];
    inds[i] = inds[maxpos];
    boxes[maxpos * 5 + 0] = tx1; boxes[maxpos * 5 + 1] = ty1; boxes[maxpos * 5 + 2] = tx2;
    boxes[maxpos * 5 + 3] = ty2; boxes[maxpos * 5 + 4] = ts;
    inds[maxpos] = ti;
    tx1 = boxes[i * 5 + 0]; ty1 = boxes[i * 5 + 1]; tx2 = boxes[i * 5 + 2];
    ty2 = boxes[i * 5 + 3]; ts = boxes[i * 5 + 4];
    pos = i + 1;
    while (pos < N) {
      const float x1 = boxes[pos * 5 + 0], y1 = boxes[pos * 5 + 1];
      const float x2 = boxes[pos * 5 + 2], y2 = boxes[pos * 5 + 3];
      const float area = (x2 - x1 + 1) * (y2 - y1 + 1);
      const float iw = ((tx2 < x2 ? tx2 : x2) - (tx1 > x1 ? tx1 : x1) + 1);
      if (iw > 0) {
        const float ih = ((ty2 < y2 ? ty2 : y2) - (ty1 > y1 ? ty1 : y1) + 1);
        if (ih > 0) {
          const float ua = (float)(double)((tx2 - tx1 + 1) * (ty2 - ty1 + 1) + area - iw * ih);
          const float ov = iw * ih / ua;
          float weight;
          if (method == 1) weight = ov > Nt ? 1 - ov : 1;
          else if (method == 2) weight = (float)exp((double)(-(ov * ov) / sigma));
          else weight = ov > Nt ? 0 : 1;
          boxes[pos * 5 + 4] = weight * boxes[pos * 5 + 4];
          if (boxes[pos * 5 + 4] < threshold) {
            for (int c = 0; c < 5; ++c) boxes[pos * 5 + c] = boxes[(N - 1) * 5 + c];
            inds[pos] = inds[N - 1];
            N = N - 1;
            pos = pos - 1;                   /* (unsigned wrap at pos = 0 cannot happen: pos > i) */
          }
        }
      }
      pos = pos + 1;
    }
  }
  return (int)N;
}

/* MinEntropyLoss.  ref: detectron/ops/min_entropy_loss_op.cc:7-45 (forward), :47-98 (gradient).
 * X [N,C] probabilities, L [1,C] image labels: over rows n and the classes with L[c] >= 0.5,
 * loss = -sum p log p / norm, p = max(X, 1e-20), norm = number of terms (fp32 serial sum in
 * (n, c) order).  dX = min(dY/norm * (-1 - log p), 1e4) on those entries, 0 elsewhere. */
float oracle_min_entropy_fwd(const float* X, const float* L, int N, int C) {
  float loss = 0;
  int norm = 0;
  for (int n = 0; n < N; ++n)
    for (int c = 0; c < C; ++c) {
      if (L[c] < 0.5) continue;
      float prob = X[n * C + c] > 1e-20f ? X[n * C + c] : 1e-20f;
      loss -= (prob * logf(prob));
      norm += 1;
    }
  return loss / norm;
}

void oracle_min_entropy_bwd(const float* X, const float* L, float dY, int N, int C, float* dX) {
  int norm = 0;
  for (int n = 0; n < N; ++n)
    for (int c = 0; c < C; ++c)
      if (!(L[c] < 0.5)) norm += 1;
  const float scale = dY / norm;
  for (int n = 0; n < N; ++n)
    for (int c = 0; c < C; ++c) {
      dX[n * C + c] = 0.f;
      if (L[c] < 0.5) continue;
      float prob = X[n * C + c] > 1e-20f ? X[n * C + c] : 1e-20f;
      float g = scale * (-1 + (-1) * (float)logf(prob));
      dX[n * C + c] = g < 1e4f ? g : 1e4f;
    }
}

/* ---------------------------------------------------------------------------
 * f-4: OICR refinement operators (one cfg flag from the hot path: WSL.OICR).
 *
 * RoILabel.  ref: detectron/ops/roi_label_op.cc:10-123 (CPU op; the .cu registers nothing else).
 * S [n, cs] proposal scores (cs = c or c+1 columns: with a background column the class columns
 * start at offset cs - c), U [n, n] IoU matrix, L [1, c] image labels, CW [c] optional class
 * weights.  Per labelled class (L[c] == 1), top_k times: the not-yet-picked proposal with the
 * strictly largest score (first index wins ties; the picked list is shared by all classes, :43-58).
 * Every proposal n then takes the picked proposal of largest IoU U[n, g] (first wins, :77-84):
 * IoU >= fg_thresh -> label class+1, weight = CW[class] or the pick's score; bg_lo <= IoU < bg_hi
 * -> label 0, same weight; otherwise label class+1 with weight 0 (:89-104).
 * The reference visits the proposals in a time-seeded std::random_shuffle order (:62-70), which
 * only matters through the num_pos / num_neg caps; with the op's defaults (9999) the caps never
 * bind for n < 9999 and the result is order-independent - this restatement (and the HIP op)
 * covers exactly that case and rejects binding caps.  stats[4] += {fg rois, bg rois, fg weight,
 * bg weight} (the op's display counters, :90-101).
 * ------------------------------------------------------------------------- */
int oracle_roi_label(const float* S, const float* U, const float* L, const float* CW, int n, int cs,
                     int c, float fg_thresh, float bg_thresh_hi, float bg_thresh_lo, int top_k,
                     int num_pos, int num_neg, int32_t* RL, float* RW, float* stats) {
  if (num_pos < n || num_neg < n) return -1;
  const int off = cs - c;
  int* hn = (int*)malloc(sizeof(int) * (size_t)(c * top_k + 1));
  int* hc = (int*)malloc(sizeof(int) * (size_t)(c * top_k + 1));
  float* hp = (float*)malloc(sizeof(float) * (size_t)(c * top_k + 1));
  int nh = 0;
  for (int cc = 0; cc < c; ++cc) {
    if (L[cc] != 1.f) continue;
    for (int k = 0; k < top_k; ++k) {
      float max_pred = -FLT_MAX;
      int max_idx = -1;
      for (int i = 0; i < n; ++i) {
        if (max_pred < S[i * cs + cc + off]) {
          int seen = 0;
          for (int j = 0; j < nh; ++j) seen |= (hn[j] == i);
          if (!seen) {
            max_pred = S[i * cs + cc + off];
            max_idx = i;
          }
        }
      }
      hn[nh] = max_idx; hc[nh] = cc; hp[nh] = max_pred; ++nh;
    }
  }
  for (int i = 0; i < n; ++i) {
    float max_iou = -FLT_MAX;
    int max_idx = -1;
    for (int j = 0; j < nh; ++j) {
      const int g = hn[j];
      if (g >= 0 && max_iou < U[(size_t)i * n + g]) {
        max_iou = U[(size_t)i * n + g];
        max_idx = j;
      }
    }
    if (max_idx < 0) { RL[i] = 0; RW[i] = 0.f; continue; }   /* no labelled class: UB upstream */
    int assign_c = hc[max_idx];
    float assign_w = CW ? CW[assign_c] : hp[max_idx];
    if (max_iou >= fg_thresh) {
      assign_c = assign_c + 1;
      stats[0] += 1.f; stats[2] += assign_w;
    } else if (max_iou >= bg_thresh_lo && max_iou < bg_thresh_hi) {
      assign_c = 0;
      stats[1] += 1.f; stats[3] += assign_w;
    } else {
      assign_c = assign_c + 1;
      assign_w = 0.f;
    }
    RL[i] = assign_c;
    RW[i] = assign_w;
  }
  free(hn); free(hc); free(hp);
  return 0;
}

/* SoftmaxWithLossN forward.  ref: detectron/ops/softmax_with_loss_n_op.cc:152-263 (label mode).
 * X [N, D] logits, T int32 [N] labels, W [N] optional sample weights -> P [N, D] softmax,
 * loss = scale * sum_i(-w_i log P[i, T_i]) / sum_i w_i  (0 when the weights sum to 0).
 * The row softmax is Caffe2's softmax_utils::SoftmaxCPU in its logarithmic form (third-party,
 * pytorch v1.3.0 caffe2/operators/softmax_utils.cc - restated from its published algorithm: row
 * max, x - max, exp, row sum, log-softmax = x - max - log(sum); P = exp(log-softmax), :206).
 * Serial fp32 sums in index order.  Returns -1 on a label outside [0, D) (the ENFORCE at :192). */
int oracle_softmax_with_loss_n_fwd(const float* X, const int32_t* T, const float* W, int N, int D,
                                   float scale, float* P, float* loss) {
  float loss_sum = 0.f, weight_sum = 0.f;
  for (int i = 0; i < N; ++i) {
    const float* x = X + (size_t)i * D;
    float* p = P + (size_t)i * D;
    float m = x[0];
    for (int d = 1; d < D; ++d) m = x[d] > m ? x[d] : m;
    float s = 0.f;
    for (int d = 0; d < D; ++d) { p[d] = expf(x[d] - m); s += p[d]; }
    const float ls = logf(s);
    for (int d = 0; d < D; ++d) p[d] = (x[d] - m) - ls;        /* log-softmax */
    if (T[i] < 0 || T[i] >= D) return -1;
    const float w = W ? W[i] : 1.f;
    loss_sum += -p[T[i]] * w;
    weight_sum += w;
    for (int d = 0; d < D; ++d) p[d] = expf(p[d]);
  }
  *loss = weight_sum != 0.f ? loss_sum * scale / weight_sum : 0.f;
  return 0;
}

/* SoftmaxWithLossNGradient.  ref: softmax_with_loss_n_op.cc:265-357 (label mode): dX = (P -
 * onehot(T)) * w_i, then scaled by scale / total * dloss where total = the NUMBER of samples with
 * w_i > 1e-12 when weights are given (:309-311: the weight sum is commented out upstream), N
 * otherwise; left unscaled when total == 0. */
void oracle_softmax_with_loss_n_bwd(const int32_t* T, const float* W, const float* P, float dloss,
                                    int N, int D, float scale, float* dX) {
  float total = 0.f;
  for (int i = 0; i < N; ++i) {
    for (int d = 0; d < D; ++d) dX[(size_t)i * D + d] = P[(size_t)i * D + d];
    dX[(size_t)i * D + T[i]] = P[(size_t)i * D + T[i]] - 1.0f;
    if (W) {
      for (int d = 0; d < D; ++d) dX[(size_t)i * D + d] *= W[i];
      if (W[i] > 1e-12) total += 1.f;
    }
  }
  if (!W) total = (float)N;
  if (total > 0) {
    const float a = scale / total * dloss;
    for (size_t k = 0; k < (size_t)N * D; ++k) dX[k] *= a;
  }
}

/* RoIEntropy.  ref: detectron/ops/roi_entropy_op.cu:24-62, 69-112 (CUDA only upstream).
 * S [n] scores and C [n] class ids (as floats) of the detections BoxWithNMSLimit kept; per class
 * c = C - rm_bg: N_c = count, CS_c = sum of scores, and
 *   E_c = 1 + sum_i (p_i log p_i) / log N_c,  p_i = S_i / CS_c   (E_c = 1 when N_c is 0 or 1)
 * i.e. one minus the normalised entropy of the class's score distribution (:64-66; the kernel
 * adds val = +p log p because of its "-1 * -1").  E [num_classes] starts at 1 (:82-84). */
void oracle_roi_entropy(const float* S, const float* C, int n, int num_classes, int rm_bg, float* E) {
  const int off = rm_bg ? -1 : 0;
  float* N = (float*)calloc((size_t)num_classes, sizeof(float));
  float* CS = (float*)calloc((size_t)num_classes, sizeof(float));
  for (int c = 0; c < num_classes; ++c) E[c] = 1.f;
  for (int i = 0; i < n; ++i) {
    const int c = (int)C[i] + off;
    N[c] += 1.f;
    CS[c] += S[i];
  }
  for (int i = 0; i < n; ++i) {
    const int c = (int)C[i] + off;
    const float p = S[i] / CS[c];
    float val = -1.0f * -1.0f * p * logf(p);
    if (N[c] == 1) val = 0.f;
    else val = val / logf(N[c]);
    E[c] += val;
  }
  free(N); free(CS);
}

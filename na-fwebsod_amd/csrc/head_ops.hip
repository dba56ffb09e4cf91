// WSDDN two-stream outputs (dual softmax, residual noise branch, per-image
// proposal sums), their backward, the IoU-graph entropy gate, the weighted
// cross-entropy loss, Stat accumulate and the fused ACM momentum-SGD update.
//
// ref: detectron/modeling/wsl_heads.py:23-56, :213-227
//      detectron/modeling/webly_heads.py:32-74, :265-391
//      detectron/ops/cross_entropy_wsl_op.cc:7-180, cross_entropy_wsl_op.h:90-110
//      detectron/ops/roi_iou_op.cu:27-62
//      detectron/ops/acm_weightdecay_momentum_sgd_op.h:48-112, _gpu.cu:7-33
//      detectron/ops/stat_op.cu:14-20
//
// All of this is [R,C] / [1,C] sized (R ~ 2000, C = 20..80): latency-bound.
// The kernels are therefore few, each embarrassingly parallel over rows or
// over (class, image) columns, with fixed-order reductions so results are
// bitwise reproducible run to run.
#include <math.h>
#include "naws_common.h"

namespace {

constexpr int TB = 256;

// Block-wide fixed-order tree reductions (TB threads).
__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
  __syncthreads();
  if (lane == 0) red[wid] = v;
  __syncthreads();
  float t = 0.f;
#pragma unroll
  for (int i = 0; i < TB / 64; ++i) t += red[i];
  return t;
}
__device__ __forceinline__ float block_max(float v, float* red) {
  v = wave_max(v);
  const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
  __syncthreads();
  if (lane == 0) red[wid] = v;
  __syncthreads();
  float t = red[0];
#pragma unroll
  for (int i = 1; i < TB / 64; ++i) t = fmaxf(t, red[i]);
  return t;
}

__device__ __forceinline__ int find_segment(const int32_t* seg_off, int nseg, int r) {
  int s = 0;
  while (s + 1 < nseg && r >= seg_off[s + 1]) ++s;
  return s;
}

// ---- softmax over the proposals of one image, one class (column) ----------
// grid (C, nseg, nb): alpha_det[b][r][c] = exp(z - max_r z) / sum_r exp(..)
__global__ __launch_bounds__(TB) void det_softmax_kernel(
    const float* __restrict__ fc8d, const float* __restrict__ noisy_fc8d, int ld,
    const int32_t* __restrict__ seg_off, int Rt, int C, float* __restrict__ alpha_det) {
  __shared__ float red[TB / 64];
  const int c = blockIdx.x, s = blockIdx.y, b = blockIdx.z;
  const int lo = seg_off[s], hi = seg_off[s + 1];
  const bool noise = (b == 1);
  float m = -INFINITY;
  for (int r = lo + threadIdx.x; r < hi; r += TB) {
    float z = fc8d[(int64_t)r * ld + c];
    if (noise) z += noisy_fc8d[(int64_t)r * ld + c];
    m = fmaxf(m, z);
  }
  m = block_max(m, red);
  float sum = 0.f;
  for (int r = lo + threadIdx.x; r < hi; r += TB) {
    float z = fc8d[(int64_t)r * ld + c];
    if (noise) z += noisy_fc8d[(int64_t)r * ld + c];
    sum += expf(z - m);
  }
  sum = block_sum(sum, red);
  float* out = alpha_det + (int64_t)b * Rt * C;
  for (int r = lo + threadIdx.x; r < hi; r += TB) {
    float z = fc8d[(int64_t)r * ld + c];
    if (noise) z += noisy_fc8d[(int64_t)r * ld + c];
    out[(int64_t)r * C + c] = expf(z - m) / sum;
  }
}

// ---- softmax over classes per row, product with alpha_det ------------------
// one lane = one (branch, row, class) element.  Every lane of a row recomputes the row's maximum
// and sum in class order - 2 C cached loads and C expf, nothing next to a launch - so that the
// launch has Rt * nb * C lanes instead of Rt * nb (125 waves on 1024 SIMDs: 28 us of latency)
// and every element is rounded exactly as in the one-lane-per-row form it replaces.
__global__ __launch_bounds__(TB) void cls_softmax_mul_kernel(
    const float* __restrict__ fc8c, const float* __restrict__ noisy_fc8c, int ld, int Rt, int C,
    int nb, const float* __restrict__ alpha_det, float* __restrict__ alpha_cls,
    float* __restrict__ rois_pred) {
  const int64_t t = (int64_t)blockIdx.x * TB + threadIdx.x;
  if (t >= (int64_t)Rt * nb * C) return;
  const int c0 = (int)(t % C);
  const int64_t br = t / C;
  const int b = (int)(br / Rt), r = (int)(br % Rt);
  const bool noise = (b == 1);
  const float* zc = fc8c + (int64_t)r * ld;
  const float* zn = noise ? noisy_fc8c + (int64_t)r * ld : nullptr;
  float m = -INFINITY;
  for (int c = 0; c < C; ++c) {
    float z = zc[c];
    if (noise) z += zn[c];
    m = fmaxf(m, z);
  }
  float sum = 0.f;
  for (int c = 0; c < C; ++c) {
    float z = zc[c];
    if (noise) z += zn[c];
    sum += expf(z - m);
  }
  const int64_t o = ((int64_t)b * Rt + r) * C;
  float z = zc[c0];
  if (noise) z += zn[c0];
  const float a = expf(z - m) / sum;
  alpha_cls[o + c0] = a;
  rois_pred[o + c0] = a * alpha_det[o + c0];
}

// ---- cls_prob[b][s][c] = sum over the image's proposals --------------------
__global__ __launch_bounds__(TB) void seg_colsum_kernel(const float* __restrict__ X,
                                                        const int32_t* __restrict__ seg_off,
                                                        int nseg, int Rt, int C,
                                                        float* __restrict__ out) {
  __shared__ float red[TB / 64];
  const int c = blockIdx.x, s = blockIdx.y, b = blockIdx.z;
  const int lo = seg_off[s], hi = seg_off[s + 1];
  const float* x = X + (int64_t)b * Rt * C;
  float acc = 0.f;
  for (int r = lo + threadIdx.x; r < hi; r += TB) acc += x[(int64_t)r * C + c];
  acc = block_sum(acc, red);
  if (threadIdx.x == 0) out[((int64_t)b * nseg + s) * C + c] = acc;
}

// ---- backward: one lane = one (row, class) element, both branches -----------
// Row softmax:  dz = y * (dy - <dy, y>).  Column softmax: the inner product
// <dy_col, y_col> = sum_r g[c]*alpha_cls*alpha_det = g[c] * cls_prob[c].
// (Every lane of a row recomputes the row's inner product in class order: same rounding as one
// lane per row, Rt * C lanes instead of Rt - 49 us of serial strided loads on 63 waves before.)
__global__ __launch_bounds__(TB) void wsddn_bwd_kernel(
    const float* __restrict__ alpha_cls, const float* __restrict__ alpha_det,
    const float* __restrict__ cls_prob, const float* __restrict__ d_cls_prob,
    const int32_t* __restrict__ seg_off, int nseg, int Rt, int C, int nb,
    float* __restrict__ d_fc8c, float* __restrict__ d_fc8d, float* __restrict__ d_nfc8c,
    float* __restrict__ d_nfc8d, int ldd) {
  const int64_t t = (int64_t)blockIdx.x * TB + threadIdx.x;
  if (t >= (int64_t)Rt * C) return;
  const int r = (int)(t / C), c = (int)(t % C);
  const int s = find_segment(seg_off, nseg, r);
  const int64_t o = (int64_t)r * ldd + c;
  float sum_c = 0.f, sum_d = 0.f;
  for (int b = 0; b < nb; ++b) {
    const float* ac = alpha_cls + ((int64_t)b * Rt + r) * C;
    const float* ad = alpha_det + ((int64_t)b * Rt + r) * C;
    const float* g = d_cls_prob + ((int64_t)b * nseg + s) * C;
    const float* y = cls_prob + ((int64_t)b * nseg + s) * C;
    float dot = 0.f;
    for (int k = 0; k < C; ++k) dot += (g[k] * ad[k]) * ac[k];
    const float dzc = ac[c] * (g[c] * ad[c] - dot);
    const float dzd = ad[c] * (g[c] * ac[c] - g[c] * y[c]);
    if (b == 0) {
      sum_c = dzc;
      sum_d = dzd;
    } else {
      sum_c += dzc;  // fan-in of Add(fc8c, noisy_fc8c)
      sum_d += dzd;
      d_nfc8c[o] = dzc;
      d_nfc8d[o] = dzd;
    }
  }
  d_fc8c[o] = sum_c;
  d_fc8d[o] = sum_d;
}

// ---- entropy gate ----------------------------------------------------------
__device__ __forceinline__ float entropy_term(float p) {
  float e = -(p * logf(p));
  return isnan(e) ? 0.f : e;  // ReplaceNaN(value=0)
}

__device__ __forceinline__ float iou_int(int ax0, int ay0, int ax1, int ay1, int bx0, int by0,
                                         int bx1, int by1) {
  int xmin = max(ax0, bx0), ymin = max(ay0, by0);
  int xmax = min(ax1, bx1), ymax = min(ay1, by1);
  int w = (int)fmax(xmax - xmin + 1., 0.);
  int h = (int)fmax(ymax - ymin + 1., 0.);
  float inters = (float)(w * h);
  float uni = (float)((ax1 - ax0 + 1.) * (ay1 - ay0 + 1.) + (bx1 - bx0 + 1.) * (by1 - by0 + 1.) -
                      (double)inters);
  return inters / uni;
}

constexpr int GR = 64;    // rows per workgroup
constexpr int GJ = 64;    // j-tile staged in LDS
// GCC: classes accumulated per pass (registers); every pass recomputes the IoUs, so a launch takes
// the smallest of 20 / 40 / 80 that covers C in one pass (C = 80: one pass instead of four).
// The sums run over j in the same order whatever GCC is: results are bit-identical.

// grid (ceil(max_seg/GR), nseg, JCH).  Dpart[q][r][c] = sum_{j in chunk q} J(r,j) E[j,c]
template <int GCC>
__global__ __launch_bounds__(GR) void gate_D_kernel(const float* __restrict__ rois,
                                                    const float* __restrict__ rois_pred,
                                                    const int32_t* __restrict__ seg_off,
                                                    int Rt, int C, int JCH,
                                                    float* __restrict__ Dpart) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  int* jbox = reinterpret_cast<int*>(smem_raw);            // [GJ][4]
  float* jE = reinterpret_cast<float*>(smem_raw) + GJ * 4;  // [GJ][GCC]
  const int s = blockIdx.y, q = blockIdx.z;
  const int lo = seg_off[s], hi = seg_off[s + 1];
  const int len = hi - lo;
  const int per = (len + JCH - 1) / JCH;
  const int jlo = lo + q * per, jhi = min(hi, jlo + per);
  const int r = lo + blockIdx.x * GR + threadIdx.x;
  const bool active = r < hi;
  int bx0 = 0, by0 = 0, bx1 = 0, by1 = 0;
  if (active) {
    bx0 = (int)rois[r * 5 + 1]; by0 = (int)rois[r * 5 + 2];
    bx1 = (int)rois[r * 5 + 3]; by1 = (int)rois[r * 5 + 4];
  }
  for (int c0 = 0; c0 < C; c0 += GCC) {
    const int cc = min(GCC, C - c0);
    float acc[GCC];
#pragma unroll
    for (int k = 0; k < GCC; ++k) acc[k] = 0.f;
    for (int j0 = jlo; j0 < jhi; j0 += GJ) {
      const int nj = min(GJ, jhi - j0);
      __syncthreads();
      if ((int)threadIdx.x < nj) {
        const int j = j0 + threadIdx.x;
        jbox[threadIdx.x * 4 + 0] = (int)rois[j * 5 + 1];
        jbox[threadIdx.x * 4 + 1] = (int)rois[j * 5 + 2];
        jbox[threadIdx.x * 4 + 2] = (int)rois[j * 5 + 3];
        jbox[threadIdx.x * 4 + 3] = (int)rois[j * 5 + 4];
      }
      for (int i = threadIdx.x; i < nj * GCC; i += GR) {
        const int jj = i / GCC, k = i % GCC;
        jE[i] = (k < cc) ? entropy_term(rois_pred[(int64_t)(j0 + jj) * C + c0 + k]) : 0.f;
      }
      __syncthreads();
      if (active) {
#pragma unroll 4
        for (int jj = 0; jj < nj; ++jj) {
          const float w = (j0 + jj == r)
                              ? 1.0f
                              : iou_int(jbox[jj * 4], jbox[jj * 4 + 1], jbox[jj * 4 + 2],
                                        jbox[jj * 4 + 3], bx0, by0, bx1, by1);
#pragma unroll
          for (int k = 0; k < GCC; ++k) acc[k] += w * jE[jj * GCC + k];
        }
      }
    }
    if (active) {
#pragma unroll
      for (int k = 0; k < GCC; ++k)
        if (k < cc) Dpart[((int64_t)q * Rt + r) * C + c0 + k] = acc[k];
    }
  }
}

// hatE_sum, norm, clip, class weights in two deterministic levels.  Level 1, grid (row chunks of
// GF_ROWS, nseg): one lane per (row, class) element - rois_pred and the JCH partial D planes are
// read as contiguous runs - the e * e / leaky(d) terms parked in LDS as [row][class], then lane c
// adds its class over the chunk's rows in row order.  Level 2, grid (nseg): lane c adds the chunk
// sums in chunk order and finishes.  (One workgroup per (class, image) walking the rows with a
// class-strided access took 44 us for 2 x 2000 x 20 elements.)
constexpr int GF_ROWS = 16;
__global__ __launch_bounds__(TB) void gate_partial_kernel(
    const float* __restrict__ rois_pred, const float* __restrict__ Dpart,
    const int32_t* __restrict__ seg_off, int Rt, int C, int JCH, int nchunk,
    float* __restrict__ part) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* vals = reinterpret_cast<float*>(smem_raw);          // [GF_ROWS][C]
  const int ch = blockIdx.x, s = blockIdx.y;
  const int lo = seg_off[s] + ch * GF_ROWS, hi = min(seg_off[s + 1], lo + GF_ROWS);
  const int n = max(hi - lo, 0) * C;
  for (int i = threadIdx.x; i < n; i += TB) {
    const int64_t o = (int64_t)lo * C + i;                    // element (row lo + i / C, class i % C)
    const float e = entropy_term(rois_pred[o]);
    float d = 0.f;
    for (int q = 0; q < JCH; ++q) d += Dpart[(int64_t)q * Rt * C + o];
    d = d >= 0.f ? d : 0.01f * d;  // LeakyRelu(alpha=0.01)
    const float g = e / d;         // Div
    vals[i] = e * g;               // Mul
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += TB) {
    float acc = 0.f;
    for (int r = 0; r < hi - lo; ++r) acc += vals[r * C + c];   // ReduceSum, row order
    part[((int64_t)s * nchunk + ch) * C + c] = acc;
  }
}

__global__ __launch_bounds__(TB) void gate_finish_kernel(
    const float* __restrict__ part, const float* __restrict__ cls_prob,
    const float* __restrict__ labels_oh, const int32_t* __restrict__ seg_off, int C, int nchunk,
    float* __restrict__ cw, float* __restrict__ cw_noise, float* __restrict__ hatE_sum,
    float* __restrict__ hatE_norm) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* grp = reinterpret_cast<float*>(smem_raw);             // [G][C]
  const int s = blockIdx.x;
  const int lo = seg_off[s], hi = seg_off[s + 1];
  const int used = (hi - lo + GF_ROWS - 1) / GF_ROWS;          // chunks that hold rows of this image
  // G = TB / C lane groups: group g adds the chunks g, g + G, ... of its class (in that order),
  // then lane c adds the G group sums in group order - fixed association, G-fold fewer serial loads
  const int G = max(1, TB / C);
  for (int c = threadIdx.x % C, g = threadIdx.x / C; g < G && c < C; g += TB) {
    float acc = 0.f;
    for (int k = g; k < used; k += G) acc += part[((int64_t)s * nchunk + k) * C + c];
    grp[g * C + c] = acc;
  }
  if (C > TB) {               // more classes than lanes: lane c walks its classes, one group
    for (int c = threadIdx.x + TB; c < C; c += TB) {
      float acc = 0.f;
      for (int k = 0; k < used; ++k) acc += part[((int64_t)s * nchunk + k) * C + c];
      grp[c] = acc;
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += TB) {
    float acc = 0.f;
    for (int g = 0; g < G; ++g) acc += grp[g * C + c];
    const int o = s * C + c;
    const float y = cls_prob[o];
    const float n = (float)(hi - lo);
    const float norm = (logf(n) - logf(y)) * y;
    float v = acc / norm;
    v = (v < 0.f) ? 0.f : v;  // Clip(min=0,max=1); NaN passes through
    v = (v > 1.f) ? 1.f : v;
    const float bg = 1.0f - labels_oh[o];
    const float wn = v * bg;
    hatE_sum[o] = acc;
    hatE_norm[o] = v;
    cw_noise[o] = wn;
    cw[o] = 1.0f - wn;
  }
}

// ---- (weighted) cross entropy: one lane per problem, serial in index order -
// (lab_period: problem p reads the labels of problem p % lab_period - the two branches of the
// noise-aware head share one labels_oh per image, so no [2, nseg, C] copy of it is ever made)
__global__ void wce_fwd_kernel(const float* __restrict__ X, const float* __restrict__ L,
                               const float* __restrict__ W, int N, int C, int is_mean,
                               float* __restrict__ Y, int lab_period) {
  if (threadIdx.x != 0) return;
  const int n = N * C;
  X += (int64_t)blockIdx.x * n; L += (int64_t)(blockIdx.x % lab_period) * n;
  if (W) W += (int64_t)blockIdx.x * n;
  Y += blockIdx.x;
  const float norm = is_mean ? (float)C : 1.0f;
  float loss = 0.f;
  for (int i = 0; i < n; ++i) {
    const float prob = fmaxf(X[i], 1e-20f);
    const float one_prob = fmaxf(1.0f - X[i], 1e-20f);
    float t = L[i] * logf(prob) + (1.0f - L[i]) * logf(one_prob);
    if (W) t *= W[i];
    loss -= t;
  }
  float y = loss / norm;
  Y[0] = y * (float)(1.0 / N);
}

__global__ void wce_bwd_kernel(const float* __restrict__ X, const float* __restrict__ L,
                               const float* __restrict__ W, const float* __restrict__ dY, int N,
                               int C, int is_mean, float* __restrict__ dX, int lab_period,
                               float dy_const) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= N * C) return;
  const int64_t i = (int64_t)blockIdx.y * N * C + e;
  const float lab = L[(int64_t)(blockIdx.y % lab_period) * N * C + e];
  const float norm = is_mean ? (float)C : 1.0f;
  const float grad = dY ? dY[blockIdx.y] : dy_const;     // (dY NULL: the loss seed, one constant)
  const float prob = fmaxf(X[i], 1e-20f);
  const float one_prob = fmaxf(1.0f - X[i], 1e-20f);
  float v = fminf(grad * (-1.0f * lab / prob - (-1.0f) * (1.0f - lab) / one_prob) / norm, 1e4f);
  if (W) v *= W[i];
  dX[i] = v * (float)(1.0 / N);
}

// ---- Stat -------------------------------------------------------------------
__global__ void stat_kernel(const float* __restrict__ I, const float* __restrict__ L, int n,
                            int init, float* __restrict__ AI, float* __restrict__ AL) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float ai = init ? 0.f : AI[i];
  const float al = init ? 0.f : AL[i];
  AI[i] = I[i] * L[i] + ai;
  AL[i] = L[i] + al;
}

// ---- fused ACM weight-decay momentum SGD over a parameter arena ------------
// (sgd_elem: naws_common.h - shared with the wgrad GEMM that applies the update in its epilogue)
__device__ __forceinline__ void sgd_update4(const float4& g, float4& m, float4& p, float scale,
                                            float wd, float LR, float momentum, int nesterov) {
  sgd_elem(g.x, m.x, p.x, scale, wd, LR, momentum, nesterov);
  sgd_elem(g.y, m.y, p.y, scale, wd, LR, momentum, nesterov);
  sgd_elem(g.z, m.z, p.z, scale, wd, LR, momentum, nesterov);
  sgd_elem(g.w, m.w, p.w, scale, wd, LR, momentum, nesterov);
}

struct SgdRowmaxRegions {      // rows whose max|updated parameter| is reported (fp16x2 re-split)
  long long start[4], end[4], rowlen[4], out[4];
  int n;
};

__global__ __launch_bounds__(TB) void acm_sgd_kernel(
    const float4* __restrict__ grad, float4* __restrict__ mom, const float* __restrict__ lr,
    float4* __restrict__ param, float4* __restrict__ acm, int64_t total4,
    const int64_t* __restrict__ seg_end, const float* __restrict__ seg_lr_mult,
    const float* __restrict__ seg_wd, int nseg, float momentum, int nesterov, float scale,
    int do_update, int first, unsigned* __restrict__ rowmax, SgdRowmaxRegions rm) {
  const float base_lr = lr[0];
  for (int64_t i = (int64_t)blockIdx.x * TB + threadIdx.x; i < total4;
       i += (int64_t)gridDim.x * TB) {
    const int64_t e = i * 4;
    int lo = 0, hi = nseg - 1;  // first segment with seg_end > e
    while (lo < hi) {
      int mid = (lo + hi) >> 1;
      if (seg_end[mid] > e) hi = mid; else lo = mid + 1;
    }
    const float LR = base_lr * seg_lr_mult[lo];
    const float wd = seg_wd[lo];
    float4 g = grad[i];
    float4 a = (first || acm == nullptr) ? make_float4(0.f, 0.f, 0.f, 0.f) : acm[i];
    a.x += g.x; a.y += g.y; a.z += g.z; a.w += g.w;
    if (!do_update) { acm[i] = a; if (first) mom[i] = make_float4(0.f, 0.f, 0.f, 0.f); continue; }
    float4 p = param[i];
    float4 m = first ? make_float4(0.f, 0.f, 0.f, 0.f) : mom[i];
    float pv[4] = {p.x, p.y, p.z, p.w};
    sgd_update4(a, m, p, scale, wd, LR, momentum, nesterov);
    float mv[4] = {m.x, m.y, m.z, m.w};
    pv[0] = p.x; pv[1] = p.y; pv[2] = p.z; pv[3] = p.w;
    // the update value goes to the acmgrad OUTPUT, which is then cleared; the
    // grad blob itself is not written (acm_weightdecay_momentum_sgd_op.h:94-108)
    mom[i] = make_float4(mv[0], mv[1], mv[2], mv[3]);
    param[i] = make_float4(pv[0], pv[1], pv[2], pv[3]);
    if (acm) acm[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (rowmax) {
      // max|updated parameter| per matrix row, for the fp16x2 re-split of the weights that
      // follows (saves its 0.96 GB maxima pass).  A wave's 64 float4s are 256 consecutive floats
      // and the regions' starts / row lengths are multiples of 256 (checked on the host): the
      // wave lies in one row.
      // (region table in SGPRs; the atomic has no return value and nothing waits for it)
      const int64_t i0 = i - (threadIdx.x & 63);          // wave-uniform: keep it on the scalar unit
      const int64_t e0 = (((int64_t)__builtin_amdgcn_readfirstlane((int)(i0 >> 32)) << 32) |
                          (unsigned)__builtin_amdgcn_readfirstlane((int)i0)) * 4;
      float m = fmaxf(fmaxf(fabsf(pv[0]), fabsf(pv[1])), fmaxf(fabsf(pv[2]), fabsf(pv[3])));
#pragma unroll
      for (int d = 32; d > 0; d >>= 1) m = fmaxf(m, __shfl_xor(m, d));
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (r < rm.n && e0 >= rm.start[r] && e0 < rm.end[r]) {
          // 256-float chunk index / chunks per row (both < 2^24)
          const unsigned row = (unsigned)((e0 - rm.start[r]) >> 8) / (unsigned)(rm.rowlen[r] >> 8);
          if ((threadIdx.x & 63) == 0 && m > 0.f)
            atomicMax(rowmax + rm.out[r] + row, __float_as_uint(m));
        }
      }
    }
  }
}


// ---- the same update, writing the fp16x2 operand planes of the weight matrices itself -----------
// (VERDICT r2 #5.)  fc6_w / fc7_w are consumed by the head GEMMs as row-scaled f16 hi / lo planes
// P[2][batch][cols/16][rows][16] (naws_split_f16x2).  Re-splitting them after every update is a
// 0.96 GB read + 0.96 GB write on the update stream, beside the HBM-bound first conv layers of the
// next iteration.  A row's scale only needs an UPPER BOUND of its max|w|, and one SGD
// step moves a weight by a tiny fraction of itself: with bound = 2 x (the row maximum BEFORE the
// update, reported by the previous call) this kernel converts each updated value on the way out.
// Should any |w_new| exceed its bound, *overflow = overflow_tag is raised and the caller's
// conditional re-split (naws_split_f16x2_rows_if, queued behind this launch) redoes the planes
// from the exact maxima this kernel reports - the device decides, the host never looks.
//
// Geometry: a workgroup owns a 32-row x 256-column tile.  Pass 1: wave w, step j holds row
// 4 j + w: its 64 lanes read three 1 KB runs (grad, momentum, param: float4 per lane), update,
// write two back, fold max|w_new| over the wave, and park the scaled hi / lo halves in LDS as
// [slab][row][16 f16] (slab stride 1056 B: the four slabs a 16-lane ds_write_b64 group touches
// land on disjoint 8-bank windows).  Pass 2: the LDS image leaves as 16 x 1 KB runs per plane,
// each one (slab, 32 rows) block of the K-slab-major plane.
struct SgdPlaneRegion {
  long long start;                 // first arena element
  int rows, cols, rows_per_batch;  // rows x cols row-major; planes [2][rows / rows_per_batch][cols/16][rows_per_batch][16]
  int tile0;                       // first workgroup of this region
  unsigned short* planes;
  long long plane_stride;          // elements between the hi and the lo plane
  const unsigned* bound;           // [rows] max|w| per row before the update (bit patterns)
  unsigned* rowmax;                // [rows] max|w| per row after it (atomic max; caller zeroes)
  float* inv_scale;                // [rows]
  unsigned* colmax;                // [rows / rows_per_batch][cols] max|w| per column after the update (nullable)
};
struct SgdPlaneArgs {
  SgdPlaneRegion r[4];
  int n, tiles;                    // regions, tile workgroups
  long long lin_start[4], lin_end[4];   // the arena outside the regions, in float4 units
  int lin_block0[5];               // first linear workgroup of each range (prefix), relative to `tiles`
  int n_lin;
};
constexpr int SGP_SLAB = 1056;     // bytes per slab in the LDS image (32 rows x 32 B + 32)
constexpr int SGP_PLANE = 16 * SGP_SLAB;

__device__ __forceinline__ int sgd_segment(const int64_t* __restrict__ seg_end, int nseg, int64_t e) {
  int lo = 0, hi = nseg - 1;    // first segment with seg_end > e
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (seg_end[mid] > e) hi = mid; else lo = mid + 1;
  }
  return lo;
}

// FMT: 0 = fp16x2 (row-scaled f16 hi / lo; bounds, maxima and the overflow word as described),
// 1 = the exact 3 x bf16 split of the fp32x3 plan, 2 = the bf16 plan's single rounded plane
// (neither has a scale: no bound, no maxima, nothing can overflow).
template <int FMT>
__device__ __forceinline__ void acm_sgd_planes_block(
    const int bid, unsigned char* __restrict__ img, float* __restrict__ cmred,
    const float4* __restrict__ grad, float4* __restrict__ mom, const float base_lr,
    float4* __restrict__ param, const int64_t* __restrict__ seg_end,
    const float* __restrict__ seg_lr_mult, const float* __restrict__ seg_wd, int nseg,
    float momentum, int nesterov, float scale, int first, const SgdPlaneArgs& a,
    int* __restrict__ overflow, int overflow_tag) {
  constexpr int NPL = FMT == 0 ? 2 : (FMT == 1 ? 3 : 1);
  if (bid >= a.tiles) {
    // ---- everything outside the matrix regions (biases, fc8): the plain float4 update
    const int lb = bid - a.tiles;
    int q = 0;
#pragma unroll
    for (int k = 1; k < 4; ++k) q += (k < a.n_lin && lb >= a.lin_block0[k]) ? 1 : 0;
    const int nb = a.lin_block0[q + 1] - a.lin_block0[q];
    for (int64_t i = a.lin_start[q] + (int64_t)(lb - a.lin_block0[q]) * TB + threadIdx.x;
         i < a.lin_end[q]; i += (int64_t)nb * TB) {
      const int sg = sgd_segment(seg_end, nseg, i * 4);
      float4 g = grad[i], p = param[i];
      float4 m = first ? make_float4(0.f, 0.f, 0.f, 0.f) : mom[i];
      sgd_update4(g, m, p, scale, seg_wd[sg], base_lr * seg_lr_mult[sg], momentum, nesterov);
      mom[i] = m;
      param[i] = p;
    }
    return;
  }
  int ri = 0;
#pragma unroll
  for (int k = 1; k < 4; ++k) ri += (k < a.n && bid >= a.r[k].tile0) ? 1 : 0;
  const SgdPlaneRegion& R = a.r[ri];
  const int ct = R.cols >> 8;                         // column tiles per row band
  const int t = bid - R.tile0;
  const int band = t / ct, ctile = t - band * ct;     // column tile fastest: neighbours stream on
  const int r0 = band * 32, c0 = ctile * 256;
  const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
  // ONE hyper-parameter run per region: the CALLER guarantees that a region lies inside one run of
  // seg_end (engine._one_hyper_run disables this kernel otherwise; seg_end is device memory, so the
  // C entry point cannot look)
  const int sg = sgd_segment(seg_end, nseg, R.start);
  const float LR = base_lr * seg_lr_mult[sg], wd = seg_wd[sg];
  const int64_t base4 = (R.start + (int64_t)r0 * R.cols + c0) >> 2;
  const int row4 = R.cols >> 2;
  // the 32 rows' bounds in lanes 0..31, fetched once (a per-row load + wait inside the loop
  // serialises the wave on a memory round trip per row)
  unsigned bound_lane = 0;
  if constexpr (FMT == 0) bound_lane = R.bound[r0 + (lane & 31)];
  float rmax[8];
  float cmx[4] = {0.f, 0.f, 0.f, 0.f};     // max|w_new| of this lane's four columns over its eight rows
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    float4 g[4], m[4], p[4];
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {                  // 12 x 1 KB wave-loads in flight
      const int row = (half * 4 + jj) * 4 + wid;
      const int64_t i = base4 + (int64_t)row * row4 + lane;
      g[jj] = grad[i];
      p[jj] = param[i];
      m[jj] = first ? make_float4(0.f, 0.f, 0.f, 0.f) : mom[i];
    }
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      const int row = (half * 4 + jj) * 4 + wid;
      const int64_t i = base4 + (int64_t)row * row4 + lane;
      sgd_update4(g[jj], m[jj], p[jj], scale, wd, LR, momentum, nesterov);
      mom[i] = m[jj];
      param[i] = p[jj];
      const float pv[4] = {p[jj].x, p[jj].y, p[jj].z, p[jj].w};
      const int off = (lane >> 2) * SGP_SLAB + row * 32 + (lane & 3) * 8;
      float mx = 0.f;
      if constexpr (FMT == 0) {
        // the row's scale: from twice the maximum it had before this update (wave-uniform).
        // 2 x bound = exponent + 1 (a zero / denormal / huge bound is left as it is: the overflow
        // test below then sends the row to the exact re-split)
        const unsigned bb = (unsigned)__builtin_amdgcn_readlane((int)bound_lane, row);
        const unsigned b2 = ((bb >> 23) >= 1u && (bb >> 23) < 0xfeu) ? bb + (1u << 23) : bb;
        float sc, isc;
        naws_f16x2_scales(b2, sc, isc);
        unsigned short hq[4], lq[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          mx = fmaxf(mx, fabsf(pv[k]));
          cmx[k] = fmaxf(cmx[k], fabsf(pv[k]));
          const float v = pv[k] * sc;
          const _Float16 hi = (_Float16)v;
          float rr = v - (float)hi;
          if (!(fabsf(v) <= 65504.f)) rr = 0.f;         // NaN / overflow live in the hi plane only
          const _Float16 lo = (_Float16)rr;
          hq[k] = *reinterpret_cast<const unsigned short*>(&hi);
          lq[k] = *reinterpret_cast<const unsigned short*>(&lo);
        }
        *reinterpret_cast<uint2*>(img + off) =
            make_uint2(hq[0] | ((unsigned)hq[1] << 16), hq[2] | ((unsigned)hq[3] << 16));
        *reinterpret_cast<uint2*>(img + SGP_PLANE + off) =
            make_uint2(lq[0] | ((unsigned)lq[1] << 16), lq[2] | ((unsigned)lq[3] << 16));
      } else {
        // bf16 planes: a = a1 + a2 + a3 exactly (the split3 of gemm_x3.hip), or a1 alone
        unsigned short q[3][4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const __bf16 h1 = (__bf16)pv[k];
          float rr = pv[k] - (float)h1;
          if (!(fabsf(pv[k]) <= 3.4028234e38f)) rr = 0.f;   // inf / NaN live in plane 1 only
          const __bf16 h2 = (__bf16)rr;
          const __bf16 h3 = (__bf16)(rr - (float)h2);
          q[0][k] = *reinterpret_cast<const unsigned short*>(&h1);
          q[1][k] = *reinterpret_cast<const unsigned short*>(&h2);
          q[2][k] = *reinterpret_cast<const unsigned short*>(&h3);
        }
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl)
          *reinterpret_cast<uint2*>(img + pl * SGP_PLANE + off) = make_uint2(
              q[pl][0] | ((unsigned)q[pl][1] << 16), q[pl][2] | ((unsigned)q[pl][3] << 16));
      }
      // (a NaN weight must reach the overflow test: fmaxf drops NaNs, so it is carried as +inf)
      rmax[half * 4 + jj] = (pv[0] != pv[0] || pv[1] != pv[1] || pv[2] != pv[2] || pv[3] != pv[3])
                                ? __uint_as_float(0x7f800000u) : mx;
    }
  }
  if constexpr (FMT == 0)
  // Fold the wave's eight per-lane row maxima across the 64 lanes by a halving butterfly - each
  // exchange keeps half of the rows on each side, 4 + 2 + 1 + 3 shuffles instead of 8 x 6 (the
  // first version, one full wave reduction per row, was issue-bound on its ds_bpermute chains:
  // 72 % SQ_WAIT_INST_ANY at 4.6 TB/s).  Afterwards lane l holds row (l >> 3) & 7, complete.
  {
    float v4[4], v2[2], v1;
    const bool up32 = (lane & 32) != 0, up16 = (lane & 16) != 0, up8 = (lane & 8) != 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {            // xor 32: lanes < 32 keep rows 0..3, the others 4..7
      const float mine = up32 ? rmax[4 + i] : rmax[i], give = up32 ? rmax[i] : rmax[4 + i];
      v4[i] = fmaxf(mine, __shfl_xor(give, 32));
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {            // xor 16: keep rows {0,1} / {2,3} of the half
      const float mine = up16 ? v4[2 + i] : v4[i], give = up16 ? v4[i] : v4[2 + i];
      v2[i] = fmaxf(mine, __shfl_xor(give, 16));
    }
    {
      const float mine = up8 ? v2[1] : v2[0], give = up8 ? v2[0] : v2[1];
      v1 = fmaxf(mine, __shfl_xor(give, 8));
    }
    v1 = fmaxf(v1, __shfl_xor(v1, 4));
    v1 = fmaxf(v1, __shfl_xor(v1, 2));
    v1 = fmaxf(v1, __shfl_xor(v1, 1));
    // lane l now holds step j = 4 (l >> 5) + 2 ((l >> 4) & 1) + ((l >> 3) & 1) = row 4 j + wid
    const int j = ((lane >> 5) << 2) | (((lane >> 4) & 1) << 1) | ((lane >> 3) & 1);
    const int row = j * 4 + wid;
    const unsigned bb = (unsigned)__shfl((int)bound_lane, row);     // (all lanes take part)
    if ((lane & 7) == 0) {
      const unsigned b2 = ((bb >> 23) >= 1u && (bb >> 23) < 0xfeu) ? bb + (1u << 23) : bb;
      if (ctile == 0) {
        float sc, isc;
        naws_f16x2_scales(b2, sc, isc);
        R.inv_scale[r0 + row] = isc;
      }
      const bool is_inf = __float_as_uint(v1) == 0x7f800000u;       // a NaN / inf weight in the row
      if (v1 > 0.f && !is_inf) naws_atomic_max_bits(R.rowmax + r0 + row, v1);
      if (!(v1 <= __uint_as_float(b2)) || is_inf) atomicMax(overflow, overflow_tag);
    }
  }
  if constexpr (FMT == 0) {
    if (R.colmax) {
#pragma unroll
      for (int k = 0; k < 4; ++k) cmred[wid * 256 + lane * 4 + k] = cmx[k];
    }
  }
  __syncthreads();
  const int batch = r0 / R.rows_per_batch, rb0 = r0 - batch * R.rows_per_batch;
  if constexpr (FMT == 0) {
    // max|w| per column after the update (the scale of the TRANSPOSED operand planes, fc7_w^T for
    // the dgrad: naws_split_f16x2_dual then needs no maxima pass): one guarded atomic per column
    // and tile, all 256 in parallel.  NaNs are skipped as fmaxf skips them in the stand-alone pass.
    if (R.colmax) {
      const int t = threadIdx.x;
      const float v = fmaxf(fmaxf(cmred[t], cmred[256 + t]), fmaxf(cmred[512 + t], cmred[768 + t]));
      if (v > 0.f) naws_atomic_max_bits(R.colmax + (long long)batch * R.cols + c0 + t, v);
    }
  }
  // ---- pass 2: the image leaves as (slab, 32 rows) blocks: 1 KB contiguous per wave-store
  const long long sp = (long long)R.cols * R.rows_per_batch;       // one batch item of one plane
#pragma unroll
  for (int pl = 0; pl < NPL; ++pl) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int idx = k * TB + threadIdx.x;            // 16-byte chunk of the plane's image
      const int slab = idx >> 6, rem = idx & 63;       // rem = row * 2 + half
      const uint4 v = *reinterpret_cast<const uint4*>(img + pl * SGP_PLANE + slab * SGP_SLAB + rem * 16);
      const long long dst = pl * R.plane_stride + batch * sp +
                            ((long long)((c0 >> 4) + slab) * R.rows_per_batch + rb0) * 16 + rem * 8;
      *reinterpret_cast<uint4*>(R.planes + dst) = v;
    }
  }
}

// One workgroup per tile (the launch in use) ...
template <int FMT>
__global__ __launch_bounds__(TB) void acm_sgd_planes_kernel(
    const float4* __restrict__ grad, float4* __restrict__ mom, const float* __restrict__ lr,
    float4* __restrict__ param, const int64_t* __restrict__ seg_end,
    const float* __restrict__ seg_lr_mult, const float* __restrict__ seg_wd, int nseg,
    float momentum, int nesterov, float scale, int first, SgdPlaneArgs a, int* __restrict__ overflow,
    int overflow_tag) {
  constexpr int NPL = FMT == 0 ? 2 : (FMT == 1 ? 3 : 1);
  __shared__ __attribute__((aligned(16))) unsigned char img[NPL * SGP_PLANE];
  __shared__ float cmred[FMT == 0 ? 4 * 256 : 1];     // column maxima of the four waves (colmax)
  acm_sgd_planes_block<FMT>(blockIdx.x, img, cmred, grad, mom, lr[0], param, seg_end, seg_lr_mult,
                            seg_wd, nseg, momentum, nesterov, scale, first, a, overflow, overflow_tag);
}

// ... or a fixed number of resident workgroups walking the tiles (knob "sgd_wgs"): the update then
// occupies a bounded share of every CU's wave slots and LDS, and the conv body queued beside it
// keeps its own (a tile per workgroup fills the CUs with 4 x 38 KB of LDS images, which the conv
// kernels' 43 KB workgroups cannot share).  A separate kernel: the loop costs the one-tile form
// 34 VGPRs (96 -> 130, five -> three waves per SIMD, 0.71 -> 0.92 ms on the bench arena).
template <int FMT>
__global__ __launch_bounds__(TB) void acm_sgd_planes_walk_kernel(
    const float4* __restrict__ grad, float4* __restrict__ mom, const float* __restrict__ lr,
    float4* __restrict__ param, const int64_t* __restrict__ seg_end,
    const float* __restrict__ seg_lr_mult, const float* __restrict__ seg_wd, int nseg,
    float momentum, int nesterov, float scale, int first, SgdPlaneArgs a, int* __restrict__ overflow,
    int overflow_tag, int blocks) {
  constexpr int NPL = FMT == 0 ? 2 : (FMT == 1 ? 3 : 1);
  __shared__ __attribute__((aligned(16))) unsigned char img[NPL * SGP_PLANE];
  __shared__ float cmred[FMT == 0 ? 4 * 256 : 1];
  const float base_lr = lr[0];
  for (int bid = blockIdx.x; bid < blocks; bid += gridDim.x) {
    acm_sgd_planes_block<FMT>(bid, img, cmred, grad, mom, base_lr, param, seg_end, seg_lr_mult, seg_wd,
                              nseg, momentum, nesterov, scale, first, a, overflow, overflow_tag);
    __syncthreads();                                  // the LDS image is rewritten by the next tile
  }
}


// ---- MinEntropyLoss ----------------------------------------------------------
// replaces: detectron/ops/min_entropy_loss_op.cc:7-98.  One workgroup; fp64 partial sums per lane
// (the reference adds ~N*C fp32 terms serially: its own rounding is ~1e-6 relative, the fp64
// tree is closer to the exact sum than that).
__global__ __launch_bounds__(256) void min_entropy_fwd_kernel(const float* __restrict__ X,
                                                              const float* __restrict__ Lb, int N,
                                                              int C, float* __restrict__ Y) {
  __shared__ double sh[256];
  __shared__ int shn[256];
  double acc = 0.0;
  int cnt = 0;
  for (long long i = threadIdx.x; i < (long long)N * C; i += 256) {
    const int c = (int)(i % C);
    if (Lb[c] < 0.5f) continue;
    const float p = fmaxf(X[i], 1e-20f);
    acc -= (double)(p * logf(p));
    cnt += 1;
  }
  sh[threadIdx.x] = acc; shn[threadIdx.x] = cnt;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) { sh[threadIdx.x] += sh[threadIdx.x + s]; shn[threadIdx.x] += shn[threadIdx.x + s]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) Y[0] = (float)(sh[0] / (double)shn[0]);
}

__global__ __launch_bounds__(256) void min_entropy_bwd_kernel(const float* __restrict__ X,
                                                              const float* __restrict__ Lb,
                                                              const float* __restrict__ dY, int N,
                                                              int C, float* __restrict__ dX) {
  int pos = 0;
  for (int c = 0; c < C; ++c) pos += (Lb[c] < 0.5f) ? 0 : 1;
  const float scale = dY[0] / (float)((long long)pos * N);
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < (long long)N * C;
       i += (long long)gridDim.x * 256) {
    const int c = (int)(i % C);
    float g = 0.f;
    if (!(Lb[c] < 0.5f)) {
      const float p = fmaxf(X[i], 1e-20f);
      g = fminf(scale * (-1.f + (-1.f) * logf(p)), 1e4f);
    }
    dX[i] = g;
  }
}

}  // namespace

extern "C" int naws_wsddn_outputs_fwd(const float* fc8c, const float* fc8d,
                                      const float* noisy_fc8c, const float* noisy_fc8d, int ld,
                                      const int32_t* seg_off, int nseg, int Rt, int C,
                                      float* alpha_cls, float* alpha_det, float* rois_pred,
                                      float* cls_prob, void* stream) {
  if (Rt <= 0 || C <= 0 || nseg <= 0 || ld < C) return NAWS_ERR_SHAPE;
  if ((noisy_fc8c == nullptr) != (noisy_fc8d == nullptr)) return NAWS_ERR_ARG;
  NAWS_REQUIRE_PTR(fc8c); NAWS_REQUIRE_PTR(fc8d); NAWS_REQUIRE_PTR(seg_off);
  NAWS_REQUIRE_PTR(alpha_cls); NAWS_REQUIRE_PTR(alpha_det); NAWS_REQUIRE_PTR(rois_pred);
  NAWS_REQUIRE_PTR(cls_prob);
  if (nseg > 65535 || C > 65535) return NAWS_ERR_UNSUPPORTED;
  const int nb = noisy_fc8c ? 2 : 1;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(det_softmax_kernel, dim3(C, nseg, nb), dim3(TB), 0, s, fc8d, noisy_fc8d, ld,
                     seg_off, Rt, C, alpha_det);
  const int64_t elems = (int64_t)Rt * nb * C;
  if (naws_cdiv(elems, TB) > 0x7fffffffLL) return NAWS_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(cls_softmax_mul_kernel, dim3((unsigned)naws_cdiv(elems, TB)), dim3(TB), 0, s,
                     fc8c, noisy_fc8c, ld, Rt, C, nb, alpha_det, alpha_cls, rois_pred);
  hipLaunchKernelGGL(seg_colsum_kernel, dim3(C, nseg, nb), dim3(TB), 0, s, rois_pred, seg_off,
                     nseg, Rt, C, cls_prob);
  return naws_check_launch();
}

extern "C" int naws_wsddn_outputs_bwd(const float* alpha_cls, const float* alpha_det,
                                      const float* rois_pred, const float* cls_prob,
                                      const float* d_cls_prob, const int32_t* seg_off, int nseg,
                                      int Rt, int C, int nb, float* d_fc8c, float* d_fc8d,
                                      float* d_noisy_fc8c, float* d_noisy_fc8d, int ldd,
                                      void* stream) {
  (void)rois_pred;
  if (Rt <= 0 || C <= 0 || nseg <= 0 || ldd < C) return NAWS_ERR_SHAPE;
  if (nb != 1 && nb != 2) return NAWS_ERR_ARG;
  NAWS_REQUIRE_PTR(alpha_cls); NAWS_REQUIRE_PTR(alpha_det); NAWS_REQUIRE_PTR(cls_prob);
  NAWS_REQUIRE_PTR(d_cls_prob); NAWS_REQUIRE_PTR(seg_off);
  NAWS_REQUIRE_PTR(d_fc8c); NAWS_REQUIRE_PTR(d_fc8d);
  if (nb == 2) { NAWS_REQUIRE_PTR(d_noisy_fc8c); NAWS_REQUIRE_PTR(d_noisy_fc8d); }
  if (naws_cdiv((int64_t)Rt * C, TB) > 0x7fffffffLL) return NAWS_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(wsddn_bwd_kernel, dim3((unsigned)naws_cdiv((int64_t)Rt * C, TB)), dim3(TB), 0,
                     (hipStream_t)stream, alpha_cls, alpha_det, cls_prob, d_cls_prob, seg_off, nseg,
                     Rt, C, nb, d_fc8c, d_fc8d, d_noisy_fc8c, d_noisy_fc8d, ldd);
  return naws_check_launch();
}

// j-chunks per row block: the pair loop is a serial chain per wave (~80 instructions per j), so
// the launch is sized for two waves per SIMD at 2 x 2000 proposals (64 rows x 64 j per wave: a
// lone wave cannot cover its own LDS / IoU latencies - 47 us at one wave per SIMD, 34 at two; the
// finish then adds 32 partial planes instead of 16: +8 us)
static int gate_chunks(int max_seg_len) {
  int q = (max_seg_len + 63) / 64;
  return q < 1 ? 1 : (q > 32 ? 32 : q);
}

extern "C" int64_t naws_entropy_gate_workspace_floats(int Rt, int C, int nseg, int max_seg_len) {
  if (Rt <= 0 || C <= 0 || max_seg_len <= 0 || nseg <= 0) return 0;
  // the JCH partial D planes + the per-chunk class sums of the two-level finish
  return (int64_t)gate_chunks(max_seg_len) * Rt * C +
         (int64_t)nseg * naws_cdiv(max_seg_len, GF_ROWS) * C;
}

extern "C" int naws_entropy_gate_fwd(const float* rois, const float* rois_pred,
                                     const float* cls_prob, const float* labels_oh,
                                     const int32_t* seg_off, int nseg, int Rt, int C,
                                     int max_seg_len, float* workspace, float* class_weight,
                                     float* class_weight_noise, float* hatE_sum,
                                     float* hatE_sum_norm, void* stream) {
  if (Rt <= 0 || C <= 0 || nseg <= 0 || max_seg_len <= 0 || max_seg_len > Rt)
    return NAWS_ERR_SHAPE;
  NAWS_REQUIRE_PTR(rois); NAWS_REQUIRE_PTR(rois_pred); NAWS_REQUIRE_PTR(cls_prob);
  NAWS_REQUIRE_PTR(labels_oh); NAWS_REQUIRE_PTR(seg_off); NAWS_REQUIRE_PTR(workspace);
  NAWS_REQUIRE_PTR(class_weight); NAWS_REQUIRE_PTR(class_weight_noise);
  NAWS_REQUIRE_PTR(hatE_sum); NAWS_REQUIRE_PTR(hatE_sum_norm);
  if (nseg > 65535 || C > 65535) return NAWS_ERR_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  const int JCH = gate_chunks(max_seg_len);
  const dim3 dgrid((unsigned)naws_cdiv(max_seg_len, GR), nseg, JCH);
#define NAWS_GATE_D(GCCV)                                                                          \
  hipLaunchKernelGGL(gate_D_kernel<GCCV>, dgrid, dim3(GR),                                         \
                     GJ * 4 * sizeof(int) + GJ * GCCV * sizeof(float), s, rois, rois_pred, seg_off, \
                     Rt, C, JCH, workspace)
  if (C <= 20) NAWS_GATE_D(20);
  else if (C <= 40) NAWS_GATE_D(40);
  else NAWS_GATE_D(80);
#undef NAWS_GATE_D
  const int nchunk = (int)naws_cdiv(max_seg_len, GF_ROWS);
  if ((size_t)GF_ROWS * C * sizeof(float) > 64 * 1024 || nchunk > 65535) return NAWS_ERR_UNSUPPORTED;
  float* part = workspace + (int64_t)JCH * Rt * C;
  hipLaunchKernelGGL(gate_partial_kernel, dim3((unsigned)nchunk, nseg), dim3(TB),
                     (size_t)GF_ROWS * C * sizeof(float), s, rois_pred, (const float*)workspace,
                     seg_off, Rt, C, JCH, nchunk, part);
  const int G = std::max(1, TB / C);
  hipLaunchKernelGGL(gate_finish_kernel, dim3(nseg), dim3(TB), (size_t)G * C * sizeof(float), s,
                     (const float*)part, cls_prob,
                     labels_oh, seg_off, C, nchunk, class_weight, class_weight_noise, hatE_sum,
                     hatE_sum_norm);
  return naws_check_launch();
}

extern "C" int naws_weighted_ce_fwd(const float* X, const float* L, const float* W, int N, int C,
                                    int is_mean, int nprob, float* Y, void* stream) {
  if (N <= 0 || C <= 0 || nprob <= 0 || nprob > 65535) return NAWS_ERR_SHAPE;
  NAWS_REQUIRE_PTR(X); NAWS_REQUIRE_PTR(L); NAWS_REQUIRE_PTR(Y);
  hipLaunchKernelGGL(wce_fwd_kernel, dim3(nprob), dim3(64), 0, (hipStream_t)stream, X, L, W, N, C,
                     is_mean, Y, nprob);
  return naws_check_launch();
}

// nprob problems whose labels repeat with period lab_period (L: [lab_period, N, C]): problem p is
// scored against L[p % lab_period].  The loss tail of the noise-aware head: 2 branches x nseg
// images, one labels_oh per image.
extern "C" int naws_weighted_ce_shared_fwd(const float* X, const float* L, const float* W, int N, int C,
                                           int is_mean, int nprob, int lab_period, float* Y,
                                           void* stream) {
  if (N <= 0 || C <= 0 || nprob <= 0 || nprob > 65535 || lab_period <= 0 || nprob % lab_period != 0)
    return NAWS_ERR_SHAPE;
  NAWS_REQUIRE_PTR(X); NAWS_REQUIRE_PTR(L); NAWS_REQUIRE_PTR(Y);
  hipLaunchKernelGGL(wce_fwd_kernel, dim3(nprob), dim3(64), 0, (hipStream_t)stream, X, L, W, N, C,
                     is_mean, Y, lab_period);
  return naws_check_launch();
}

// ... and its gradient; dY NULL: every problem's upstream gradient is dy_const (the loss seed 1.0
// of blob.py:167-173 without a tensor of ones).
extern "C" int naws_weighted_ce_shared_bwd(const float* X, const float* L, const float* W,
                                           const float* dY, float dy_const, int N, int C, int is_mean,
                                           int nprob, int lab_period, float* dX, void* stream) {
  if (N <= 0 || C <= 0 || nprob <= 0 || nprob > 65535 || lab_period <= 0 || nprob % lab_period != 0)
    return NAWS_ERR_SHAPE;
  NAWS_REQUIRE_PTR(X); NAWS_REQUIRE_PTR(L); NAWS_REQUIRE_PTR(dX);
  hipLaunchKernelGGL(wce_bwd_kernel, dim3((unsigned)naws_cdiv((int64_t)N * C, 64), nprob), dim3(64), 0,
                     (hipStream_t)stream, X, L, W, dY, N, C, is_mean, dX, lab_period, dy_const);
  return naws_check_launch();
}

extern "C" int naws_weighted_ce_bwd(const float* X, const float* L, const float* W,
                                    const float* dY, int N, int C, int is_mean, int nprob,
                                    float* dX, void* stream) {
  if (N <= 0 || C <= 0 || nprob <= 0 || nprob > 65535) return NAWS_ERR_SHAPE;
  NAWS_REQUIRE_PTR(X); NAWS_REQUIRE_PTR(L); NAWS_REQUIRE_PTR(dY); NAWS_REQUIRE_PTR(dX);
  hipLaunchKernelGGL(wce_bwd_kernel, dim3((unsigned)naws_cdiv((int64_t)N * C, 64), nprob), dim3(64), 0,
                     (hipStream_t)stream, X, L, W, dY, N, C, is_mean, dX, nprob, 1.0f);
  return naws_check_launch();
}

extern "C" int naws_stat_accumulate(const float* I, const float* L, int n, int init, float* AI,
                                    float* AL, void* stream) {
  if (n <= 0) return NAWS_ERR_SHAPE;
  NAWS_REQUIRE_PTR(I); NAWS_REQUIRE_PTR(L); NAWS_REQUIRE_PTR(AI); NAWS_REQUIRE_PTR(AL);
  hipLaunchKernelGGL(stat_kernel, dim3((unsigned)naws_cdiv(n, 64)), dim3(64), 0,
                     (hipStream_t)stream, I, L, n, init, AI, AL);
  return naws_check_launch();
}

extern "C" int naws_acm_sgd_update_rowmax(const float* grad, float* momentum_buf, const float* lr,
                                          float* param, float* acmgrad, int64_t total,
                                          const int64_t* seg_end, const float* seg_lr_mult,
                                          const float* seg_wd, int nseg, float momentum,
                                          int nesterov, int iter_size, int gpu_num,
                                          int64_t iter_count, uint32_t* rowmax,
                                          const int64_t* rm_table_host, int n_rm, void* stream) {
  if (total <= 0 || nseg <= 0 || iter_size <= 0 || gpu_num <= 0 || iter_count < 0)
    return NAWS_ERR_SHAPE;
  if (total % 4 != 0) return NAWS_ERR_ARG;
  NAWS_REQUIRE_PTR(grad); NAWS_REQUIRE_PTR(momentum_buf); NAWS_REQUIRE_PTR(lr);
  NAWS_REQUIRE_PTR(param); NAWS_REQUIRE_PTR(seg_end);
  if (acmgrad == nullptr && iter_size != 1) return NAWS_ERR_NULL;
  NAWS_REQUIRE_PTR(seg_lr_mult); NAWS_REQUIRE_PTR(seg_wd);
  if ((((uintptr_t)grad | (uintptr_t)momentum_buf | (uintptr_t)param | (uintptr_t)acmgrad) % 16))
    return NAWS_ERR_ARG;
  SgdRowmaxRegions rm{};
  if (rowmax) {
    if (n_rm <= 0 || n_rm > 4) return NAWS_ERR_ARG;
    NAWS_REQUIRE_PTR(rm_table_host);
    for (int r = 0; r < n_rm; ++r) {
      const int64_t* t = rm_table_host + 4 * r;
      if (t[0] < 0 || t[1] > total || t[0] >= t[1] || t[2] <= 0 || t[3] < 0) return NAWS_ERR_SHAPE;
      if (t[0] % 256 != 0 || t[2] % 256 != 0 || (t[1] - t[0]) % t[2] != 0) return NAWS_ERR_ARG;
      rm.start[r] = t[0]; rm.end[r] = t[1]; rm.rowlen[r] = t[2]; rm.out[r] = t[3];
    }
    rm.n = n_rm;
  }
  const int64_t total4 = total / 4;
  const int do_update = ((iter_count + 1) % iter_size == 0) ? 1 : 0;
  const float scale = (float)(1.0 / ((double)iter_size * (double)gpu_num));
  const int blocks = (int)std::min<int64_t>(naws_cdiv(total4, TB), 256 * 16);
  hipLaunchKernelGGL(acm_sgd_kernel, dim3(blocks), dim3(TB), 0, (hipStream_t)stream,
                     (const float4*)grad, (float4*)momentum_buf, lr, (float4*)param, (float4*)acmgrad,
                     total4, seg_end, seg_lr_mult, seg_wd, nseg, momentum, nesterov, scale,
                     do_update, iter_count == 0 ? 1 : 0, (unsigned*)(do_update ? rowmax : nullptr),
                     rm);
  return naws_check_launch();
}

extern "C" int naws_acm_sgd_update(const float* grad, float* momentum_buf, const float* lr,
                                   float* param, float* acmgrad, int64_t total,
                                   const int64_t* seg_end, const float* seg_lr_mult,
                                   const float* seg_wd, int nseg, float momentum, int nesterov,
                                   int iter_size, int gpu_num, int64_t iter_count, void* stream) {
  return naws_acm_sgd_update_rowmax(grad, momentum_buf, lr, param, acmgrad, total, seg_end,
                                    seg_lr_mult, seg_wd, nseg, momentum, nesterov, iter_size,
                                    gpu_num, iter_count, nullptr, nullptr, 0, stream);
}

extern "C" int naws_acm_sgd_update_planes(int format, const float* grad, float* momentum_buf,
                                          const float* lr, float* param, int64_t total,
                                          const int64_t* seg_end,
                                         const float* seg_lr_mult, const float* seg_wd, int nseg,
                                         float momentum, int nesterov, int gpu_num,
                                         int64_t iter_count, const naws_sgd_plane_region* regions,
                                         int n_regions, int32_t* overflow, int32_t overflow_tag,
                                         void* stream) {
  if (total <= 0 || nseg <= 0 || gpu_num <= 0 || iter_count < 0) return NAWS_ERR_SHAPE;
  if (total % 4 != 0 || n_regions <= 0 || n_regions > 4) return NAWS_ERR_ARG;
  if (format < NAWS_PLANES_F16X2 || format > NAWS_PLANES_BF16) return NAWS_ERR_ARG;
  const bool scaled = format == NAWS_PLANES_F16X2;
  NAWS_REQUIRE_PTR(grad); NAWS_REQUIRE_PTR(momentum_buf); NAWS_REQUIRE_PTR(lr);
  NAWS_REQUIRE_PTR(param); NAWS_REQUIRE_PTR(seg_end); NAWS_REQUIRE_PTR(seg_lr_mult);
  NAWS_REQUIRE_PTR(seg_wd); NAWS_REQUIRE_PTR(regions);
  if (scaled) NAWS_REQUIRE_PTR(overflow);
  if ((((uintptr_t)grad | (uintptr_t)momentum_buf | (uintptr_t)param) % 16)) return NAWS_ERR_ARG;
  SgdPlaneArgs a{};
  long long tiles = 0, cursor = 0;
  int n_lin = 0;
  long long lin_blocks = 0;
  a.lin_block0[0] = 0;
  auto add_linear = [&](long long s4, long long e4) {       // float4 units
    if (e4 <= s4) return true;
    if (n_lin == 4) return false;
    a.lin_start[n_lin] = s4; a.lin_end[n_lin] = e4;
    lin_blocks += std::min<long long>(naws_cdiv(e4 - s4, TB), 1024);
    a.lin_block0[++n_lin] = (int)lin_blocks;
    return true;
  };
  int nr = 0;
  for (int i = 0; i < n_regions; ++i) {
    const naws_sgd_plane_region& g = regions[i];
    if (g.rows <= 0 || g.cols <= 0 || g.rows_per_batch <= 0) return NAWS_ERR_SHAPE;
    const long long n = (long long)g.rows * g.cols;
    if (g.start < cursor || g.start + n > total) return NAWS_ERR_SHAPE;     // ascending, disjoint
    if (g.start % 4 != 0 || n % 4 != 0) return NAWS_ERR_UNSUPPORTED;
    if (!add_linear(cursor / 4, g.start / 4)) return NAWS_ERR_UNSUPPORTED;
    cursor = g.start + n;
    if (g.planes == nullptr) continue;     // updated elsewhere (naws_gemm_f32_f16x2_nt_xk_sgd): left alone
    // rows < rows_per_batch: a block of rows of ONE batch item of a larger matrix (planes points at
    // the block's first row inside that matrix's planes; the sharded update, engine.py)
    if (g.rows % 32 != 0 || g.cols % 256 != 0 || g.rows_per_batch % 32 != 0 ||
        (g.rows > g.rows_per_batch && g.rows % g.rows_per_batch != 0))
      return NAWS_ERR_UNSUPPORTED;
    if (g.rows < g.rows_per_batch && g.colmax) return NAWS_ERR_ARG;
    if (scaled) {
      NAWS_REQUIRE_PTR(g.bound); NAWS_REQUIRE_PTR(g.rowmax); NAWS_REQUIRE_PTR(g.inv_scale);
      if (g.bound == g.rowmax) return NAWS_ERR_ARG;
    }
    if (((uintptr_t)g.planes & 15) != 0) return NAWS_ERR_ARG;
    if (format == NAWS_PLANES_BF16 && g.cols % 64 != 0) return NAWS_ERR_UNSUPPORTED;
    SgdPlaneRegion& r = a.r[nr++];
    r.start = g.start; r.rows = g.rows; r.cols = g.cols; r.rows_per_batch = g.rows_per_batch;
    r.tile0 = (int)tiles; r.planes = (unsigned short*)g.planes; r.plane_stride = g.plane_stride;
    r.bound = g.bound; r.rowmax = g.rowmax; r.inv_scale = g.inv_scale;
    r.colmax = scaled ? g.colmax : nullptr;
    tiles += (long long)(g.rows / 32) * (g.cols / 256);
  }
  if (!add_linear(cursor / 4, total / 4)) return NAWS_ERR_UNSUPPORTED;
  if (tiles + lin_blocks > 0x7fffffffLL) return NAWS_ERR_UNSUPPORTED;
  a.n = nr; a.tiles = (int)tiles; a.n_lin = n_lin;
  for (int k = n_lin + 1; k < 5; ++k) a.lin_block0[k] = (int)lin_blocks;
  const float scale = (float)(1.0 / (double)gpu_num);
  // 256 resident workgroups walk the tiles of the two- and three-plane formats instead of one
  // workgroup per tile (whose 38 / 51 KB LDS images leave the conv body beside the update no room).
  // In-process A/B on the route with a gradient in between (tools/ab_engine.py --env sgd_wgs):
  // fp16x2 13.01 -> 12.88 ms/step, fp32x3 21.61 -> 21.52; the bf16 plan's 17 KB images co-reside as
  // they are (walking: slower).  Knob "sgd_wgs": n > 0 forces n x 256 workgroups, -1 one workgroup
  // per tile, n < -1 exactly -n workgroups (the tests walk several tiles per workgroup on small
  // arenas).
  const long long blocks = tiles + lin_blocks;
  int per_cu = naws_knob(NAWS_KNOB_SGD_WGS);
  if (per_cu == 0) per_cu = format == NAWS_PLANES_BF16 ? -1 : 1;
  const long long grid = per_cu > 0 ? std::min<long long>(blocks, (long long)per_cu * 256)
                                    : (per_cu < -1 ? std::min<long long>(blocks, -per_cu) : blocks);
#define NAWS_SGD_PLANES(F)                                                                          \
  if (grid == blocks)                                                                               \
    hipLaunchKernelGGL(acm_sgd_planes_kernel<F>, dim3((unsigned)blocks), dim3(TB), 0,                \
                       (hipStream_t)stream, (const float4*)grad, (float4*)momentum_buf, lr,          \
                       (float4*)param, seg_end, seg_lr_mult, seg_wd, nseg, momentum, nesterov,       \
                       scale, iter_count == 0 ? 1 : 0, a, overflow, overflow_tag);                   \
  else                                                                                              \
    hipLaunchKernelGGL(acm_sgd_planes_walk_kernel<F>, dim3((unsigned)grid), dim3(TB), 0,             \
                       (hipStream_t)stream, (const float4*)grad, (float4*)momentum_buf, lr,          \
                       (float4*)param, seg_end, seg_lr_mult, seg_wd, nseg, momentum, nesterov,       \
                       scale, iter_count == 0 ? 1 : 0, a, overflow, overflow_tag, (int)blocks)
  if (format == NAWS_PLANES_F16X2) { NAWS_SGD_PLANES(0); }
  else if (format == NAWS_PLANES_BF16X3) { NAWS_SGD_PLANES(1); }
  else { NAWS_SGD_PLANES(2); }
#undef NAWS_SGD_PLANES
  return naws_check_launch();
}

extern "C" int naws_acm_sgd_update_f16x2(const float* grad, float* momentum_buf, const float* lr,
                                         float* param, int64_t total, const int64_t* seg_end,
                                         const float* seg_lr_mult, const float* seg_wd, int nseg,
                                         float momentum, int nesterov, int gpu_num,
                                         int64_t iter_count, const naws_sgd_plane_region* regions,
                                         int n_regions, int32_t* overflow, int32_t overflow_tag,
                                         void* stream) {
  return naws_acm_sgd_update_planes(NAWS_PLANES_F16X2, grad, momentum_buf, lr, param, total, seg_end,
                                    seg_lr_mult, seg_wd, nseg, momentum, nesterov, gpu_num,
                                    iter_count, regions, n_regions, overflow, overflow_tag, stream);
}

extern "C" int naws_min_entropy_loss_fwd(const float* X, const float* L, int N, int C, float* Y,
                                         void* stream) {
  if (N <= 0 || C <= 0) return NAWS_ERR_SHAPE;
  NAWS_REQUIRE_PTR(X); NAWS_REQUIRE_PTR(L); NAWS_REQUIRE_PTR(Y);
  hipLaunchKernelGGL(min_entropy_fwd_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, X, L, N, C, Y);
  return naws_check_launch();
}

extern "C" int naws_min_entropy_loss_bwd(const float* X, const float* L, const float* dY, int N,
                                         int C, float* dX, void* stream) {
  if (N <= 0 || C <= 0) return NAWS_ERR_SHAPE;
  NAWS_REQUIRE_PTR(X); NAWS_REQUIRE_PTR(L); NAWS_REQUIRE_PTR(dY); NAWS_REQUIRE_PTR(dX);
  const long long total = (long long)N * C;
  hipLaunchKernelGGL(min_entropy_bwd_kernel, dim3((unsigned)std::min<long long>(naws_cdiv(total, 256), 1024)),
                     dim3(256), 0, (hipStream_t)stream, X, L, dY, N, C, dX);
  return naws_check_launch();
}

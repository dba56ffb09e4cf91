// the adjacent loss operators one cfg flag from the hot path.
//   RoILabel            ref: detectron/ops/roi_label_op.cc:10-123        (OICR pseudo labels)
//   SoftmaxWithLossN    ref: detectron/ops/softmax_with_loss_n_op.cc:152-357 (+ .cu:…: same maths)
//   RoIEntropy          ref: detectron/ops/roi_entropy_op.cu:24-112
// All of them are small, latency-bound reductions over <= a few thousand proposals x <= 81 classes:
// one or two launches each, every floating-point sum in a FIXED tree order (no float atomics), so
// results are reproducible run to run - the reference's CPU loops are serial, its CUDA RoIEntropy
// uses atomics.
#include <float.h>
#include "naws_common.h"

namespace {

constexpr int TB = 256;
constexpr int MAX_PICKS = 1024;    // labelled classes x top_k

// (value, index) argmax with "first index wins ties" = the reference's strict '<' scan order.
struct Best {
  float v;
  int i;
};
__device__ __forceinline__ Best better(Best a, Best b) {
  if (b.i < 0) return a;
  if (a.i < 0) return b;
  if (b.v > a.v || (b.v == a.v && b.i < a.i)) return b;
  return a;
}
__device__ __forceinline__ Best block_best(Best x, Best* sh) {
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) {
    Best o;
    o.v = __shfl_xor(x.v, d);
    o.i = __shfl_xor(x.i, d);
    x = better(x, o);
  }
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[w] = x;
  __syncthreads();
  Best r = sh[0];
  for (int k = 1; k < TB / 64; ++k) r = better(r, sh[k]);
  return r;
}

// fixed-order block sum (every thread gets the result)
__device__ __forceinline__ float block_sum(float v, float* sh) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  float r = sh[0];
  for (int k = 1; k < TB / 64; ++k) r += sh[k];
  return r;
}

// ---- RoILabel ----------------------------------------------------------------------------------
// picks: int32 [1 + 3 * max_picks] = {count, n[max], c[max], p bits[max]}
__global__ __launch_bounds__(TB) void roi_label_pick_kernel(const float* __restrict__ S,
                                                            const float* __restrict__ L, int n,
                                                            int cs, int c, int top_k,
                                                            int* __restrict__ picks, int max_picks) {
  __shared__ Best sh[TB / 64];
  __shared__ int s_cnt;
  __shared__ int hn[MAX_PICKS];                       // the shared picked list (exclusion test)
  int* gn = picks + 1;
  int* hc = picks + 1 + max_picks;
  int* hp = picks + 1 + 2 * max_picks;
  const int off = cs - c;
  if (threadIdx.x == 0) s_cnt = 0;
  __syncthreads();
  for (int cc = 0; cc < c; ++cc) {
    if (L[cc] != 1.f) continue;                       // block-uniform
    for (int k = 0; k < top_k; ++k) {
      const int cnt = s_cnt;
      Best b;
      b.v = -FLT_MAX; b.i = -1;
      for (int i = threadIdx.x; i < n; i += TB) {
        const float v = S[(size_t)i * cs + cc + off];
        if (b.v < v) {                                // strict: -FLT_MAX and NaN never win
          bool seen = false;
          for (int j = 0; j < cnt; ++j) seen |= (hn[j] == i);
          if (!seen) { b.v = v; b.i = i; }
        }
      }
      b = block_best(b, sh);
      if (threadIdx.x == 0) {
        hn[cnt] = b.i; gn[cnt] = b.i; hc[cnt] = cc;
        hp[cnt] = __float_as_int(b.i >= 0 ? b.v : -FLT_MAX);
        s_cnt = cnt + 1;
      }
      __syncthreads();
    }
  }
  if (threadIdx.x == 0) picks[0] = s_cnt;
}

__global__ __launch_bounds__(TB) void roi_label_assign_kernel(
    const float* __restrict__ U, const float* __restrict__ CW, int n, const int* __restrict__ picks,
    int max_picks, float fg_thresh, float bg_hi, float bg_lo, int32_t* __restrict__ RL,
    float* __restrict__ RW, float* __restrict__ part /* [blocks][4] */) {
  __shared__ float sh[TB / 64];
  const int cnt = picks[0];
  const int* hn = picks + 1;
  const int* hc = picks + 1 + max_picks;
  const int* hp = picks + 1 + 2 * max_picks;
  const int i = blockIdx.x * TB + threadIdx.x;
  float st[4] = {0.f, 0.f, 0.f, 0.f};
  if (i < n) {
    float max_iou = -FLT_MAX;
    int max_idx = -1;
    for (int j = 0; j < cnt; ++j) {
      const int g = hn[j];
      if (g < 0) continue;
      const float u = U[(size_t)i * n + g];
      if (max_iou < u) { max_iou = u; max_idx = j; }
    }
    int lab = 0;
    float w = 0.f;
    if (max_idx >= 0) {
      lab = hc[max_idx];
      w = CW ? CW[lab] : __int_as_float(hp[max_idx]);
      if (max_iou >= fg_thresh) {
        lab += 1; st[0] = 1.f; st[2] = w;
      } else if (max_iou >= bg_lo && max_iou < bg_hi) {
        lab = 0; st[1] = 1.f; st[3] = w;
      } else {
        lab += 1; w = 0.f;
      }
    }
    RL[i] = lab;
    RW[i] = w;
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float s = block_sum(st[k], sh);
    if (threadIdx.x == 0) part[blockIdx.x * 4 + k] = s;
  }
}

__global__ void roi_label_stats_kernel(const float* __restrict__ part, int blocks,
                                       float* __restrict__ stats) {
  if (threadIdx.x < 4) {
    float s = 0.f;
    for (int b = 0; b < blocks; ++b) s += part[b * 4 + threadIdx.x];
    stats[threadIdx.x] += s;
  }
}

// ---- SoftmaxWithLossN --------------------------------------------------------------------------
__global__ __launch_bounds__(TB) void swl_fwd_kernel(const float* __restrict__ X,
                                                     const int32_t* __restrict__ T,
                                                     const float* __restrict__ W, int N, int D,
                                                     float* __restrict__ P,
                                                     float* __restrict__ part /* [blocks][2] */) {
  __shared__ float sh[TB / 64];
  const int i = blockIdx.x * TB + threadIdx.x;
  float li = 0.f, wi = 0.f;
  if (i < N) {
    const float* x = X + (size_t)i * D;
    float* p = P + (size_t)i * D;
    float m = x[0];
    for (int d = 1; d < D; ++d) m = x[d] > m ? x[d] : m;
    float s = 0.f;
    for (int d = 0; d < D; ++d) s += expf(x[d] - m);
    const float ls = logf(s);
    const int t = T[i];
    wi = W ? W[i] : 1.f;
    // a label outside [0, D) is the reference's ENFORCE (.cc:192): poison the loss instead
    li = (t >= 0 && t < D) ? -((x[t] - m) - ls) * wi : __int_as_float(0x7fc00000);
    for (int d = 0; d < D; ++d) p[d] = expf((x[d] - m) - ls);
  }
  const float a = block_sum(li, sh);
  const float b = block_sum(wi, sh);
  if (threadIdx.x == 0) { part[blockIdx.x * 2] = a; part[blockIdx.x * 2 + 1] = b; }
}

__global__ void swl_fwd_finish_kernel(const float* __restrict__ part, int blocks, float scale,
                                      float* __restrict__ loss) {
  if (threadIdx.x == 0) {
    float ls = 0.f, ws = 0.f;
    for (int b = 0; b < blocks; ++b) { ls += part[2 * b]; ws += part[2 * b + 1]; }
    loss[0] = ws != 0.f ? ls * scale / ws : 0.f;
  }
}

__global__ __launch_bounds__(TB) void swl_count_kernel(const float* __restrict__ W, int N,
                                                       float* __restrict__ total) {
  __shared__ float sh[TB / 64];
  float c = 0.f;
  for (int i = threadIdx.x; i < N; i += TB) c += ((double)W[i] > 1e-12) ? 1.f : 0.f;   // (the reference compares in double)
  c = block_sum(c, sh);
  if (threadIdx.x == 0) total[0] = c;
}

__global__ __launch_bounds__(TB) void swl_bwd_kernel(const int32_t* __restrict__ T,
                                                     const float* __restrict__ W,
                                                     const float* __restrict__ P,
                                                     const float* __restrict__ dloss,
                                                     const float* __restrict__ total_w, int N, int D,
                                                     float scale, float* __restrict__ dX) {
  const long long k = (long long)blockIdx.x * TB + threadIdx.x;
  if (k >= (long long)N * D) return;
  const int i = (int)(k / D), d = (int)(k - (long long)i * D);
  float v = P[k];
  if (d == T[i]) v -= 1.0f;
  if (W) v *= W[i];
  const float total = W ? total_w[0] : (float)N;
  if (total > 0.f) v *= scale / total * dloss[0];
  dX[k] = v;
}

// ---- RoIEntropy --------------------------------------------------------------------------------
// one block per class: fixed-order sums of the class's count / score sum, then of p log p
__global__ __launch_bounds__(TB) void roi_entropy_kernel(const float* __restrict__ S,
                                                         const float* __restrict__ C, int n,
                                                         int off, float* __restrict__ E) {
  __shared__ float sh[TB / 64];
  const int c = blockIdx.x;
  float cnt = 0.f, cs = 0.f;
  for (int i = threadIdx.x; i < n; i += TB)
    if ((int)C[i] + off == c) { cnt += 1.f; cs += S[i]; }
  cnt = block_sum(cnt, sh);
  cs = block_sum(cs, sh);
  float e = 0.f;
  if (cnt != 1.f) {
    const float ln = logf(cnt);
    for (int i = threadIdx.x; i < n; i += TB)
      if ((int)C[i] + off == c) {
        const float p = S[i] / cs;
        e += (-1.0f * -1.0f * p * logf(p)) / ln;
      }
  }
  e = block_sum(e, sh);
  if (threadIdx.x == 0) E[c] = 1.f + e;
}

// mean_ += E where E != 1 (Add_A_not_1, roi_entropy_op.cu:18-27)
__global__ void roi_entropy_mean_kernel(const float* __restrict__ E, int nc, int init,
                                        float* __restrict__ mean) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= nc) return;
  float m = init ? 0.f : mean[c];
  if (E[c] != 1.f) m = E[c] + m;
  mean[c] = m;
}

}  // namespace

extern "C" int64_t naws_roi_label_workspace_bytes(int n, int c, int top_k) {
  if (n <= 0 || c <= 0 || top_k <= 0) return 0;
  const int64_t picks = 1 + 3 * (int64_t)c * top_k;
  return (picks + 3) / 4 * 16 + (int64_t)naws_cdiv(n, TB) * 4 * sizeof(float);
}

extern "C" int naws_roi_label_fwd(const float* S, const float* U, const float* L, const float* CW,
                                  int n, int cs, int c, float fg_thresh, float bg_thresh_hi,
                                  float bg_thresh_lo, int top_k, int num_pos, int num_neg,
                                  void* workspace, int32_t* RL, float* RW, float* stats,
                                  void* stream) {
  // ENFORCE sites roi_label_op.cc:16-23
  if (n <= 0 || c <= 0 || top_k <= 0 || !(cs == c || cs == c + 1)) return NAWS_ERR_SHAPE;
  NAWS_REQUIRE_PTR(S); NAWS_REQUIRE_PTR(U); NAWS_REQUIRE_PTR(L); NAWS_REQUIRE_PTR(workspace);
  NAWS_REQUIRE_PTR(RL); NAWS_REQUIRE_PTR(RW); NAWS_REQUIRE_PTR(stats);
  // binding caps depend on the reference's time-seeded shuffle (:62-70): not reproducible
  if (num_pos < n || num_neg < n || (int64_t)c * top_k > MAX_PICKS) return NAWS_ERR_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  const int max_picks = c * top_k;
  int* picks = (int*)workspace;
  float* part = (float*)((char*)workspace + ((1 + 3 * (int64_t)max_picks) + 3) / 4 * 16);
  const int blocks = (int)naws_cdiv(n, TB);
  hipLaunchKernelGGL(roi_label_pick_kernel, dim3(1), dim3(TB), 0, s, S, L, n, cs, c, top_k, picks,
                     max_picks);
  hipLaunchKernelGGL(roi_label_assign_kernel, dim3(blocks), dim3(TB), 0, s, U, CW, n, picks,
                     max_picks, fg_thresh, bg_thresh_hi, bg_thresh_lo, RL, RW, part);
  hipLaunchKernelGGL(roi_label_stats_kernel, dim3(1), dim3(64), 0, s, part, blocks, stats);
  return naws_check_launch();
}

extern "C" int64_t naws_softmax_with_loss_n_workspace_floats(int N) {
  return N > 0 ? 2 * naws_cdiv(N, TB) + 4 : 0;
}

extern "C" int naws_softmax_with_loss_n_fwd(const float* X, const int32_t* T, const float* W, int N,
                                            int D, float scale, float* workspace, float* P,
                                            float* loss, void* stream) {
  if (N <= 0 || D <= 0) return NAWS_ERR_SHAPE;
  if (!(scale >= 0.f)) return NAWS_ERR_ARG;            // CAFFE_ENFORCE(scale_ >= 0), .h:33
  NAWS_REQUIRE_PTR(X); NAWS_REQUIRE_PTR(T); NAWS_REQUIRE_PTR(P); NAWS_REQUIRE_PTR(loss);
  NAWS_REQUIRE_PTR(workspace);
  hipStream_t s = (hipStream_t)stream;
  const int blocks = (int)naws_cdiv(N, TB);
  hipLaunchKernelGGL(swl_fwd_kernel, dim3(blocks), dim3(TB), 0, s, X, T, W, N, D, P, workspace);
  hipLaunchKernelGGL(swl_fwd_finish_kernel, dim3(1), dim3(64), 0, s, workspace, blocks, scale, loss);
  return naws_check_launch();
}

extern "C" int naws_softmax_with_loss_n_bwd(const int32_t* T, const float* W, const float* P,
                                            const float* dloss, int N, int D, float scale,
                                            float* workspace, float* dX, void* stream) {
  if (N <= 0 || D <= 0) return NAWS_ERR_SHAPE;
  if (!(scale >= 0.f)) return NAWS_ERR_ARG;
  NAWS_REQUIRE_PTR(T); NAWS_REQUIRE_PTR(P); NAWS_REQUIRE_PTR(dloss); NAWS_REQUIRE_PTR(dX);
  NAWS_REQUIRE_PTR(workspace);
  hipStream_t s = (hipStream_t)stream;
  if (W) hipLaunchKernelGGL(swl_count_kernel, dim3(1), dim3(TB), 0, s, W, N, workspace);
  const long long total = (long long)N * D;
  hipLaunchKernelGGL(swl_bwd_kernel, dim3((unsigned)naws_cdiv(total, TB)), dim3(TB), 0, s, T, W, P,
                     dloss, workspace, N, D, scale, dX);
  return naws_check_launch();
}

extern "C" int naws_roi_entropy_fwd(const float* S, const float* C, int n, int num_classes,
                                    int rm_bg, float* E, float* mean, int init, void* stream) {
  if (n < 0 || num_classes <= 0) return NAWS_ERR_SHAPE;
  NAWS_REQUIRE_PTR(E);
  if (n > 0) { NAWS_REQUIRE_PTR(S); NAWS_REQUIRE_PTR(C); }
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(roi_entropy_kernel, dim3(num_classes), dim3(TB), 0, s, S, C, n, rm_bg ? -1 : 0, E);
  if (mean)
    hipLaunchKernelGGL(roi_entropy_mean_kernel, dim3((unsigned)naws_cdiv(num_classes, 64)), dim3(64),
                       0, s, E, num_classes, init, mean);
  return naws_check_launch();
}

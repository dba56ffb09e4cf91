// inference post-processing on the GPU (BASELINE configs[4]: 10-pass TTA,
// 4000 proposals).  Everything between the image upload and the final detections stays in HBM:
//
//   naws_roi_dedup_fwd      ref: detectron/core/test_wsl.py:998-1026 (_project_im_rois: float64
//                           product, float32 blob), :293-301 (box flip), :125-133 (dedup hash
//                           round(rois * DEDUP_BOXES) . [1, 1e3, 1e6, 1e9, 1e12], np.unique with
//                           return_index / return_inverse).  One workgroup per TTA pass: project,
//                           (flip,) hash, bitonic-sort the 64-bit (hash, index) keys in LDS, flag
//                           group starts, scan -> unique rois in ascending-hash order (np.unique's
//                           order), first-occurrence index and inverse map.
//   naws_tta_accumulate     ref: :173-176 (scores[inv_index]) + :260-261 (np.mean over the passes,
//                           accumulated pass by pass in float32 like numpy's axis-0 reduction;
//                           the division is numpy's: double quotient rounded to float32).
//   naws_det_limit_fwd      ref: :803-863 (box_results_with_nms_and_limit after the per-class NMS):
//                           the image-wide DETECTIONS_PER_IM cut - threshold = the limit-th
//                           largest kept score (np.sort(all)[-limit]), keep score >= threshold -
//                           and compaction to (class, row, score) triples in the reference's
//                           order (class ascending, row ascending).
//
// All integer / index outputs are bit-identical to the host (numpy) path they replace; scores are
// float32 sums in the same order.
#include <float.h>
#include "naws_common.h"

namespace {

constexpr int DT = 1024;                       // threads of the dedup workgroup
constexpr int MAXN = 16384;                    // proposals per pass (keys live in LDS: 128 KB)
constexpr long long HASH_BIAS = 1LL << 49;     // hashes are integers of magnitude < 2^49

__device__ __forceinline__ void project_roi(const float* __restrict__ boxes, int i, int flip,
                                            float im_width, double im_scale, float* r) {
  float x1 = boxes[i * 4 + 0], y1 = boxes[i * 4 + 1], x2 = boxes[i * 4 + 2], y2 = boxes[i * 4 + 3];
  if (flip) {            // boxes_hf[:, 0::4] = w - boxes[:, 2::4] - 1 (float32 arithmetic)
    const float nx1 = im_width - x2 - 1.f, nx2 = im_width - x1 - 1.f;
    x1 = nx1; x2 = nx2;
  }
  r[0] = (float)((double)x1 * im_scale);
  r[1] = (float)((double)y1 * im_scale);
  r[2] = (float)((double)x2 * im_scale);
  r[3] = (float)((double)y2 * im_scale);
}

struct DedupPass {
  double im_scale;
  float im_width;
  int flip;
  float batch_index;
};

__global__ __launch_bounds__(DT) void roi_dedup_kernel(
    const float* __restrict__ boxes, const float* __restrict__ obn, int n, int np2,
    const DedupPass* __restrict__ passes, float dedup, float* __restrict__ rois_out,
    float* __restrict__ obn_out, int* __restrict__ index_out, int* __restrict__ inv_out,
    int* __restrict__ count_out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  unsigned long long* keys = reinterpret_cast<unsigned long long*>(smem_raw);   // [np2]
  int* part = reinterpret_cast<int*>(keys + np2);                               // [DT]
  const DedupPass ps = passes[blockIdx.x];
  float* rois_o = rois_out + (size_t)blockIdx.x * n * 5;
  float* obn_o = obn_out + (size_t)blockIdx.x * n;
  int* index_o = index_out + (size_t)blockIdx.x * n;
  int* inv_o = inv_out + (size_t)blockIdx.x * n;

  for (int i = threadIdx.x; i < np2; i += DT) {
    unsigned long long key = ~0ull;
    if (i < n) {
      float r[4];
      project_roi(boxes, i, ps.flip, ps.im_width, ps.im_scale, r);
      // np.round(float32 * 0.125): exact scaling, round half to even; the dot with
      // [1, 1e3, 1e6, 1e9, 1e12] is exact in float64 (every term and the sum < 2^53)
      const double h = (double)rintf(ps.batch_index * dedup) + (double)rintf(r[0] * dedup) * 1e3 +
                       (double)rintf(r[1] * dedup) * 1e6 + (double)rintf(r[2] * dedup) * 1e9 +
                       (double)rintf(r[3] * dedup) * 1e12;
      key = ((unsigned long long)((long long)h + HASH_BIAS) << 14) | (unsigned long long)i;
    }
    keys[i] = key;
  }
  __syncthreads();
  // bitonic sort, ascending (keys are unique: the index is part of the key)
  for (int k = 2; k <= np2; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int t = threadIdx.x; t < np2 / 2; t += DT) {
        const int lo = ((t / j) * 2 * j) + (t % j), hi = lo + j;
        const bool up = ((lo & k) == 0);
        const unsigned long long a = keys[lo], b = keys[hi];
        if ((a > b) == up) { keys[lo] = b; keys[hi] = a; }
      }
      __syncthreads();
    }
  }
  // group starts -> ids by a block scan over per-thread chunks
  const int chunk = np2 / DT > 0 ? np2 / DT : 1;
  const int begin = threadIdx.x * chunk;
  int cnt = 0;
  for (int i = begin; i < begin + chunk && i < n; ++i)
    cnt += (i == 0 || (keys[i] >> 14) != (keys[i - 1] >> 14)) ? 1 : 0;
  part[threadIdx.x] = cnt;
  __syncthreads();
  for (int d = 1; d < DT; d <<= 1) {                   // inclusive Hillis-Steele scan
    const int v = threadIdx.x >= d ? part[threadIdx.x - d] : 0;
    __syncthreads();
    part[threadIdx.x] += v;
    __syncthreads();
  }
  int gid = part[threadIdx.x] - cnt;                   // groups before this chunk
  for (int i = begin; i < begin + chunk && i < n; ++i) {
    const bool start = (i == 0 || (keys[i] >> 14) != (keys[i - 1] >> 14));
    if (start) ++gid;
    const int o = (int)(keys[i] & 16383ull);
    inv_o[o] = gid - 1;
    if (start) {
      float r[4];
      project_roi(boxes, o, ps.flip, ps.im_width, ps.im_scale, r);
      index_o[gid - 1] = o;
      rois_o[(gid - 1) * 5 + 0] = ps.batch_index;
      rois_o[(gid - 1) * 5 + 1] = r[0]; rois_o[(gid - 1) * 5 + 2] = r[1];
      rois_o[(gid - 1) * 5 + 3] = r[2]; rois_o[(gid - 1) * 5 + 4] = r[3];
      obn_o[gid - 1] = obn[o] + 1.0f;                  // np.add(obn_scores, 1.0), :1058
    }
  }
  if (threadIdx.x == DT - 1) count_out[blockIdx.x] = part[DT - 1];
}

__global__ void tta_accumulate_kernel(const float* __restrict__ S, const int* __restrict__ inv, int n,
                                      int k, int first, float* __restrict__ acc) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)n * k) return;
  const int r = (int)(i / k), c = (int)(i - (long long)r * k);
  const float v = S[(size_t)(inv ? inv[r] : r) * k + c];
  acc[i] = first ? v : acc[i] + v;
}

__global__ void tta_finish_kernel(float* __restrict__ acc, long long total, int npass) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < total) acc[i] = (float)((double)acc[i] / (double)npass);
}

// ---- DETECTIONS_PER_IM cut + compaction: one workgroup --------------------------------------
constexpr int LT = 1024;
__global__ __launch_bounds__(LT) void det_limit_kernel(const float* __restrict__ scores /* [R][K] */,
                                                       const unsigned char* __restrict__ keep /* [C][R] */,
                                                       int C, int R, int K, int limit, int cap,
                                                       int* __restrict__ out_count,
                                                       int* __restrict__ out_cls,
                                                       int* __restrict__ out_row,
                                                       float* __restrict__ out_score) {
  __shared__ int hist[256];
  __shared__ int part[LT];
  __shared__ unsigned s_prefix, s_mask;
  __shared__ int s_rank;
  const long long total = (long long)C * R;
  // kept count
  int cnt = 0;
  for (long long i = threadIdx.x; i < total; i += LT) cnt += keep[i] ? 1 : 0;
  part[threadIdx.x] = cnt;
  __syncthreads();
  for (int d = LT / 2; d > 0; d >>= 1) {
    if (threadIdx.x < d) part[threadIdx.x] += part[threadIdx.x + d];
    __syncthreads();
  }
  const int nkept = part[0];
  __syncthreads();
  unsigned thr_bits = 0;                                // keep everything
  if (limit > 0 && nkept > limit) {
    // radix select of the limit-th largest kept score (scores > 0: float order == bit order)
    if (threadIdx.x == 0) { s_prefix = 0; s_mask = 0; s_rank = limit; }
    __syncthreads();
    for (int shift = 24; shift >= 0; shift -= 8) {
      if (threadIdx.x < 256) hist[threadIdx.x] = 0;
      __syncthreads();
      const unsigned prefix = s_prefix, mask = s_mask;
      for (long long i = threadIdx.x; i < total; i += LT) {
        if (!keep[i]) continue;
        const int c = (int)(i / R), r = (int)(i - (long long)c * R);
        const unsigned b = __float_as_uint(scores[(size_t)r * K + c + 1]);
        if ((b & mask) == prefix) atomicAdd(&hist[(b >> shift) & 255], 1);
      }
      __syncthreads();
      if (threadIdx.x == 0) {
        int rank = s_rank, d = 255;
        for (; d > 0; --d) {
          if (hist[d] >= rank) break;
          rank -= hist[d];
        }
        s_rank = rank;
        s_prefix = prefix | ((unsigned)d << shift);
        s_mask = mask | (255u << shift);
      }
      __syncthreads();
    }
    thr_bits = s_prefix;
  }
  // compaction in (class, row) order
  const long long chunk = (total + LT - 1) / LT;
  const long long b0 = threadIdx.x * chunk, b1 = min(b0 + chunk, total);
  cnt = 0;
  for (long long i = b0; i < b1; ++i) {
    if (!keep[i]) continue;
    const int c = (int)(i / R), r = (int)(i - (long long)c * R);
    cnt += (__float_as_uint(scores[(size_t)r * K + c + 1]) >= thr_bits) ? 1 : 0;
  }
  __syncthreads();
  part[threadIdx.x] = cnt;
  __syncthreads();
  for (int d = 1; d < LT; d <<= 1) {
    const int v = threadIdx.x >= d ? part[threadIdx.x - d] : 0;
    __syncthreads();
    part[threadIdx.x] += v;
    __syncthreads();
  }
  int pos = part[threadIdx.x] - cnt;
  for (long long i = b0; i < b1; ++i) {
    if (!keep[i]) continue;
    const int c = (int)(i / R), r = (int)(i - (long long)c * R);
    const float s = scores[(size_t)r * K + c + 1];
    if (__float_as_uint(s) < thr_bits) continue;
    if (pos < cap) { out_cls[pos] = c + 1; out_row[pos] = r; out_score[pos] = s; }
    ++pos;
  }
  if (threadIdx.x == LT - 1) out_count[0] = part[LT - 1];
}

}  // namespace

extern "C" int naws_roi_dedup_fwd(const float* boxes, const float* obn_scores, int n, int npass,
                                  const void* passes, float dedup_boxes, float* rois_out,
                                  float* obn_out, int32_t* index_out, int32_t* inv_out,
                                  int32_t* count_out, void* stream) {
  if (n <= 0 || npass <= 0) return NAWS_ERR_SHAPE;
  if (n > MAXN) return NAWS_ERR_UNSUPPORTED;
  if (!(dedup_boxes > 0.f)) return NAWS_ERR_ARG;
  NAWS_REQUIRE_PTR(boxes); NAWS_REQUIRE_PTR(obn_scores); NAWS_REQUIRE_PTR(passes);
  NAWS_REQUIRE_PTR(rois_out); NAWS_REQUIRE_PTR(obn_out); NAWS_REQUIRE_PTR(index_out);
  NAWS_REQUIRE_PTR(inv_out); NAWS_REQUIRE_PTR(count_out);
  int np2 = DT;                                        // >= one element per thread
  while (np2 < n) np2 <<= 1;
  const size_t lds = (size_t)np2 * 8 + DT * sizeof(int);
  if (naws_allow_lds(roi_dedup_kernel) != NAWS_OK) return NAWS_ERR_LAUNCH;
  hipLaunchKernelGGL(roi_dedup_kernel, dim3(npass), dim3(DT), lds, (hipStream_t)stream, boxes,
                     obn_scores, n, np2, (const DedupPass*)passes, dedup_boxes, rois_out, obn_out,
                     index_out, inv_out, count_out);
  return naws_check_launch();
}

extern "C" int naws_tta_accumulate(const float* scores, const int32_t* inv_index, int n, int k,
                                   int first, float* acc, void* stream) {
  if (n <= 0 || k <= 0) return NAWS_ERR_SHAPE;
  NAWS_REQUIRE_PTR(scores); NAWS_REQUIRE_PTR(acc);
  const long long total = (long long)n * k;
  hipLaunchKernelGGL(tta_accumulate_kernel, dim3((unsigned)naws_cdiv(total, 256)), dim3(256), 0,
                     (hipStream_t)stream, scores, inv_index, n, k, first, acc);
  return naws_check_launch();
}

extern "C" int naws_tta_finish(float* acc, int64_t total, int npass, void* stream) {
  if (total <= 0 || npass <= 0) return NAWS_ERR_SHAPE;
  NAWS_REQUIRE_PTR(acc);
  hipLaunchKernelGGL(tta_finish_kernel, dim3((unsigned)naws_cdiv(total, 256)), dim3(256), 0,
                     (hipStream_t)stream, acc, (long long)total, npass);
  return naws_check_launch();
}

extern "C" int naws_det_limit_fwd(const float* scores, const uint8_t* keep, int C, int R, int K,
                                  int limit, int cap, int32_t* out_count, int32_t* out_cls,
                                  int32_t* out_row, float* out_score, void* stream) {
  if (C <= 0 || R <= 0 || K != C + 1 || cap <= 0 || limit < 0) return NAWS_ERR_SHAPE;
  NAWS_REQUIRE_PTR(scores); NAWS_REQUIRE_PTR(keep); NAWS_REQUIRE_PTR(out_count);
  NAWS_REQUIRE_PTR(out_cls); NAWS_REQUIRE_PTR(out_row); NAWS_REQUIRE_PTR(out_score);
  hipLaunchKernelGGL(det_limit_kernel, dim3(1), dim3(LT), 0, (hipStream_t)stream, scores, keep, C, R,
                     K, limit, cap, out_count, out_cls, out_row, out_score);
  return naws_check_launch();
}

// Per-class greedy NMS of the inference path on the GPU.
//
// replaces: detectron/utils/cython_nms.pyx:36-87 (`nms`), called once per class from
// detectron/core/test_wsl.py:803-863 (box_results_with_nms_and_limit).  Same arithmetic in the
// same order (fp32, +1 pixel areas, suppress when inter / (a_i + a_j - inter) >= thresh), so
// the kept set is bit-identical to the CPU loop for a given visiting order.
//
// All classes of an image go in one launch pair.  The caller passes, per class ("batch" item),
// its candidate boxes already gathered in descending-score order and the candidate count.
//   1. nms_mask_kernel: bit j of mask[b][i][w] = box 64w+j is suppressed by box i (j > i).
//      One workgroup = a 64 x 64 block of the (i, j) plane, column boxes staged in LDS.
//   2. nms_scan_kernel: one wave per class walks the rows in blocks of 64: the greedy decision
//      inside a block only needs the 64 x 64 diagonal bits (resolved in registers, lane = row),
//      then every lane ORs the kept rows' words of its own columns into the running
//      "removed" set - 64 independent loads per lane per block instead of a dependent load
//      per box.
#include <stdlib.h>
#include "naws_common.h"

namespace {

__device__ __forceinline__ bool nms_suppresses(const float4 a, const float4 b, float thresh) {
  // plain operators, each rounded on its own (built with -ffp-contract=fast-honor-pragmas)
#pragma clang fp contract(off)
  const float aw = a.z - a.x + 1.f, ah = a.w - a.y + 1.f;
  const float bw = b.z - b.x + 1.f, bh = b.w - b.y + 1.f;
  const float aa = aw * ah, ab = bw * bh;
  const float xx1 = a.x >= b.x ? a.x : b.x, yy1 = a.y >= b.y ? a.y : b.y;
  const float xx2 = a.z <= b.z ? a.z : b.z, yy2 = a.w <= b.w ? a.w : b.w;
  float w = xx2 - xx1 + 1.f, h = yy2 - yy1 + 1.f;
  w = 0.f >= w ? 0.f : w;
  h = 0.f >= h ? 0.f : h;
  const float inter = w * h;
  const float sum = aa + ab;
  const float uni = sum - inter;
  const float ovr = inter / uni;
  return ovr >= thresh;
}

__global__ __launch_bounds__(64) void nms_mask_kernel(const float4* __restrict__ boxes,
                                                      const int* __restrict__ counts, int n_max,
                                                      int words, float thresh,
                                                      unsigned long long* __restrict__ mask) {
  const int b = blockIdx.z, rb = blockIdx.y, cb = blockIdx.x;
  const int n = counts[b];
  if (rb * 64 >= n || cb < rb) return;           // upper triangle (incl. diagonal) only
  __shared__ float4 colbox[64];
  const float4* bx = boxes + (long long)b * n_max;
  const int c = cb * 64 + threadIdx.x;
  colbox[threadIdx.x] = c < n ? bx[c] : make_float4(0.f, 0.f, 0.f, 0.f);
  __syncthreads();
  const int i = rb * 64 + threadIdx.x;
  if (i >= n) return;
  const float4 me = bx[i];
  unsigned long long bits = 0;
  const int jmax = min(64, n - cb * 64);
  for (int j = 0; j < jmax; ++j) {
    const int col = cb * 64 + j;
    if (col > i && nms_suppresses(me, colbox[j], thresh)) bits |= 1ull << j;
  }
  mask[((long long)b * n_max + i) * words + cb] = bits;
}

// one wave per class; lane l owns removed-words l, l+64, ...
template <int WPL>
__global__ __launch_bounds__(64) void nms_scan_kernel(const unsigned long long* __restrict__ mask,
                                                      const int* __restrict__ counts, int n_max,
                                                      int words, int* __restrict__ keep) {
  const int b = blockIdx.x, lane = threadIdx.x;
  const int n = counts[b];
  const unsigned long long* mb = mask + (long long)b * n_max * words;
  int* kb = keep + (long long)b * n_max;
  unsigned long long removed[WPL];
#pragma unroll
  for (int k = 0; k < WPL; ++k) removed[k] = 0;
  const int nblk = (n + 63) / 64;
  for (int blk = 0; blk < nblk; ++blk) {
    // the removed-word of this block lives in lane (blk % 64), slot blk / 64
    unsigned long long rem = 0;
#pragma unroll
    for (int k = 0; k < WPL; ++k)
      if (k == blk / 64) rem = removed[k];
    rem = __shfl(rem, blk % 64);
    const int i = blk * 64 + lane;
    const unsigned long long diag = (i < n) ? mb[(long long)i * words + blk] : 0ull;
    // greedy walk inside the block: row r is kept iff not removed when reached
    unsigned long long keptbits = 0;
    for (int r = 0; r < 64 && blk * 64 + r < n; ++r) {
      const unsigned long long dr = __shfl(diag, r);
      if (!((rem >> r) & 1ull)) {
        keptbits |= 1ull << r;
        rem |= dr;
      }
    }
    if (i < n) kb[i] = (int)((keptbits >> lane) & 1ull);
    // fold the kept rows into the running removed set (columns beyond this block)
#pragma unroll
    for (int k = 0; k < WPL; ++k) {
      const int w = lane + k * 64;
      if (w > blk && w < words) {
        unsigned long long acc = 0;
        unsigned long long kbits = keptbits;     // wave-uniform
        const unsigned long long* col = mb + (long long)blk * 64 * words + w;
        while (kbits) {                          // 8 independent loads in flight per round
          unsigned long long v[8];
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            const int r = kbits ? __ffsll((long long)kbits) - 1 : -1;
            kbits &= kbits - 1;                  // 0 stays 0
            v[q] = r >= 0 ? col[(long long)r * words] : 0ull;
          }
#pragma unroll
          for (int q = 0; q < 8; ++q) acc |= v[q];
        }
        removed[k] |= acc;
      }
    }
  }
  for (int i = n + lane; i < n_max; i += 64) kb[i] = 0;
}

}  // namespace

extern "C" int64_t naws_nms_workspace_bytes(int batch, int n_max) {
  if (batch <= 0 || n_max <= 0) return 0;
  return (int64_t)batch * n_max * ((n_max + 63) / 64) * 8;
}

extern "C" int naws_nms_sorted_fwd(const float* boxes, const int32_t* counts, int batch, int n_max,
                                   float thresh, void* workspace, int32_t* keep, void* stream) {
  if (batch <= 0 || n_max <= 0) return NAWS_ERR_SHAPE;
  NAWS_REQUIRE_PTR(boxes); NAWS_REQUIRE_PTR(counts); NAWS_REQUIRE_PTR(workspace); NAWS_REQUIRE_PTR(keep);
  if (((uintptr_t)boxes & 15) != 0 || ((uintptr_t)workspace & 7) != 0) return NAWS_ERR_ARG;
  const int words = (n_max + 63) / 64;
  if (batch > 65535 || words > 65535 || words > 64 * 4) return NAWS_ERR_UNSUPPORTED;   // n_max <= 16384
  hipStream_t s = (hipStream_t)stream;
  dim3 grid(words, words, batch);
  hipLaunchKernelGGL(nms_mask_kernel, grid, dim3(64), 0, s, (const float4*)boxes, counts, n_max,
                     words, thresh, (unsigned long long*)workspace);
  int rc = naws_check_launch();
  if (rc != NAWS_OK) return rc;
  const unsigned long long* m = (const unsigned long long*)workspace;
  if (words <= 64)
    hipLaunchKernelGGL(nms_scan_kernel<1>, dim3(batch), dim3(64), 0, s, m, counts, n_max, words, keep);
  else if (words <= 128)
    hipLaunchKernelGGL(nms_scan_kernel<2>, dim3(batch), dim3(64), 0, s, m, counts, n_max, words, keep);
  else
    hipLaunchKernelGGL(nms_scan_kernel<4>, dim3(batch), dim3(64), 0, s, m, counts, n_max, words, keep);
  return naws_check_launch();
}

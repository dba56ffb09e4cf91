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

// ---- Soft-NMS (TEST.SOFT_NMS) --------------------------------------------------------------------
// replaces: detectron/utils/cython_nms.pyx:98-203 (`soft_nms`), called per class from
// detectron/core/test_wsl.py:826-834.  The reference is a sequential in-place loop whose OUTPUT
// ORDER is part of its result (position i takes the first maximum of positions i..N-1 by a swap;
// a box whose decayed score falls below the threshold is overwritten by the LAST box and N
// shrinks), so this kernel keeps the whole list of one class in LDS and replays the rounds:
// one workgroup per class, per round
//   1. block arg-max over [i, N) (largest score, smallest position: the first maximum),
//   2. the swap,
//   3. every later box decayed in parallel (the same float expressions, un-contracted),
//   4. only if a box failed: the overwrite-by-the-last-box scan, which is a two-pointer
//      compaction - survivors below the new N stay, each hole (ascending) takes the last
//      remaining survivor from the tail - done with one block scan of (hole, filler) counts.
// Scores, kept indices and their order equal the oracle's statement-for-statement C loop.
constexpr int SN_T = 1024;

struct SoftNmsLds {
  float *x1, *y1, *x2, *y2, *sc;
  int *ind, *flag, *hole, *scan;
};

__device__ __forceinline__ float soft_nms_decay(float tx1, float ty1, float tx2, float ty2, float x1,
                                                float y1, float x2, float y2, float s, float sigma,
                                                float Nt, int method, bool& touched) {
#pragma clang fp contract(off)
  const float area = (x2 - x1 + 1) * (y2 - y1 + 1);
  const float iw = ((tx2 < x2 ? tx2 : x2) - (tx1 > x1 ? tx1 : x1) + 1);
  touched = false;
  if (iw > 0) {
    const float ih = ((ty2 < y2 ? ty2 : y2) - (ty1 > y1 ? ty1 : y1) + 1);
    if (ih > 0) {
      const float ua = (float)(double)((tx2 - tx1 + 1) * (ty2 - ty1 + 1) + area - iw * ih);
      const float ov = iw * ih / ua;
      float weight;
      if (method == 1) weight = ov > Nt ? 1 - ov : 1;
      else if (method == 2) weight = (float)exp((double)(-(ov * ov) / sigma));
      else weight = ov > Nt ? 0 : 1;
      touched = true;
      return weight * s;
    }
  }
  return s;
}

// block-wide exclusive scan of two counts packed as lo | hi << 16 (each total < 65536)
__device__ __forceinline__ unsigned sn_block_scan(unsigned v, int* scan, unsigned& total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned inc = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const unsigned t = __shfl_up(inc, d);
    if (lane >= d) inc += t;
  }
  __syncthreads();
  if (lane == 63) scan[wave] = (int)inc;
  __syncthreads();
  if (wave == 0) {
    unsigned w = lane < SN_T / 64 ? (unsigned)scan[lane] : 0u, winc = w;
#pragma unroll
    for (int d = 1; d < 16; d <<= 1) {
      const unsigned t = __shfl_up(winc, d);
      if (lane >= d) winc += t;
    }
    if (lane < SN_T / 64) scan[32 + lane] = (int)(winc - w);
    if (lane == SN_T / 64 - 1) scan[63] = (int)winc;
  }
  __syncthreads();
  total = (unsigned)scan[63];
  return inc - v + (unsigned)scan[32 + wave];
}

__global__ __launch_bounds__(SN_T) void soft_nms_kernel(const float* __restrict__ dets,
                                                        const int* __restrict__ counts, int n_max,
                                                        float sigma, float Nt, float threshold,
                                                        int method, float* __restrict__ out,
                                                        int* __restrict__ keep,
                                                        int* __restrict__ out_counts) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smx[];
  const int cls = blockIdx.x, tid = threadIdx.x;
  const int n0 = min(max(counts[cls], 0), n_max);
  float* x1 = reinterpret_cast<float*>(smx);
  float* y1 = x1 + n_max; float* x2 = y1 + n_max; float* y2 = x2 + n_max; float* sc = y2 + n_max;
  int* ind = reinterpret_cast<int*>(sc + n_max);
  int* flag = ind + n_max;           // 1 = failed this round
  int* hole = flag + n_max;          // positions of the holes, ascending
  int* scan = hole + n_max;          // 64 words of scan scratch + 4 of round state
  const float* d = dets + (size_t)cls * n_max * 5;
  for (int p = tid; p < n0; p += SN_T) {
    x1[p] = d[p * 5 + 0]; y1[p] = d[p * 5 + 1]; x2[p] = d[p * 5 + 2]; y2[p] = d[p * 5 + 3];
    sc[p] = d[p * 5 + 4]; ind[p] = p;
  }
  __syncthreads();
  int N = n0;
  for (int i = 0; i < N; ++i) {
    // ---- 1. first maximum of positions i .. N-1
    float bs = 0.f;
    int bp = 0x7fffffff;                               // (no candidate yet)
    for (int p = i + tid; p < N; p += SN_T) {          // ascending p: strict '>' keeps the first
      const float s = sc[p];
      if (bp == 0x7fffffff || s > bs) { bs = s; bp = p; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float os = __shfl_xor(bs, o);
      const int op = __shfl_xor(bp, o);
      if (op != 0x7fffffff && (bp == 0x7fffffff || os > bs || (os == bs && op < bp))) { bs = os; bp = op; }
    }
    __syncthreads();
    if ((tid & 63) == 0) { scan[(tid >> 6) * 2] = __float_as_int(bs); scan[(tid >> 6) * 2 + 1] = bp; }
    __syncthreads();
    if (tid == 0) {
      scan[64] = 0;                                    // this round's "a box failed" word
      float ms = __int_as_float(scan[0]);
      int mp = scan[1];
      for (int w = 1; w < SN_T / 64; ++w) {
        const float os = __int_as_float(scan[2 * w]);
        const int op = scan[2 * w + 1];
        if (op != 0x7fffffff && (mp == 0x7fffffff || os > ms || (os == ms && op < mp))) { ms = os; mp = op; }
      }
      // ---- 2. the swap (position i <-> the maximum's position)
      if (mp != i && mp != 0x7fffffff) {
        float t;
        int ti;
        t = x1[i]; x1[i] = x1[mp]; x1[mp] = t;  t = y1[i]; y1[i] = y1[mp]; y1[mp] = t;
        t = x2[i]; x2[i] = x2[mp]; x2[mp] = t;  t = y2[i]; y2[i] = y2[mp]; y2[mp] = t;
        t = sc[i]; sc[i] = sc[mp]; sc[mp] = t;  ti = ind[i]; ind[i] = ind[mp]; ind[mp] = ti;
      }
    }
    __syncthreads();
    // ---- 3. decay every later box against box i
    const float tx1 = x1[i], ty1 = y1[i], tx2 = x2[i], ty2 = y2[i];
    int nfail = 0;
    for (int p = i + 1 + tid; p < N; p += SN_T) {
      bool touched;
      const float ns = soft_nms_decay(tx1, ty1, tx2, ty2, x1[p], y1[p], x2[p], y2[p], sc[p], sigma,
                                      Nt, method, touched);
      int f = 0;
      if (touched) {
        sc[p] = ns;
        f = ns < threshold ? 1 : 0;
      }
      flag[p] = f;
      nfail += f;
    }
    // (no __syncthreads_or: it brings static LDS, and this kernel's dynamic limit is the full 160 KB)
    if (nfail) atomicOr(&scan[64], 1);
    __syncthreads();
    const int any_fail = scan[64];
    if (!any_fail) continue;
    // ---- 4. overwrite-by-the-last-box = two-pointer compaction.  Each thread owns a contiguous
    // chunk of [i+1, N): count the failures first (total -> the new N), then holes / fillers
    const int span = N - i - 1;
    const int chunk = (span + SN_T - 1) / SN_T;
    const int c0 = i + 1 + tid * chunk, c1 = min(c0 + chunk, N);
    unsigned fails = 0;
    for (int p = c0; p < c1; ++p) fails += (unsigned)flag[p];
    unsigned total;
    sn_block_scan(fails, scan, total);
    const int newN = N - (int)(total & 0xffffu);
    unsigned hf = 0;                                   // holes (lo) and fillers (hi) in my chunk
    for (int p = c0; p < c1; ++p) {
      if (flag[p] && p < newN) hf += 1u;
      if (!flag[p] && p >= newN) hf += 1u << 16;
    }
    unsigned tot2;
    const unsigned base = sn_block_scan(hf, scan, tot2);
    const int nholes = (int)(tot2 & 0xffffu);          // == number of fillers
    int hrank = (int)(base & 0xffffu);
    for (int p = c0; p < c1; ++p)
      if (flag[p] && p < newN) hole[hrank++] = p;
    __syncthreads();
    // fillers in DESCENDING position order take the holes in ascending order
    int frank = (int)(base >> 16);                     // fillers before my chunk (ascending rank)
    for (int p = c0; p < c1; ++p) {
      if (!flag[p] && p >= newN) {
        const int h = hole[nholes - 1 - frank];
        x1[h] = x1[p]; y1[h] = y1[p]; x2[h] = x2[p]; y2[h] = y2[p]; sc[h] = sc[p]; ind[h] = ind[p];
        ++frank;
      }
    }
    N = newN;
    __syncthreads();
  }
  float* o = out + (size_t)cls * n_max * 5;
  for (int p = tid; p < N; p += SN_T) {
    o[p * 5 + 0] = x1[p]; o[p * 5 + 1] = y1[p]; o[p * 5 + 2] = x2[p]; o[p * 5 + 3] = y2[p];
    o[p * 5 + 4] = sc[p];
    keep[(size_t)cls * n_max + p] = ind[p];
  }
  if (tid == 0) out_counts[cls] = N;
}

}  // namespace

extern "C" int naws_soft_nms_fwd(const float* dets, const int32_t* counts, int batch, int n_max,
                                 float sigma, float overlap_thresh, float score_thresh, int method,
                                 float* out_dets, int32_t* keep, int32_t* out_counts, void* stream) {
  if (batch <= 0 || n_max <= 0) return NAWS_ERR_SHAPE;
  if (method < 0 || method > 2 || !(sigma > 0.f)) return NAWS_ERR_ARG;
  NAWS_REQUIRE_PTR(dets); NAWS_REQUIRE_PTR(counts); NAWS_REQUIRE_PTR(out_dets);
  NAWS_REQUIRE_PTR(keep); NAWS_REQUIRE_PTR(out_counts);
  const size_t lds = (size_t)n_max * 8 * 4 + 68 * 4;
  if (lds > 160 * 1024) return NAWS_ERR_UNSUPPORTED;            // n_max <= 5111
  if (naws_allow_lds(soft_nms_kernel) != NAWS_OK) return NAWS_ERR_LAUNCH;
  hipLaunchKernelGGL(soft_nms_kernel, dim3(batch), dim3(SN_T), lds, (hipStream_t)stream, dets, counts,
                     n_max, sigma, overlap_thresh, score_thresh, method, out_dets, keep, out_counts);
  return naws_check_launch();
}



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

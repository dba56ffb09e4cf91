// RoIPoolF (+ fused RoIFeatureBoost), RoIFeatureBoost fwd/bwd, RoIIoU for gfx950.
//
// ref: detectron/ops/roi_loop_pool_op.cu:31-101 (RoIPoolF arithmetic; :72 original
//      empty-bin rule), detectron/ops/roi_feature_boost_op.cc:8-66,
//      detectron/ops/roi_iou_op.cu:27-62.
//
// HBM-bound byte work: the NHWC kernel maps one lane to one channel so every
// window pixel is one contiguous read per workgroup, stages the
// [channels x bins] tile in LDS and writes the (c, ph, pw)-ordered output row
// as one contiguous stream.
#include <float.h>
#include <stdlib.h>
#include "naws_common.h"

namespace {

struct RoiBins {
  int batch, start_w, start_h, roi_w, roi_h;
};

// Integer RoI frame: must stay bit-exact with the reference (roundf on the
// fp32 product, max(.,1) on the extents).
__device__ __forceinline__ RoiBins roi_frame(const float* roi, float spatial_scale) {
  RoiBins b;
  b.batch = (int)roi[0];
  int start_w = (int)roundf(roi[1] * spatial_scale);
  int start_h = (int)roundf(roi[2] * spatial_scale);
  int end_w = (int)roundf(roi[3] * spatial_scale);
  int end_h = (int)roundf(roi[4] * spatial_scale);
  b.start_w = start_w;
  b.start_h = start_h;
  b.roi_w = max(end_w - start_w + 1, 1);
  b.roi_h = max(end_h - start_h + 1, 1);
  return b;
}

__device__ __forceinline__ void bin_range(int p, float bin_size, int roi_start, int limit,
                                          int& lo, int& hi) {
  int s = (int)floorf((float)p * bin_size);
  int e = (int)ceilf((float)(p + 1) * bin_size);
  lo = min(max(s + roi_start, 0), limit);
  hi = min(max(e + roi_start, 0), limit);
}

// ---- NHWC: one workgroup = one RoI x CT channels -------------------------
constexpr int CT = 256;

template <bool WITH_ARGMAX>
__global__ __launch_bounds__(CT) void roi_pool_nhwc_kernel(
    const float* __restrict__ X, int C, int H, int W, const float* __restrict__ rois,
    const float* __restrict__ boost, int PH, int PW, float spatial_scale,
    float* __restrict__ Y, int32_t* __restrict__ argmax) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int nb = PH * PW;
  float* tile = reinterpret_cast<float*>(smem_raw);
  int32_t* itile = reinterpret_cast<int32_t*>(smem_raw) + CT * nb;

  const int r = blockIdx.x;
  const int c0 = blockIdx.y * CT;
  const int c = c0 + threadIdx.x;
  const bool active = c < C;

  const RoiBins rb = roi_frame(rois + (int64_t)r * 5, spatial_scale);
  const float bin_h = (float)rb.roi_h / (float)PH;
  const float bin_w = (float)rb.roi_w / (float)PW;
  const float scale = boost ? boost[r] : 1.0f;
  const float* Xn = X + (int64_t)rb.batch * H * W * C + (active ? c : 0);

  for (int ph = 0; ph < PH; ++ph) {
    int hs, he;
    bin_range(ph, bin_h, rb.start_h, H, hs, he);
    for (int pw = 0; pw < PW; ++pw) {
      int ws, we;
      bin_range(pw, bin_w, rb.start_w, W, ws, we);
      const bool empty = (he <= hs) || (we <= ws);
      float best = empty ? 0.0f : -FLT_MAX;
      int besti = -1;
      if (active) {
        for (int h = hs; h < he; ++h) {
          const float* row = Xn + (int64_t)h * W * C;
          int w = ws;
          for (; w + 4 <= we; w += 4) {  // 4 independent loads in flight
            float v0 = row[(int64_t)(w + 0) * C];
            float v1 = row[(int64_t)(w + 1) * C];
            float v2 = row[(int64_t)(w + 2) * C];
            float v3 = row[(int64_t)(w + 3) * C];
            if (v0 > best) { best = v0; besti = h * W + w; }
            if (v1 > best) { best = v1; besti = h * W + w + 1; }
            if (v2 > best) { best = v2; besti = h * W + w + 2; }
            if (v3 > best) { best = v3; besti = h * W + w + 3; }
          }
          for (; w < we; ++w) {
            float v = row[(int64_t)w * C];
            if (v > best) { best = v; besti = h * W + w; }
          }
        }
      }
      const int bin = ph * PW + pw;
      tile[threadIdx.x * nb + bin] = best * scale;
      if (WITH_ARGMAX) itile[threadIdx.x * nb + bin] = besti;
    }
  }
  __syncthreads();
  const int nch = min(CT, C - c0);
  const int count = nch * nb;
  const int64_t base = ((int64_t)r * C + c0) * nb;
  for (int i = threadIdx.x; i < count; i += CT) {
    Y[base + i] = tile[i];
    if (WITH_ARGMAX) argmax[base + i] = itile[i];
  }
}

// ---- NHWC fast path (no argmax, C % 4 == 0): one WAVE = one RoI x 256 channels,
// each lane owns 4 consecutive channels (16-byte loads: 1 KiB per pixel per wave,
// 4 pixels in flight).  Only one row of bins is staged in LDS (7 KB), so the
// kernel runs at full wave occupancy: this gather is bytes-in-flight bound.
__global__ __launch_bounds__(64) void roi_pool_nhwc_v4_kernel(
    const float* __restrict__ X, int C, int H, int W, const float* __restrict__ rois,
    const float* __restrict__ boost, int PH, int PW, float spatial_scale,
    float* __restrict__ Y) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* tile = reinterpret_cast<float*>(smem_raw);  // [4][64][PW]
  const int r = blockIdx.x;
  const int c0 = blockIdx.y * 256;
  const int lane = threadIdx.x;
  const int c = c0 + lane * 4;
  const bool active = c < C;
  const RoiBins rb = roi_frame(rois + (int64_t)r * 5, spatial_scale);
  const float bin_h = (float)rb.roi_h / (float)PH;
  const float bin_w = (float)rb.roi_w / (float)PW;
  const float scale = boost ? boost[r] : 1.0f;
  const float* Xn = X + (int64_t)rb.batch * H * W * C + (active ? c : 0);
  const int nb = PH * PW;
  const int nch = min(256, C - c0);
  const int64_t obase = ((int64_t)r * C + c0) * nb;

  for (int ph = 0; ph < PH; ++ph) {
    int hs, he;
    bin_range(ph, bin_h, rb.start_h, H, hs, he);
    for (int pw = 0; pw < PW; ++pw) {
      int ws, we;
      bin_range(pw, bin_w, rb.start_w, W, ws, we);
      const bool empty = (he <= hs) || (we <= ws);
      const float init = empty ? 0.0f : -FLT_MAX;
      float4 best = make_float4(init, init, init, init);
      if (active) {
        for (int h = hs; h < he; ++h) {
          const float* row = Xn + (int64_t)h * W * C;
          int w = ws;
          for (; w + 4 <= we; w += 4) {
            const float4 v0 = *reinterpret_cast<const float4*>(row + (int64_t)(w + 0) * C);
            const float4 v1 = *reinterpret_cast<const float4*>(row + (int64_t)(w + 1) * C);
            const float4 v2 = *reinterpret_cast<const float4*>(row + (int64_t)(w + 2) * C);
            const float4 v3 = *reinterpret_cast<const float4*>(row + (int64_t)(w + 3) * C);
#define NAWS_UPD(v) \
  best.x = (v.x > best.x) ? v.x : best.x; best.y = (v.y > best.y) ? v.y : best.y; \
  best.z = (v.z > best.z) ? v.z : best.z; best.w = (v.w > best.w) ? v.w : best.w;
            NAWS_UPD(v0) NAWS_UPD(v1) NAWS_UPD(v2) NAWS_UPD(v3)
          }
          for (; w < we; ++w) {
            const float4 v = *reinterpret_cast<const float4*>(row + (int64_t)w * C);
            NAWS_UPD(v)
          }
#undef NAWS_UPD
        }
      }
      tile[(0 * 64 + lane) * PW + pw] = best.x * scale;
      tile[(1 * 64 + lane) * PW + pw] = best.y * scale;
      tile[(2 * 64 + lane) * PW + pw] = best.z * scale;
      tile[(3 * 64 + lane) * PW + pw] = best.w * scale;
    }
    __syncthreads();
    const int count = nch * PW;
    for (int i = lane; i < count; i += 64) {
      const int cl = i / PW, pw = i - cl * PW;
      Y[obase + (int64_t)cl * nb + ph * PW + pw] = tile[((cl & 3) * 64 + (cl >> 2)) * PW + pw];
    }
    __syncthreads();
  }
}

// ---- NHWC, XCD-sliced (no argmax, C % 64 == 0): one wave = one RoI x 64 channels.
// blockIdx.x = slice + nslices * roi, so with the round-robin workgroup->XCD
// placement every XCD keeps gathering from the same 64-channel slice of the feature
// map (2.3 MB per 74x124 image: L2-resident) instead of all 8 L2s thrashing over the
// whole map.  Placement only affects speed.  Lane = (bin group g = lane>>4, channel
// quad cg = lane&15): four bins of a bin row are pooled concurrently, each 16-lane
// group reading 256 contiguous bytes per pixel.
//
// PLANES: instead of the fp32 feature rows the kernel writes the head GEMM's operand directly -
// the f16 hi / lo planes P[2][K/16][R][16] of Y[r][:] * s_r (K = C*PH*PW; csrc/gemm_x3.hip,
// naws_split_f16x2 layout) and 1/s_r.  The scale needs no pass over Y: max|Y[r]| <= max|X| of the
// roi's image (amax_words[batch], handed over by the conv body) * |boost[r]|.
struct RoiPlaneOut {
  unsigned short* P;           // planes
  float* inv_scale;            // [R]
  const unsigned* amax_words;  // per image: bit pattern of an upper bound of max|X[n]|
  int n_words;
  long long plane;             // elements between the hi and the lo plane (= K * R)
  int R;
  int bf16;                    // 1: ONE bf16 plane [K/16][R][16] of Y itself (the bf16 plan): no scale
};

// HIER: the window maxima are taken over precomputed block maxima instead of single pixels.
// M2[y][x] / M4[y][x] = max of X over the 2x2 / 4x4 block whose top-left pixel is (y, x)
// (roi_maxmaps_kernel, same NHWC shape as X).  max is idempotent, so a window [hs,he) x [ws,we) with
// both sides >= L is exactly the max of the L x L blocks at rows hs, hs+L, ... and he-L (the last
// one shifted back inside the window; likewise in x): ceil(dh/L) * ceil(dw/L) reads instead of
// dh * dw - the gather (8.8 GB from L2 for 4000 proposals on a 74 x 124 map) shrinks ~8x.  Same
// strict '>' chain from -FLT_MAX as the reference loop (NaNs never win); values bit-identical.
// A workgroup of NW waves pools RG consecutive rois of one slice.  The (bin row, bin group) steps of
// a roi - each a dependent load -> max -> LDS chain, 14 of them for 7x7 bins - are dealt round-robin
// to its NW / RG waves (the hierarchical form is latency-, not byte-bound: more chains in flight).
// RG > 1 (operand planes): the K-slab pieces of RG neighbouring rois are adjacent in the plane
// layout, so the write-out emits RG * 32-byte runs instead of single 32-byte pieces (measured: the
// scattered 32-byte pieces, not the gather, bound the hierarchical kernel).
template <bool PLANES, bool HIER = false, int NW = 1, int RG = 1>
__global__ __launch_bounds__(64 * NW) void roi_pool_nhwc_xcd_kernel(
    const float* __restrict__ X, int C, int H, int W, const float* __restrict__ rois, int R,
    const float* __restrict__ boost, int PH, int PW, float spatial_scale, int nslices,
    float* __restrict__ Y, RoiPlaneOut po, const float* __restrict__ M2 = nullptr,
    const float* __restrict__ M4 = nullptr) {
  static_assert(NW % RG == 0, "waves per roi");
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int nb = PH * PW;
  float* tiles = reinterpret_cast<float*>(smem_raw);   // [RG][64 channels][PH*PW]: the output order
  float* s_scale = tiles + RG * 64 * nb;                // [RG] operand scale of each roi (PLANES)
  const int slice = blockIdx.x % nslices;
  const int r0 = (blockIdx.x / nslices) * RG;
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int q = wave % RG;                              // this wave's roi within the group
  const int r = r0 + q;
  const int cg = lane & 15, g = lane >> 4;
  const int c0 = slice * 64;
  float* tile = tiles + q * 64 * nb;

  if (r < R) {
    const RoiBins rb = roi_frame(rois + (int64_t)r * 5, spatial_scale);
    const float bin_h = (float)rb.roi_h / (float)PH;
    const float bin_w = (float)rb.roi_w / (float)PW;
    const float scale = boost ? boost[r] : 1.0f;
    const int64_t img_off = (int64_t)rb.batch * H * W * C + c0 + cg * 4;
    const float* Xn = X + img_off;
    if constexpr (PLANES) {
      if (wave < RG && lane == 0 && !po.bf16) {
        const float bound =
            __uint_as_float(po.amax_words[min(rb.batch, po.n_words - 1)]) * fabsf(scale);
        float sc, isc;
        naws_f16x2_scales(__float_as_uint(bound), sc, isc);
        s_scale[q] = sc;
        if (slice == 0) po.inv_scale[r] = isc;
      }
    }

    const int pwg = (PW + 3) / 4;                       // bin groups per bin row
    for (int step = wave / RG; step < PH * pwg; step += NW / RG) {
      const int ph = step / pwg, pw0 = (step - ph * pwg) * 4;
      int hs, he;
      bin_range(ph, bin_h, rb.start_h, H, hs, he);
      const int pw = pw0 + g;
      const bool lane_on = pw < PW;
      int ws = 0, we = 0;
      if (lane_on) bin_range(pw, bin_w, rb.start_w, W, ws, we);
      const bool empty = (he <= hs) || (we <= ws);
      const float init = empty ? 0.0f : -FLT_MAX;
      float4 best = make_float4(init, init, init, init);
#define NAWS_UPD(v) \
  best.x = (v.x > best.x) ? v.x : best.x; best.y = (v.y > best.y) ? v.y : best.y; \
  best.z = (v.z > best.z) ? v.z : best.z; best.w = (v.w > best.w) ? v.w : best.w;
      int hbeg = hs;
      if constexpr (HIER) {
        const int dmin = min(he - hs, we - ws);
        if (dmin >= 2) {
          const int Lb = dmin >= 4 ? 4 : 2;
          const float* S = (dmin >= 4 ? M4 : M2) + img_off;
          const int hl = he - Lb, wl = we - Lb;          // last block position inside the window
          for (int h = hs; h < he; h += 2 * Lb) {
            const float* q0 = S + (int64_t)min(h, hl) * W * C;
            const float* q1 = S + (int64_t)min(h + Lb, hl) * W * C;   // (repeats q0's row at the end)
            int w = ws;
            for (; w + 2 * Lb < we; w += 2 * Lb) {
              const float4 a0 = *reinterpret_cast<const float4*>(q0 + (int64_t)w * C);
              const float4 a1 = *reinterpret_cast<const float4*>(q0 + (int64_t)(w + Lb) * C);
              const float4 b0 = *reinterpret_cast<const float4*>(q1 + (int64_t)w * C);
              const float4 b1 = *reinterpret_cast<const float4*>(q1 + (int64_t)(w + Lb) * C);
              NAWS_UPD(a0) NAWS_UPD(a1) NAWS_UPD(b0) NAWS_UPD(b1)
            }
            {
              const int w0 = min(w, wl), w1 = min(w + Lb, wl);
              const float4 a0 = *reinterpret_cast<const float4*>(q0 + (int64_t)w0 * C);
              const float4 a1 = *reinterpret_cast<const float4*>(q0 + (int64_t)w1 * C);
              const float4 b0 = *reinterpret_cast<const float4*>(q1 + (int64_t)w0 * C);
              const float4 b1 = *reinterpret_cast<const float4*>(q1 + (int64_t)w1 * C);
              NAWS_UPD(a0) NAWS_UPD(a1) NAWS_UPD(b0) NAWS_UPD(b1)
            }
          }
          hbeg = he;                                     // window done: skip the pixel loops
        }
      }
      // max is order-independent: window rows are taken in pairs so that up to eight
      // independent 16-byte loads are in flight per step
      int h = hbeg;
      for (; h + 2 <= he; h += 2) {
        const float* q0 = Xn + (int64_t)h * W * C;
        const float* q1 = q0 + (int64_t)W * C;
        int w = ws;
        for (; w + 4 <= we; w += 4) {
          const float4 a0 = *reinterpret_cast<const float4*>(q0 + (int64_t)(w + 0) * C);
          const float4 a1 = *reinterpret_cast<const float4*>(q0 + (int64_t)(w + 1) * C);
          const float4 a2 = *reinterpret_cast<const float4*>(q0 + (int64_t)(w + 2) * C);
          const float4 a3 = *reinterpret_cast<const float4*>(q0 + (int64_t)(w + 3) * C);
          const float4 b0 = *reinterpret_cast<const float4*>(q1 + (int64_t)(w + 0) * C);
          const float4 b1 = *reinterpret_cast<const float4*>(q1 + (int64_t)(w + 1) * C);
          const float4 b2 = *reinterpret_cast<const float4*>(q1 + (int64_t)(w + 2) * C);
          const float4 b3 = *reinterpret_cast<const float4*>(q1 + (int64_t)(w + 3) * C);
          NAWS_UPD(a0) NAWS_UPD(a1) NAWS_UPD(a2) NAWS_UPD(a3)
          NAWS_UPD(b0) NAWS_UPD(b1) NAWS_UPD(b2) NAWS_UPD(b3)
        }
        for (; w < we; ++w) {
          const float4 a = *reinterpret_cast<const float4*>(q0 + (int64_t)w * C);
          const float4 b = *reinterpret_cast<const float4*>(q1 + (int64_t)w * C);
          NAWS_UPD(a) NAWS_UPD(b)
        }
      }
      for (; h < he; ++h) {
        const float* row = Xn + (int64_t)h * W * C;
        int w = ws;
        for (; w + 4 <= we; w += 4) {
          const float4 v0 = *reinterpret_cast<const float4*>(row + (int64_t)(w + 0) * C);
          const float4 v1 = *reinterpret_cast<const float4*>(row + (int64_t)(w + 1) * C);
          const float4 v2 = *reinterpret_cast<const float4*>(row + (int64_t)(w + 2) * C);
          const float4 v3 = *reinterpret_cast<const float4*>(row + (int64_t)(w + 3) * C);
          NAWS_UPD(v0) NAWS_UPD(v1) NAWS_UPD(v2) NAWS_UPD(v3)
        }
        for (; w < we; ++w) {
          const float4 v = *reinterpret_cast<const float4*>(row + (int64_t)w * C);
          NAWS_UPD(v)
        }
      }
#undef NAWS_UPD
      if (lane_on) {
        const int bin = ph * PW + pw;
        tile[(cg * 4 + 0) * nb + bin] = best.x * scale;
        tile[(cg * 4 + 1) * nb + bin] = best.y * scale;
        tile[(cg * 4 + 2) * nb + bin] = best.z * scale;
        tile[(cg * 4 + 3) * nb + bin] = best.w * scale;
      }
    }
  }
  __syncthreads();
  const int count = 64 * nb;
  if constexpr (!PLANES) {
    // each (roi, 64-channel slice) block of the output is one contiguous run of 64*PH*PW floats:
    // written once, fully coalesced, from the LDS tile
    for (int i = threadIdx.x; i < RG * count; i += 64 * NW) {
      const int qq = i / count, k = i - qq * count;
      if (r0 + qq < R) Y[((int64_t)(r0 + qq) * C + c0) * nb + k] = tiles[i];
    }
  } else {
    // the same runs are count/16 K-slabs of the rois' operand rows: per slab, the RG rois' 32-byte
    // pieces are adjacent (neighbouring groups' pieces too, written by the same XCD)
    const int slab0 = slice * (count / 16);
    for (int i = threadIdx.x; i < RG * (count / 8); i += 64 * NW) {
      const int hh = i & 1, qq = (i >> 1) % RG, j = i / (2 * RG);
      if (r0 + qq >= R) continue;
      const float* src = tiles + qq * count + j * 16 + hh * 8;
      if (po.bf16) {                                   // (uniform over the launch)
        unsigned short q8[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const __bf16 b = (__bf16)src[e];
          q8[e] = *reinterpret_cast<const unsigned short*>(&b);
        }
        uint4 w;
        w.x = q8[0] | ((unsigned)q8[1] << 16); w.y = q8[2] | ((unsigned)q8[3] << 16);
        w.z = q8[4] | ((unsigned)q8[5] << 16); w.w = q8[6] | ((unsigned)q8[7] << 16);
        *reinterpret_cast<uint4*>(po.P + ((long long)(slab0 + j) * po.R + r0 + qq) * 16 + hh * 8) = w;
        continue;
      }
      const float sc = s_scale[qq];
      unsigned short hi[8], lo[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float t = src[e] * sc;
        const _Float16 a = (_Float16)t;
        float rr = t - (float)a;
        if (!(fabsf(t) <= 65504.f)) rr = 0.f;
        const _Float16 b = (_Float16)rr;
        hi[e] = *reinterpret_cast<const unsigned short*>(&a);
        lo[e] = *reinterpret_cast<const unsigned short*>(&b);
      }
      const long long dst = ((long long)(slab0 + j) * po.R + r0 + qq) * 16 + hh * 8;
      uint4 wh, wl;
      wh.x = hi[0] | ((unsigned)hi[1] << 16); wh.y = hi[2] | ((unsigned)hi[3] << 16);
      wh.z = hi[4] | ((unsigned)hi[5] << 16); wh.w = hi[6] | ((unsigned)hi[7] << 16);
      wl.x = lo[0] | ((unsigned)lo[1] << 16); wl.y = lo[2] | ((unsigned)lo[3] << 16);
      wl.z = lo[4] | ((unsigned)lo[5] << 16); wl.w = lo[6] | ((unsigned)lo[7] << 16);
      *reinterpret_cast<uint4*>(po.P + dst) = wh;
      *reinterpret_cast<uint4*>(po.P + po.plane + dst) = wl;
    }
  }
}

// f16 operand planes [2][K/16][R][16] (K-contiguous rows = rois) -> [2][Rpad/16][K][16]
// (K-contiguous rows = features), the operand form of dW = dY^T X: a blocked 16 x 16 transposition
// of 2-byte elements; rois >= R are written as zero.  One workgroup = 16 rois x 256 features.
__global__ __launch_bounds__(256) void planes_transpose_kernel(const unsigned short* __restrict__ P,
                                                               int R, int K, long long plane_in,
                                                               long long plane_out,
                                                               unsigned short* __restrict__ Q) {
  __shared__ unsigned short t[16][16][18];         // [slab][roi][k % 16], padded
  const int k0 = blockIdx.x * 256, r0 = blockIdx.y * 16, pl = blockIdx.z;
  const unsigned short* src = P + pl * plane_in;
  unsigned short* dst = Q + pl * plane_out;
  for (int u = threadIdx.x; u < 512; u += 256) {   // 16 slabs x 16 rois x 2 halves of 16 bytes
    const int hh = u & 1, i = (u >> 1) & 15, sl = u >> 5;
    const int slab = k0 / 16 + sl;
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if (slab * 16 < K && r0 + i < R)
      v = *reinterpret_cast<const uint4*>(src + ((long long)slab * R + r0 + i) * 16 + hh * 8);
    unsigned short* d = &t[sl][i][hh * 8];
    d[0] = v.x & 0xffff; d[1] = v.x >> 16; d[2] = v.y & 0xffff; d[3] = v.y >> 16;
    d[4] = v.z & 0xffff; d[5] = v.z >> 16; d[6] = v.w & 0xffff; d[7] = v.w >> 16;
  }
  __syncthreads();
  const int k = k0 + threadIdx.x;
  if (k >= K) return;
  const int sl = threadIdx.x >> 4, kk = threadIdx.x & 15;
  unsigned w[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) w[i] = t[sl][2 * i][kk] | ((unsigned)t[sl][2 * i + 1][kk] << 16);
  unsigned short* o = dst + ((long long)blockIdx.y * K + k) * 16;
  *reinterpret_cast<uint4*>(o) = make_uint4(w[0], w[1], w[2], w[3]);
  *reinterpret_cast<uint4*>(o + 8) = make_uint4(w[4], w[5], w[6], w[7]);
}

// Block-maxima maps of an NHWC tensor for the hierarchical pooling above: M2 / M4 [N][H][W][C],
// M_L[y][x] = max of X over [y, min(y+L, H)) x [x, min(x+L, W)) (blocks reaching over the border are
// never read by the pooling kernel).  Two passes of four reads each (round 3; one pass of sixteen
// took 0.06 ms for two 74 x 124 x 512 maps): M2 from X, then M4[y][x] = max(M2[y][x], M2[y][x+2],
// M2[y+2][x], M2[y+2][x+2]) - the same strict '>' chain from -FLT_MAX, so a NaN never wins in
// either pass and M4 equals the sixteen-way maximum bit for bit (indices clamped at the border,
// where the value is never used).  One thread = one pixel x 4 channels.
template <int STEP>
__global__ __launch_bounds__(256) void roi_maxmap_kernel(const float* __restrict__ S, int H, int W,
                                                         int C, float* __restrict__ D,
                                                         long long total4) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total4) return;
  const int c4 = C / 4;
  const int cq = (int)(i % c4);
  const long long pix = i / c4;
  const int x = (int)(pix % W), y = (int)((pix / W) % H);
  const int dy = (y + STEP < H) ? STEP : 0, dx = (x + STEP < W) ? STEP : 0;
  const float* base = S + (pix * C + cq * 4);
  const float4 v00 = *reinterpret_cast<const float4*>(base);
  const float4 v01 = *reinterpret_cast<const float4*>(base + (long long)dx * C);
  const float4 v10 = *reinterpret_cast<const float4*>(base + (long long)dy * W * C);
  const float4 v11 = *reinterpret_cast<const float4*>(base + ((long long)dy * W + dx) * C);
  float4 m = make_float4(-FLT_MAX, -FLT_MAX, -FLT_MAX, -FLT_MAX);
#define NAWS_MX(b, v) \
  b.x = (v.x > b.x) ? v.x : b.x; b.y = (v.y > b.y) ? v.y : b.y; \
  b.z = (v.z > b.z) ? v.z : b.z; b.w = (v.w > b.w) ? v.w : b.w;
  NAWS_MX(m, v00) NAWS_MX(m, v01) NAWS_MX(m, v10) NAWS_MX(m, v11)
#undef NAWS_MX
  *reinterpret_cast<float4*>(D + (pix * C + cq * 4)) = m;
}

// ---- NCHW: one lane = one output element (op-level API on reference layout)
__global__ void roi_pool_nchw_kernel(int64_t total, const float* __restrict__ X, int C, int H,
                                     int W, const float* __restrict__ rois,
                                     const float* __restrict__ boost, int PH, int PW,
                                     float spatial_scale, float* __restrict__ Y,
                                     int32_t* __restrict__ argmax) {
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    int pw = (int)(idx % PW);
    int ph = (int)((idx / PW) % PH);
    int c = (int)((idx / PW / PH) % C);
    int r = (int)(idx / PW / PH / C);
    const RoiBins rb = roi_frame(rois + (int64_t)r * 5, spatial_scale);
    const float bin_h = (float)rb.roi_h / (float)PH;
    const float bin_w = (float)rb.roi_w / (float)PW;
    int hs, he, ws, we;
    bin_range(ph, bin_h, rb.start_h, H, hs, he);
    bin_range(pw, bin_w, rb.start_w, W, ws, we);
    const bool empty = (he <= hs) || (we <= ws);
    float best = empty ? 0.0f : -FLT_MAX;
    int besti = -1;
    const float* Xc = X + ((int64_t)rb.batch * C + c) * H * W;
    for (int h = hs; h < he; ++h)
      for (int w = ws; w < we; ++w) {
        float v = Xc[h * W + w];
        if (v > best) { best = v; besti = h * W + w; }
      }
    Y[idx] = best * (boost ? boost[r] : 1.0f);
    if (argmax) argmax[idx] = besti;
  }
}

__global__ void row_scale_kernel(const float* __restrict__ X, const float* __restrict__ S,
                                 int64_t total, int F, float* __restrict__ Y) {
  // 16-byte lanes when F % 4 == 0 (rows then never straddle a float4)
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    Y[i] = X[i] * S[i / F];
  }
}
__global__ void row_scale4_kernel(const float4* __restrict__ X, const float* __restrict__ S,
                                  int64_t total4, int F4, float4* __restrict__ Y) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4;
       i += (int64_t)gridDim.x * blockDim.x) {
    float s = S[i / F4];
    float4 v = X[i];
    v.x *= s; v.y *= s; v.z *= s; v.w *= s;
    Y[i] = v;
  }
}

// ---- RoIIoU --------------------------------------------------------------
__global__ void roi_iou_kernel(const float* __restrict__ rois, int n, float* __restrict__ J) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;  // column
  const int j = blockIdx.y;                              // row
  if (i >= n) return;
  const int64_t idx = (int64_t)j * n + i;
  if (i == j) { J[idx] = 1.0f; return; }
  int ixmin = (int)rois[i * 5 + 1], iymin = (int)rois[i * 5 + 2];
  int ixmax = (int)rois[i * 5 + 3], iymax = (int)rois[i * 5 + 4];
  int jxmin = (int)rois[j * 5 + 1], jymin = (int)rois[j * 5 + 2];
  int jxmax = (int)rois[j * 5 + 3], jymax = (int)rois[j * 5 + 4];
  int xmin = max(ixmin, jxmin), ymin = max(iymin, jymin);
  int xmax = min(ixmax, jxmax), ymax = min(iymax, jymax);
  int w = (int)fmax(xmax - xmin + 1., 0.);
  int h = (int)fmax(ymax - ymin + 1., 0.);
  float inters = (float)(w * h);
  float uni = (float)((ixmax - ixmin + 1.) * (iymax - iymin + 1.) +
                      (jxmax - jxmin + 1.) * (jymax - jymin + 1.) - (double)inters);
  J[idx] = inters / uni;
}

}  // namespace

extern "C" int naws_roi_pool_f_fwd(const float* X, int layout, int N, int C, int H, int W,
                                   const float* rois, int R, const float* boost, int pooled_h,
                                   int pooled_w, float spatial_scale, float* Y,
                                   int32_t* argmax, void* stream) {
  if (R < 0 || N <= 0 || C <= 0 || H <= 0 || W <= 0 || pooled_h <= 0 || pooled_w <= 0)
    return NAWS_ERR_SHAPE;
  if (layout != NAWS_LAYOUT_NCHW && layout != NAWS_LAYOUT_NHWC) return NAWS_ERR_ARG;
  if (R == 0) return NAWS_OK;
  NAWS_REQUIRE_PTR(X);
  NAWS_REQUIRE_PTR(rois);
  NAWS_REQUIRE_PTR(Y);
  hipStream_t s = (hipStream_t)stream;
  const int nb = pooled_h * pooled_w;
  if (layout == NAWS_LAYOUT_NHWC) {
    size_t lds = (size_t)CT * nb * sizeof(float) * (argmax ? 2 : 1);
    if (lds > 160 * 1024) return NAWS_ERR_UNSUPPORTED;
    dim3 grid(R, (unsigned)naws_cdiv(C, CT));
    if (argmax)
      hipLaunchKernelGGL(roi_pool_nhwc_kernel<true>, grid, dim3(CT), lds, s, X, C, H, W, rois,
                         boost, pooled_h, pooled_w, spatial_scale, Y, argmax);
    else if (C % 64 == 0 && ((uintptr_t)X % 16) == 0 && pooled_h * pooled_w <= 256 &&
             (int64_t)R * (C / 64) < 0x7fffffffLL)
      hipLaunchKernelGGL(roi_pool_nhwc_xcd_kernel<false>, dim3((unsigned)(R * (C / 64))), dim3(64),
                         (size_t)(64 * pooled_h * pooled_w + 1) * sizeof(float), s, X, C, H, W, rois,
                         R, boost, pooled_h, pooled_w, spatial_scale, C / 64, Y, RoiPlaneOut{});
    else if (C % 4 == 0 && ((uintptr_t)X % 16) == 0 && pooled_w <= 64)
      hipLaunchKernelGGL(roi_pool_nhwc_v4_kernel, dim3(R, (unsigned)naws_cdiv(C, 256)), dim3(64),
                         (size_t)4 * 64 * pooled_w * sizeof(float), s, X, C, H, W, rois, boost,
                         pooled_h, pooled_w, spatial_scale, Y);
    else
      hipLaunchKernelGGL(roi_pool_nhwc_kernel<false>, grid, dim3(CT), lds, s, X, C, H, W, rois,
                         boost, pooled_h, pooled_w, spatial_scale, Y, argmax);
  } else {
    int64_t total = (int64_t)R * C * nb;
    int blocks = (int)std::min<int64_t>(naws_cdiv(total, 256), 256 * 32);
    hipLaunchKernelGGL(roi_pool_nchw_kernel, dim3(blocks), dim3(256), 0, s, total, X, C, H, W,
                       rois, boost, pooled_h, pooled_w, spatial_scale, Y, argmax);
  }
  return naws_check_launch();
}

// RoIPoolF + boost straight into the fp16x2 operand of the head GEMM (see RoiPlaneOut above).
// planes: f16 [2][K/16][R][16], K = C * pooled_h * pooled_w; scales: fp32 [2][R] ([1] = 1/s_r, [0]
// unused) - the pair naws_gemm_f32_f16x2_nt takes as (A2, scaleA).
extern "C" int naws_roi_pool_f_f16x2_fwd(const float* X, int N, int C, int H, int W,
                                         const float* rois, int R, const float* boost,
                                         int pooled_h, int pooled_w, float spatial_scale,
                                         const uint32_t* amax_words, int n_words, void* planes,
                                         float* scales, void* stream) {
  if (R <= 0 || N <= 0 || C <= 0 || H <= 0 || W <= 0 || pooled_h <= 0 || pooled_w <= 0 || n_words <= 0)
    return NAWS_ERR_SHAPE;
  NAWS_REQUIRE_PTR(X); NAWS_REQUIRE_PTR(rois); NAWS_REQUIRE_PTR(amax_words);
  NAWS_REQUIRE_PTR(planes); NAWS_REQUIRE_PTR(scales);
  const long long K = (long long)C * pooled_h * pooled_w;
  if (C % 64 != 0 || pooled_h * pooled_w > 256 || K % 32 != 0 || (int64_t)R * (C / 64) >= 0x7fffffffLL)
    return NAWS_ERR_UNSUPPORTED;
  if ((((uintptr_t)X | (uintptr_t)planes) & 15) != 0) return NAWS_ERR_ARG;
  RoiPlaneOut po;
  po.P = (unsigned short*)planes; po.inv_scale = scales + R; po.amax_words = (const unsigned*)amax_words;
  po.n_words = n_words; po.plane = K * R; po.R = R; po.bf16 = 0;
  hipLaunchKernelGGL(roi_pool_nhwc_xcd_kernel<true>, dim3((unsigned)(R * (C / 64))), dim3(64),
                     (size_t)(64 * pooled_h * pooled_w + 1) * sizeof(float), (hipStream_t)stream, X, C,
                     H, W, rois, R, boost, pooled_h, pooled_w, spatial_scale, C / 64, (float*)nullptr,
                     po);
  return naws_check_launch();
}

// The hierarchical forms (see roi_pool_nhwc_xcd_kernel<., HIER>): `workspace` holds the two
// block-maxima maps, naws_roi_pool_workspace_floats(N, C, H, W) floats, built here on the same stream.
extern "C" int64_t naws_roi_pool_workspace_floats(int N, int C, int H, int W) {
  if (N <= 0 || C <= 0 || H <= 0 || W <= 0) return 0;
  return (int64_t)2 * N * H * W * C;
}

static int roi_maxmaps(const float* X, int N, int C, int H, int W, float* M2, float* M4,
                       hipStream_t s) {
  const long long total4 = (long long)N * H * W * (C / 4);
  if (naws_cdiv(total4, 256) > 0x7fffffffLL) return NAWS_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(roi_maxmap_kernel<1>, dim3((unsigned)naws_cdiv(total4, 256)), dim3(256), 0, s, X, H,
                     W, C, M2, total4);
  hipLaunchKernelGGL(roi_maxmap_kernel<2>, dim3((unsigned)naws_cdiv(total4, 256)), dim3(256), 0, s,
                     (const float*)M2, H, W, C, M4, total4);
  return naws_check_launch();
}

extern "C" int naws_roi_pool_f_nhwc_hier_fwd(const float* X, int N, int C, int H, int W,
                                             const float* rois, int R, const float* boost,
                                             int pooled_h, int pooled_w, float spatial_scale,
                                             float* workspace, float* Y, void* stream) {
  if (R < 0 || N <= 0 || C <= 0 || H <= 0 || W <= 0 || pooled_h <= 0 || pooled_w <= 0)
    return NAWS_ERR_SHAPE;
  if (R == 0) return NAWS_OK;
  NAWS_REQUIRE_PTR(X); NAWS_REQUIRE_PTR(rois); NAWS_REQUIRE_PTR(Y); NAWS_REQUIRE_PTR(workspace);
  if (C % 64 != 0 || pooled_h * pooled_w > 256 || (int64_t)R * (C / 64) >= 0x7fffffffLL)
    return NAWS_ERR_UNSUPPORTED;
  if ((((uintptr_t)X | (uintptr_t)workspace) & 15) != 0) return NAWS_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  const int rc = roi_maxmaps(X, N, C, H, W, workspace, workspace + (long long)N * H * W * C, s);
  if (rc != NAWS_OK) return rc;
  hipLaunchKernelGGL((roi_pool_nhwc_xcd_kernel<false, true, 4>), dim3((unsigned)(R * (C / 64))), dim3(256),
                     (size_t)(64 * pooled_h * pooled_w + 1) * sizeof(float), s, X, C, H, W, rois, R,
                     boost, pooled_h, pooled_w, spatial_scale, C / 64, Y, RoiPlaneOut{},
                     (const float*)workspace, (const float*)(workspace + (long long)N * H * W * C));
  return naws_check_launch();
}

// the pooling launch over block-maxima maps that already exist
static int roi_pool_planes_mapped(const float* X, int N, int C, int H, int W, const float* rois,
                                  int R, const float* boost, int pooled_h, int pooled_w,
                                  float spatial_scale, const uint32_t* amax_words, int n_words,
                                  const float* M2, const float* M4, void* planes, float* scales,
                                  hipStream_t s, int bf16 = 0, int R_total = 0, int r_first = 0) {
  const long long K = (long long)C * pooled_h * pooled_w;
  // (R_total > 0: this launch pools rois [r_first, r_first + R) of R_total - one image's
  // proposals on that image's stream; planes / scales / rois / boost describe all R_total rois)
  if (R_total <= 0) { R_total = R; r_first = 0; }
  rois += (long long)r_first * 5;
  if (boost) boost += r_first;
  RoiPlaneOut po;
  po.P = (unsigned short*)planes + (long long)r_first * 16;
  po.inv_scale = scales ? scales + R_total + r_first : nullptr;
  po.amax_words = (const unsigned*)amax_words;
  po.n_words = n_words; po.plane = K * R_total; po.R = R_total; po.bf16 = bf16;
  const int nw = naws_knob(NAWS_KNOB_ROI_NW);       // A/B knob (tools/bench_roi.py): NW * 10 + RG
  (void)nw;
#define NAWS_ROI_LAUNCH(NWV, RGV)                                                                    \
  hipLaunchKernelGGL((roi_pool_nhwc_xcd_kernel<true, true, NWV, RGV>),                               \
                     dim3((unsigned)(naws_cdiv(R, RGV) * (C / 64))), dim3(64 * NWV),                  \
                     (size_t)(RGV * 64 * pooled_h * pooled_w + RGV) * sizeof(float), s, X, C, H, W,  \
                     rois, R, boost, pooled_h, pooled_w, spatial_scale, C / 64, (float*)nullptr, po, \
                     M2, M4)
#ifdef NAWS_AB   // the other (waves, rois per workgroup) forms: A/B build only (tools/bench_roi.py)
  if (nw == 41) NAWS_ROI_LAUNCH(4, 1);
  else if (nw == 44) NAWS_ROI_LAUNCH(4, 4);
  else if (nw == 82) NAWS_ROI_LAUNCH(8, 2);
  else if (nw == 84) NAWS_ROI_LAUNCH(8, 4);
  else
#endif
  NAWS_ROI_LAUNCH(4, 2);     // measured best (tools/bench_roi.py): 0.35 ms incl. the maps vs 0.58 direct
#undef NAWS_ROI_LAUNCH
  return naws_check_launch();
}

static int roi_pool_planes_check(const float* X, int N, int C, int H, int W, const float* rois, int R,
                                 int pooled_h, int pooled_w, const uint32_t* amax_words, int n_words,
                                 const void* planes, const float* scales) {
  if (R <= 0 || N <= 0 || C <= 0 || H <= 0 || W <= 0 || pooled_h <= 0 || pooled_w <= 0 || n_words <= 0)
    return NAWS_ERR_SHAPE;
  NAWS_REQUIRE_PTR(X); NAWS_REQUIRE_PTR(rois); NAWS_REQUIRE_PTR(amax_words);
  NAWS_REQUIRE_PTR(planes); NAWS_REQUIRE_PTR(scales);
  const long long K = (long long)C * pooled_h * pooled_w;
  if (C % 64 != 0 || pooled_h * pooled_w > 256 || K % 32 != 0 || (int64_t)R * (C / 64) >= 0x7fffffffLL)
    return NAWS_ERR_UNSUPPORTED;
  if ((((uintptr_t)X | (uintptr_t)planes) & 15) != 0) return NAWS_ERR_ARG;
  return NAWS_OK;
}

extern "C" int naws_roi_pool_f_f16x2_hier_fwd(const float* X, int N, int C, int H, int W,
                                              const float* rois, int R, const float* boost,
                                              int pooled_h, int pooled_w, float spatial_scale,
                                              const uint32_t* amax_words, int n_words,
                                              float* workspace, void* planes, float* scales,
                                              void* stream) {
  int rc = roi_pool_planes_check(X, N, C, H, W, rois, R, pooled_h, pooled_w, amax_words, n_words,
                                 planes, scales);
  if (rc != NAWS_OK) return rc;
  NAWS_REQUIRE_PTR(workspace);
  if (((uintptr_t)workspace & 15) != 0) return NAWS_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  float* M4 = workspace + (long long)N * H * W * C;
  rc = roi_maxmaps(X, N, C, H, W, workspace, M4, s);
  if (rc != NAWS_OK) return rc;
  return roi_pool_planes_mapped(X, N, C, H, W, rois, R, boost, pooled_h, pooled_w, spatial_scale,
                                amax_words, n_words, workspace, M4, planes, scales, s);
}

// The two halves of naws_roi_pool_f_f16x2_hier_fwd as separate entries: the engine builds each
// image's maps at the tail of that image's conv chain (its own stream, beside the other image's
// last layers) and pools once the chains have joined.
extern "C" int naws_roi_maxmaps_fwd(const float* X, int N, int C, int H, int W, float* M2, float* M4,
                                    void* stream) {
  if (N <= 0 || C <= 0 || H <= 0 || W <= 0) return NAWS_ERR_SHAPE;
  NAWS_REQUIRE_PTR(X); NAWS_REQUIRE_PTR(M2); NAWS_REQUIRE_PTR(M4);
  if (C % 4 != 0) return NAWS_ERR_UNSUPPORTED;
  if ((((uintptr_t)X | (uintptr_t)M2 | (uintptr_t)M4) & 15) != 0 || M2 == M4) return NAWS_ERR_ARG;
  return roi_maxmaps(X, N, C, H, W, M2, M4, (hipStream_t)stream);
}

extern "C" int naws_roi_pool_f_f16x2_mapped_fwd(const float* X, int N, int C, int H, int W,
                                                const float* rois, int R, const float* boost,
                                                int pooled_h, int pooled_w, float spatial_scale,
                                                const uint32_t* amax_words, int n_words,
                                                const float* M2, const float* M4, void* planes,
                                                float* scales, void* stream) {
  const int rc = roi_pool_planes_check(X, N, C, H, W, rois, R, pooled_h, pooled_w, amax_words,
                                       n_words, planes, scales);
  if (rc != NAWS_OK) return rc;
  NAWS_REQUIRE_PTR(M2); NAWS_REQUIRE_PTR(M4);
  if ((((uintptr_t)M2 | (uintptr_t)M4) & 15) != 0) return NAWS_ERR_ARG;
  return roi_pool_planes_mapped(X, N, C, H, W, rois, R, boost, pooled_h, pooled_w, spatial_scale,
                                amax_words, n_words, M2, M4, planes, scales, (hipStream_t)stream);
}

// The same for rois [r_first, r_first + count) of R_total only (rois, boost, planes and scales are
// those of ALL R_total rois): the engine pools each image's proposals at the tail of that image's
// conv chain, on its stream, beside the other image's last layers.
extern "C" int naws_roi_pool_f_f16x2_mapped_range_fwd(const float* X, int N, int C, int H, int W,
                                                      const float* rois, int R_total, int r_first,
                                                      int count, const float* boost, int pooled_h,
                                                      int pooled_w, float spatial_scale,
                                                      const uint32_t* amax_words, int n_words,
                                                      const float* M2, const float* M4, void* planes,
                                                      float* scales, void* stream) {
  const int rc = roi_pool_planes_check(X, N, C, H, W, rois, R_total, pooled_h, pooled_w, amax_words,
                                       n_words, planes, scales);
  if (rc != NAWS_OK) return rc;
  if (r_first < 0 || count < 0 || r_first + count > R_total) return NAWS_ERR_SHAPE;
  if (count == 0) return NAWS_OK;
  NAWS_REQUIRE_PTR(M2); NAWS_REQUIRE_PTR(M4);
  if ((((uintptr_t)M2 | (uintptr_t)M4) & 15) != 0) return NAWS_ERR_ARG;
  return roi_pool_planes_mapped(X, N, C, H, W, rois, count, boost, pooled_h, pooled_w, spatial_scale,
                                amax_words, n_words, M2, M4, planes, scales, (hipStream_t)stream, 0,
                                R_total, r_first);
}

// RoIPoolF (+ boost) over existing block-maxima maps (naws_roi_maxmaps_fwd), written as the bf16
// plan's fc6 operand: ONE plane P[K/16][R][16] of Y rounded to bf16 (nearest-even), K = C * ph * pw
// (K % 64 == 0: the layout of naws_to_bf16_slab).  Same pooled values as every other form.
extern "C" int naws_roi_pool_f_bf16_slab_mapped_fwd(const float* X, int N, int C, int H, int W,
                                                    const float* rois, int R, const float* boost,
                                                    int pooled_h, int pooled_w, float spatial_scale,
                                                    const float* M2, const float* M4, void* P,
                                                    void* stream) {
  if (R <= 0 || N <= 0 || C <= 0 || H <= 0 || W <= 0 || pooled_h <= 0 || pooled_w <= 0)
    return NAWS_ERR_SHAPE;
  NAWS_REQUIRE_PTR(X); NAWS_REQUIRE_PTR(rois); NAWS_REQUIRE_PTR(P);
  NAWS_REQUIRE_PTR(M2); NAWS_REQUIRE_PTR(M4);
  const long long K = (long long)C * pooled_h * pooled_w;
  if (C % 64 != 0 || pooled_h * pooled_w > 256 || K % 64 != 0 || (int64_t)R * (C / 64) >= 0x7fffffffLL)
    return NAWS_ERR_UNSUPPORTED;
  if ((((uintptr_t)X | (uintptr_t)P | (uintptr_t)M2 | (uintptr_t)M4) & 15) != 0) return NAWS_ERR_ARG;
  return roi_pool_planes_mapped(X, N, C, H, W, rois, R, boost, pooled_h, pooled_w, spatial_scale,
                                nullptr, 1, M2, M4, P, nullptr, (hipStream_t)stream, 1);
}

// Q[Rpad/16][K][16] = transposition of the bf16 slab operand P[K/16][R][16] (rows >= R zero);
// Rpad = R rounded up to 64: the operand form of dW = dY^T X in the bf16 plan, from the forward
// operand instead of a second pass over fp32 features.
extern "C" int naws_bf16_slab_transpose(const void* P, int R, int K, int Rpad, void* Q, void* stream) {
  if (R <= 0 || K <= 0) return NAWS_ERR_SHAPE;
  if (K % 16 != 0 || Rpad != (R + 63) / 64 * 64) return NAWS_ERR_ARG;
  NAWS_REQUIRE_PTR(P); NAWS_REQUIRE_PTR(Q);
  if ((((uintptr_t)P | (uintptr_t)Q) & 15) != 0) return NAWS_ERR_ARG;
  dim3 grid((unsigned)naws_cdiv(K, 256), (unsigned)(Rpad / 16), 1);
  if (grid.y > 65535) return NAWS_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(planes_transpose_kernel, grid, dim3(256), 0, (hipStream_t)stream,
                     (const unsigned short*)P, R, K, (long long)K * R, (long long)Rpad * K,
                     (unsigned short*)Q);
  return naws_check_launch();
}

// Q[2][Rpad/16][K][16] = transposition of the f16 planes P[2][K/16][R][16]; Rpad = R rounded up to
// 32 (rows >= R zero).  With P = X * s_r this is the B operand of dW = dY^T X once dY's rows carry
// 1/s_r (naws_split_f16x2_kscaled); its own scale vector is all ones.
extern "C" int naws_f16_planes_transpose(const void* P, int R, int K, int Rpad, void* Q,
                                         void* stream) {
  if (R <= 0 || K <= 0) return NAWS_ERR_SHAPE;
  if (K % 16 != 0 || Rpad != (R + 31) / 32 * 32) return NAWS_ERR_ARG;
  NAWS_REQUIRE_PTR(P); NAWS_REQUIRE_PTR(Q);
  if ((((uintptr_t)P | (uintptr_t)Q) & 15) != 0) return NAWS_ERR_ARG;
  dim3 grid((unsigned)naws_cdiv(K, 256), (unsigned)(Rpad / 16), 2);
  if (grid.y > 65535) return NAWS_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(planes_transpose_kernel, grid, dim3(256), 0, (hipStream_t)stream,
                     (const unsigned short*)P, R, K, (long long)K * R, (long long)Rpad * K,
                     (unsigned short*)Q);
  return naws_check_launch();
}

static int row_scale(const float* X, const float* S, int R, int F, float* Y, void* stream) {
  if (R < 0 || F <= 0) return NAWS_ERR_SHAPE;
  if (R == 0) return NAWS_OK;
  NAWS_REQUIRE_PTR(X);
  NAWS_REQUIRE_PTR(S);
  NAWS_REQUIRE_PTR(Y);
  hipStream_t s = (hipStream_t)stream;
  int64_t total = (int64_t)R * F;
  bool vec = (F % 4 == 0) && (((uintptr_t)X | (uintptr_t)Y) % 16 == 0);
  if (vec) {
    int64_t t4 = total / 4;
    int blocks = (int)std::min<int64_t>(naws_cdiv(t4, 256), 2048);
    hipLaunchKernelGGL(row_scale4_kernel, dim3(blocks), dim3(256), 0, s, (const float4*)X, S, t4,
                       F / 4, (float4*)Y);
  } else {
    int blocks = (int)std::min<int64_t>(naws_cdiv(total, 256), 2048);
    hipLaunchKernelGGL(row_scale_kernel, dim3(blocks), dim3(256), 0, s, X, S, total, F, Y);
  }
  return naws_check_launch();
}

extern "C" int naws_roi_feature_boost_fwd(const float* X, const float* S, int R, int F, float* Y,
                                          void* stream) {
  return row_scale(X, S, R, F, Y, stream);
}
extern "C" int naws_roi_feature_boost_bwd(const float* dY, const float* S, int R, int F,
                                          float* dX, void* stream) {
  return row_scale(dY, S, R, F, dX, stream);
}

extern "C" int naws_roi_iou_fwd(const float* rois, int R, float* J, void* stream) {
  if (R < 0) return NAWS_ERR_SHAPE;
  if (R == 0) return NAWS_OK;
  NAWS_REQUIRE_PTR(rois);
  NAWS_REQUIRE_PTR(J);
  dim3 grid((unsigned)naws_cdiv(R, 256), R);
  hipLaunchKernelGGL(roi_iou_kernel, grid, dim3(256), 0, (hipStream_t)stream, rois, R, J);
  return naws_check_launch();
}

// Winograd F(4x4, 3x3) convolution for the VGG layers with 512 output channels (conv4_1 .. conv5_3),
// NHWC, in the 2 x f16 operand split.
//
// ref: detectron/modeling/VGG16.py:33-46 (conv4_x / conv5_x: 3x3, stride 1, pad == dilation).
// Y = A^T [ (G g G^T) (.) (B^T d B) ] A per 4x4 output tile (6x6 input tile, interpolation points
// 0, 1, -1, 2, -1/2, inf - see "Accuracy"): 36 multiplies for 16 outputs instead of 144 - 4x fewer
// MFMA flops than the direct sum, 1.78x fewer than F(2x2) (winograd.hip), and the transform-domain tensors V / M
// shrink from 16 x (H/2)(W/2) to 36 x (H/4)(W/4) rows: 0.5625 of F(2x2)'s bytes.  At 75 x 125 x 512
// that is 45 instead of 78 MB each way per layer-image, and 36 x 5 x 4 = 720 GEMM tiles of
// 128 x 128 - one resident round of the 256 CUs - instead of 1216.
//
// Accuracy.  The transforms run in fp32 (FMA), the 36 products per (tile, channel pair) on the f16
// matrix cores in the hi / lo split with fp32 accumulation (gemm_x3.hip, as F(2x2)).
// Interpolation points: the textbook set 0, +-1, +-2, inf was built first; the set in use pairs 2
// with -1/2 instead of -2 (Barabasz et al., "Error analysis and improving the accuracy of Winograd
// convolution for deep neural networks": reciprocal magnitudes keep the Vandermonde rows balanced).
// B^T stays small integers (exact products), A^T powers of two, G is taken in double.  Measured
// (tests/wino_error_study.py, CPU, fp32 emulation through conv4_1 .. conv5_3 at 600 x 1000, Kaiming
// / skewed statistics; per-channel error of conv5_3 / channel RMS against the fp32 direct oracle):
//     F(2x2) 2.1e-5 / 2.1e-5     F(4x4) 0, +-1, +-2: 3.8e-5 / 3.6e-5     F(4x4) 0, 1, -1, 2, -1/2: 2.6e-5 / 2.8e-5
// (the fp32 direct sum is itself 1.6e-5 / 1.9e-5 from float64); one layer on spatially white input
// against float64: 3.3e-6 of max|y| instead of 8.5e-6..1.1e-5.  |B^T d B| <= 196 max|x| (row sums of
// |B^T| are 14), so the tensor-wide power of two maps max|x| below 2^8 (F(2x2): 2^13) and the f16 pair
// keeps >= 22 significand bits down to 2^-19 of the maximum.  tests/test_gpu_fullsize_oracle.py holds
// the plan to 1e-4 on both statistics (measured on the GPU: 3.6e-5 / 3.8e-5; F(2x2) read 3.5e-5).
//
// Layouts.  V planes P[2][36][Cin/16][tiles][16] f16, M [36][tiles][Cout] fp32, U2 = naws_split_f16x2
// of U [36][Cout][Cin] (planes [2][36][Cin/16][Cout][16], scales [2][36][Cout]); frequency index
// xi = 6 i + j (row i, column j of the 6x6 transform-domain tile).  tiles = N * d*d *
// ceil(Hs/4) * ceil(Ws/4) over the (y%d, x%d) sub-grids of a dilation-d layer.
#include <stdlib.h>
#include "x3_common.h"

namespace {

struct Wino4Geom {
  int N, H, W, d, Hs, Ws, th, tw;
  long long P;
};

inline Wino4Geom wino4_geom(int N, int H, int W, int d) {
  Wino4Geom g;
  g.N = N; g.H = H; g.W = W; g.d = d;
  g.Hs = (H + d - 1) / d;
  g.Ws = (W + d - 1) / d;
  g.th = (g.Hs + 3) / 4;
  g.tw = (g.Ws + 3) / 4;
  g.P = (long long)N * d * d * g.th * g.tw;
  return g;
}

__device__ __forceinline__ void tile4_coords(const Wino4Geom& g, long long p, int& n, int& py,
                                             int& px, int& ty, int& tx) {
  tx = (int)(p % g.tw); p /= g.tw;
  ty = (int)(p % g.th); p /= g.th;
  px = (int)(p % g.d); p /= g.d;
  py = (int)(p % g.d); p /= g.d;
  n = (int)p;
}

__device__ __forceinline__ float4 f4fma(float a, float4 x, float4 y) {      // a * x + y
  return make_float4(fmaf(a, x.x, y.x), fmaf(a, x.y, y.y), fmaf(a, x.z, y.z), fmaf(a, x.w, y.w));
}
__device__ __forceinline__ float4 f4add4(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 f4sub4(float4 a, float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }

// B^T = [ 2  3 -4 -3  2  0 ]
//       [ 0  2  5  1 -2  0 ]
//       [ 0  2  1 -5  2  0 ]
//       [ 0 -1 -2  1  2  0 ]
//       [ 0 -2  1  2 -1  0 ]
//       [ 0  2  3 -4 -3  2 ]      o = B^T d over six float4 operands (every multiplier a small integer:
// the products are exact, each output carries two to four roundings)
__device__ __forceinline__ void bt6(const float4 (&d)[6], float4 (&o)[6]) {
  const float4 a = f4sub4(d[3], d[1]);               // d3 - d1
  const float4 b = f4sub4(d[4], d[2]);               // d4 - d2
  o[0] = f4fma(2.f, f4add4(d[0], d[4]), f4fma(-3.f, a, f4fma(-4.f, d[2], make_float4(0.f, 0.f, 0.f, 0.f))));
  o[1] = f4fma(2.f, f4sub4(d[1], d[4]), f4fma(5.f, d[2], d[3]));
  o[2] = f4fma(2.f, f4add4(d[1], d[4]), f4fma(-5.f, d[3], d[2]));
  o[3] = f4fma(2.f, b, a);
  o[4] = f4fma(2.f, a, f4sub4(d[2], d[4]));
  o[5] = f4fma(2.f, f4add4(d[1], d[5]), f4fma(-3.f, b, f4fma(-4.f, d[3], make_float4(0.f, 0.f, 0.f, 0.f))));
}

// A^T = [ 1  1  1  1   1    0 ]
//       [ 0  1 -1  2  -1/2  0 ]
//       [ 0  1  1  4   1/4  0 ]
//       [ 0  1 -1  8  -1/8  1 ]
__device__ __forceinline__ void at6(const float4 (&m)[6], float4 (&s)[4]) {
  const float4 a = f4add4(m[1], m[2]), b = f4sub4(m[1], m[2]);
  s[0] = f4add4(f4add4(m[0], a), f4add4(m[3], m[4]));
  s[1] = f4fma(2.f, m[3], f4fma(-0.5f, m[4], b));
  s[2] = f4fma(4.f, m[3], f4fma(0.25f, m[4], a));
  s[3] = f4add4(f4fma(8.f, m[3], f4fma(-0.125f, m[4], b)), m[5]);
}

// V planes = split of (B^T d B) * s, s = one power of two for the whole tensor: max|x| * s < 2^8,
// |B^T d B| * s <= 196 * 2^8 < 65504.  One lane = one tile x 4 channels x TWO of the six rows of
// the transform-domain tile; lane order (4 channel quads of a 16-channel slab, then tiles, then
// slabs): a wave writes 16 tiles x 32 B = 512 contiguous bytes per (xi, plane) and reads 64-byte
// pieces of 16 pixels per tap.  The three waves of a workgroup take the row pairs {1, 2}, {3, 4}
// and {0, 5} of the same 64 (tile, quad) items: rows 3 / 4 share their sub-expressions of B^T, the
// first two need input rows 1..4 only, the taps the waves have in common are L1 hits - and the
// launch has three times the waves of a one-lane-per-tile form (12 instead of 36 transform-domain
// values live per lane; the two forms time alike: 21.7 vs 21.0 us at 75 x 125 x 512, 75 vs 79 at
// 150 x 250).
template <int RP>
__device__ __forceinline__ void wino4_input_rows(const float* __restrict__ X, const Wino4Geom& g,
                                                 int Cin, int c, int n, int py, int px, int ty,
                                                 int tx, float sc, unsigned short* __restrict__ hi,
                                                 unsigned short* __restrict__ lo,
                                                 long long xi_stride) {
  constexpr int R0 = RP == 0 ? 1 : RP == 1 ? 3 : 0, R1 = RP == 0 ? 2 : RP == 1 ? 4 : 5;
  constexpr int I0 = RP == 2 ? 0 : 1, I1 = RP == 2 ? 6 : 5;       // input rows the pair needs
  float4 t0[6], t1[6];                  // (B^T d)[R0][.], (B^T d)[R1][.]
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    const int xs = 4 * tx - 1 + j;
    const int x = xs * g.d + px;
    float4 dd[6];
#pragma unroll
    for (int i = I0; i < I1; ++i) {
      const int ys = 4 * ty - 1 + i;
      const int y = ys * g.d + py;
      const bool ok = ys >= 0 && xs >= 0 && y < g.H && x < g.W;
      dd[i] = ok ? *reinterpret_cast<const float4*>(
                       X + (((long long)n * g.H + y) * g.W + x) * Cin + c)
                 : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    // (the same expressions as bt6's rows R0 / R1)
    if constexpr (RP == 0) {
      t0[j] = f4fma(2.f, f4sub4(dd[1], dd[4]), f4fma(5.f, dd[2], dd[3]));
      t1[j] = f4fma(2.f, f4add4(dd[1], dd[4]), f4fma(-5.f, dd[3], dd[2]));
    } else if constexpr (RP == 1) {
      const float4 a = f4sub4(dd[3], dd[1]), b = f4sub4(dd[4], dd[2]);
      t0[j] = f4fma(2.f, b, a);
      t1[j] = f4fma(2.f, a, f4sub4(dd[2], dd[4]));
    } else {
      const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
      t0[j] = f4fma(2.f, f4add4(dd[0], dd[4]), f4fma(-3.f, f4sub4(dd[3], dd[1]), f4fma(-4.f, dd[2], z)));
      t1[j] = f4fma(2.f, f4add4(dd[1], dd[5]), f4fma(-3.f, f4sub4(dd[4], dd[2]), f4fma(-4.f, dd[3], z)));
    }
  }
  float4 oo[6];
  bt6(t0, oo);                          // (.) B
#pragma unroll
  for (int j = 0; j < 6; ++j)
    put_h2(oo[j], sc, hi + (R0 * 6 + j) * xi_stride, lo + (R0 * 6 + j) * xi_stride);
  bt6(t1, oo);
#pragma unroll
  for (int j = 0; j < 6; ++j)
    put_h2(oo[j], sc, hi + (R1 * 6 + j) * xi_stride, lo + (R1 * 6 + j) * xi_stride);
}

// SLABS = 2: six waves per workgroup - the three row pairs of TWO neighbouring 16-channel slabs of
// the same 16 tiles, so that both 64-byte halves of every 128-byte line of X are read on one CU.
template <int SLABS>
__global__ __launch_bounds__(192 * SLABS) void wino4_input_h2_kernel(const float* __restrict__ X, Wino4Geom g,
                                                                     int Cin, const unsigned* __restrict__ amax,
                                                                     float* __restrict__ inv_scale,
                                                                     unsigned short* __restrict__ Vp,
                                                                     unsigned* __restrict__ amax_out) {
  // this layer's output maximum is accumulated by wino4_output_kernel, stream-ordered after this
  if (amax_out && blockIdx.x == 0 && threadIdx.x == 0) *amax_out = 0u;
  int e = (int)((*amax >> 23) & 0xff);
  if (*amax == 0 || e == 0xff) e = 127 + 7;
  e = min(max(e, 40), 250);
  const float sc = __uint_as_float((unsigned)(261 - e) << 23);      // 2^(7 - (e - 127))
  const float isc = __uint_as_float((unsigned)(e - 7) << 23);
  const long long total = g.P * (Cin / 4) / SLABS;                  // (tile, quad, slab group) items
  const long long xi_stride = (long long)Cin * g.P;                 // elements between the 36 xi
  const long long plane = 36 * xi_stride;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int rp = wv % 3, ss = wv / 3;
  for (long long t = (long long)blockIdx.x * 64 + (threadIdx.x & 63); t < total;
       t += (long long)gridDim.x * 64) {
    const int cq = (int)(t & 3);
    const long long p = (t >> 2) % g.P;
    const int s = (int)((t >> 2) / g.P) * SLABS + ss;
    const int c = s * 16 + cq * 4;
    if (s == 0 && cq == 0 && rp == 0) inv_scale[p] = isc;
    int n, py, px, ty, tx;
    tile4_coords(g, p, n, py, px, ty, tx);
    unsigned short* hi = Vp + ((long long)s * g.P + p) * 16 + cq * 4;
    unsigned short* lo = hi + plane;
    if (rp == 0) wino4_input_rows<0>(X, g, Cin, c, n, py, px, ty, tx, sc, hi, lo, xi_stride);
    else if (rp == 1) wino4_input_rows<1>(X, g, Cin, c, n, py, px, ty, tx, sc, hi, lo, xi_stride);
    else wino4_input_rows<2>(X, g, Cin, c, n, py, px, ty, tx, sc, hi, lo, xi_stride);
  }
}

// Y tile = A^T m A (+ bias, ReLU);  one lane = one tile x 4 output channels, channel quads fastest:
// a wave reads 1 KB runs of each of the 36 M planes and writes 1 KB runs of 16 pixels.
__global__ __launch_bounds__(256) void wino4_output_kernel(const float* __restrict__ M, Wino4Geom g,
                                                           int Cout, long long slab,
                                                           const float* __restrict__ bias,
                                                           int relu, float* __restrict__ Y,
                                                           unsigned* __restrict__ amax_out) {
  const int c4n = Cout / 4;
  const long long total = g.P * c4n;
  float vmax = 0.f;
  for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < total;
       t += (long long)gridDim.x * 256) {
    const int c4 = (int)(t % c4n);
    const long long p = t / c4n;
    int n, py, px, ty, tx;
    tile4_coords(g, p, n, py, px, ty, tx);
    const float* in = M + p * Cout + c4 * 4;
    float4 s[4][6];                     // A^T m: rows of the output tile x the six columns
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      float4 m[6], o[4];
#pragma unroll
      for (int i = 0; i < 6; ++i) m[i] = *reinterpret_cast<const float4*>(in + (i * 6 + j) * slab);
      at6(m, o);
#pragma unroll
      for (int i = 0; i < 4; ++i) s[i][j] = o[i];
    }
    const float4 b = bias ? *reinterpret_cast<const float4*>(bias + c4 * 4)
                          : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int y = (4 * ty + i) * g.d + py;
      float4 o[4];
      at6(s[i], o);                     // (.) A
      if (y >= g.H) continue;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int x = (4 * tx + j) * g.d + px;
        if (x >= g.W) continue;
        float4 v = f4add4(o[j], b);
        if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        *reinterpret_cast<float4*>(Y + (((long long)n * g.H + y) * g.W + x) * Cout + c4 * 4) = v;
        vmax = fmaxf(fmaxf(vmax, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
      }
    }
  }
  if (amax_out) {          // |y| maximum of the layer's output, for the next layer's operand scale
    __shared__ float red[4];
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, d));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = vmax;
    __syncthreads();
    if (threadIdx.x == 0) {
      const unsigned v = __float_as_uint(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])));
      // one word for the whole grid: only blocks that would raise it touch it atomically
      if (v > __hip_atomic_load(amax_out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
        atomicMax(amax_out, v);
    }
  }
}

// U[xi][o][c] = (G g G^T)[xi] from the reference blob layout [O][I][3][3], taken in double and
// rounded once (G holds 1/6, 1/15, 1/30: the weights are frozen, the transform runs once)
//   G = [ 1/2 0 0 ; 1/6 1/6 1/6 ; 1/6 -1/6 1/6 ; 1/30 1/15 2/15 ; 16/15 -8/15 4/15 ; 0 0 1/2 ]
__device__ __forceinline__ void wino4_g(double g0, double g1, double g2, double (&o)[6]) {
  o[0] = g0 / 2.0;
  o[1] = (g0 + g1 + g2) / 6.0;
  o[2] = (g0 - g1 + g2) / 6.0;
  o[3] = (g0 + 2.0 * g1 + 4.0 * g2) / 30.0;
  o[4] = (16.0 * g0 - 8.0 * g1 + 4.0 * g2) / 15.0;
  o[5] = g2 / 2.0;
}

__global__ void wino4_weight_kernel(const float* __restrict__ Wt, int Cout, int Cin,
                                    float* __restrict__ U) {
  const long long total = (long long)Cout * Cin;
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total;
       t += (long long)gridDim.x * blockDim.x) {
    const float* gk = Wt + t * 9;
    double a[6][3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {       // G g
      double o[6];
      wino4_g(gk[j], gk[3 + j], gk[6 + j], o);
#pragma unroll
      for (int i = 0; i < 6; ++i) a[i][j] = o[i];
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) {       // (.) G^T
      double o[6];
      wino4_g(a[i][0], a[i][1], a[i][2], o);
#pragma unroll
      for (int j = 0; j < 6; ++j) U[(i * 6 + j) * total + t] = (float)o[j];
    }
  }
}

}  // namespace

extern "C" int naws_winograd4_weight_transform(const float* W_oihw, int Cout, int Cin, float* U,
                                               void* stream) {
  if (Cout <= 0 || Cin <= 0) return NAWS_ERR_SHAPE;
  NAWS_REQUIRE_PTR(W_oihw); NAWS_REQUIRE_PTR(U);
  const long long total = (long long)Cout * Cin;
  hipLaunchKernelGGL(wino4_weight_kernel, dim3((unsigned)std::min<long long>(naws_cdiv(total, 256), 4096)),
                     dim3(256), 0, (hipStream_t)stream, W_oihw, Cout, Cin, U);
  return naws_check_launch();
}

extern "C" int64_t naws_winograd4_f16x2_workspace_floats(int N, int H, int W, int Cin, int Cout,
                                                         int dilation) {
  if (N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || dilation < 1) return 0;
  const Wino4Geom g = wino4_geom(N, H, W, dilation);
  return 36 * g.P * ((int64_t)Cin + Cout) + g.P + 64;
}

extern "C" int naws_conv3x3_winograd4_nhwc_f16x2_fwd(const float* X, const void* U2,
                                                     const float* scaleU, const float* bias, int N,
                                                     int H, int W, int Cin, int Cout, int dilation,
                                                     int relu, float* workspace, float* Y,
                                                     const uint32_t* amax_in, uint32_t* amax_out,
                                                     void* stream) {
  if (N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return NAWS_ERR_SHAPE;
  if (dilation < 1) return NAWS_ERR_ARG;
  if (Cin % 32 != 0 || Cout % 4 != 0) return NAWS_ERR_UNSUPPORTED;
  NAWS_REQUIRE_PTR(X); NAWS_REQUIRE_PTR(U2); NAWS_REQUIRE_PTR(scaleU);
  NAWS_REQUIRE_PTR(workspace); NAWS_REQUIRE_PTR(Y); NAWS_REQUIRE_PTR(amax_in);
  if (amax_out == amax_in) return NAWS_ERR_ARG;
  if ((((uintptr_t)X | (uintptr_t)U2 | (uintptr_t)workspace | (uintptr_t)Y) & 15) != 0)
    return NAWS_ERR_ARG;
  const Wino4Geom g = wino4_geom(N, H, W, dilation);
  if (g.P > 0x7fffffffLL / 36) return NAWS_ERR_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  const long long slabM = g.P * Cout;
  unsigned short* Vp = (unsigned short*)workspace;                   // 36 * P * Cin floats
  float* Mb = workspace + 36 * g.P * Cin;
  float* invA = Mb + 36 * slabM;                                     // P floats
  {
    const long long total = g.P * (Cin / 4);
    // (one slab per workgroup, 192 threads: 0.1023 vs 0.1004 ms per conv4_2 layer, bit-identical)
    hipLaunchKernelGGL(wino4_input_h2_kernel<2>, dim3((unsigned)std::min<long long>(naws_cdiv(total, 128), 256 * 32)),
                       dim3(384), 0, s, X, g, Cin, (const unsigned*)amax_in, invA, Vp, (unsigned*)amax_out);
    int rc = naws_check_launch();
    if (rc != NAWS_OK) return rc;
  }
  int rc = naws_gemm_f32_f16x2_nt((int)g.P, Cout, Cin, Vp, g.P * 16, 36 * g.P * Cin, invA, U2,
                                  (int64_t)Cout * 16, (int64_t)36 * Cout * Cin, scaleU, Mb, Cout, 36,
                                  g.P * Cin, (int64_t)Cout * Cin, slabM, 0, Cout, NAWS_EPI_NONE,
                                  nullptr, 0, nullptr, 0, 1.0f, 0.0f, 0, 0, stream);
  if (rc != NAWS_OK) return rc;
  {
    const long long total = g.P * (Cout / 4);
    hipLaunchKernelGGL(wino4_output_kernel, dim3((unsigned)std::min<long long>(naws_cdiv(total, 256), 256 * 16)),
                       dim3(256), 0, s, Mb, g, Cout, slabM, bias, relu, Y, (unsigned*)amax_out);
  }
  return naws_check_launch();
}

// Winograd F(2x2, 3x3) convolution for the deep VGG layers (Cin, Cout >= 128), NHWC, fp32.
//
// ref: detectron/modeling/VGG16.py:24-46 (conv3_x .. conv5_x: 3x3, stride 1, pad == dilation).
// Y = A^T [ (G g G^T) (.) (B^T d B) ] A  per 2x2 output tile: 16 multiplies for 4 outputs
// instead of 36, i.e. 2.25x fewer MFMA flops than the direct implicit GEMM; the 16
// element-wise products over (tile, Cin) x (Cin, Cout) are one batch-16 fp32 MFMA GEMM
// (naws_gemm_f32), the two transforms are HBM-bound float4 kernels.  fp32 F(2,3) keeps the
// result within ~1e-6 relative of the direct sum (well inside the 1e-4 parity bar).
// Dilation 2 (conv5_x) = four independent dense convolutions on the (y%2, x%2) sub-grids.
//
// Workspace layout: V [16][P*Cin + PAD] | M [16][P*Cout + PAD], P = N * d*d * ceil(Hs/2) *
// ceil(Ws/2); PAD = 0 (spacing the 16 slabs apart was tried against HBM-channel aliasing: no effect).
#include <stdlib.h>
#include <type_traits>
#include "x3_common.h"

namespace {

// (padding the 16 V / M slabs apart was tried against channel aliasing: no effect)
constexpr long long wino_pad() { return 0; }

struct WinoGeom {
  int N, H, W, d, Hs, Ws, th, tw;
  long long P;
};

__host__ __device__ inline WinoGeom wino_geom(int N, int H, int W, int d) {
  WinoGeom g;
  g.N = N; g.H = H; g.W = W; g.d = d;
  g.Hs = (H + d - 1) / d;
  g.Ws = (W + d - 1) / d;
  g.th = (g.Hs + 1) / 2;
  g.tw = (g.Ws + 1) / 2;
  g.P = (long long)N * d * d * g.th * g.tw;
  return g;
}

__device__ __forceinline__ void tile_coords(const WinoGeom& g, long long p, int& n, int& py,
                                            int& px, int& ty, int& tx) {
  tx = (int)(p % g.tw); p /= g.tw;
  ty = (int)(p % g.th); p /= g.th;
  px = (int)(p % g.d); p /= g.d;
  py = (int)(p % g.d); p /= g.d;
  n = (int)p;
}

__device__ __forceinline__ float4 f4sub(float4 a, float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
__device__ __forceinline__ float4 f4add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }

// V[xi][p][c] = (B^T d B)[xi];  one lane = one tile x 4 channels
__global__ __launch_bounds__(256) void wino_input_kernel(const float* __restrict__ X, WinoGeom g,
                                                         int Cin, long long slab,
                                                         float* __restrict__ V) {
  const int c4n = Cin / 4;
  const long long total = g.P * c4n;
  for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < total;
       t += (long long)gridDim.x * 256) {
    const int c4 = (int)(t % c4n);
    const long long p = t / c4n;
    int n, py, px, ty, tx;
    tile_coords(g, p, n, py, px, ty, tx);
    float4 dd[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int ys = 2 * ty - 1 + i;
      const int y = ys * g.d + py;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int xs = 2 * tx - 1 + j;
        const int x = xs * g.d + px;
        const bool ok = ys >= 0 && xs >= 0 && y < g.H && x < g.W;
        dd[i][j] = ok ? *reinterpret_cast<const float4*>(
                            X + (((long long)n * g.H + y) * g.W + x) * Cin + c4 * 4)
                      : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
    float4 tt[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {       // B^T d
      tt[0][j] = f4sub(dd[0][j], dd[2][j]);
      tt[1][j] = f4add(dd[1][j], dd[2][j]);
      tt[2][j] = f4sub(dd[2][j], dd[1][j]);
      tt[3][j] = f4sub(dd[1][j], dd[3][j]);
    }
    float* out = V + p * Cin + c4 * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {       // (.) B
      *reinterpret_cast<float4*>(out + (i * 4 + 0) * slab) = f4sub(tt[i][0], tt[i][2]);
      *reinterpret_cast<float4*>(out + (i * 4 + 1) * slab) = f4add(tt[i][1], tt[i][2]);
      *reinterpret_cast<float4*>(out + (i * 4 + 2) * slab) = f4sub(tt[i][2], tt[i][1]);
      *reinterpret_cast<float4*>(out + (i * 4 + 3) * slab) = f4sub(tt[i][1], tt[i][3]);
    }
  }
}

// ---- fp16x2 form: the input transform emits the GEMM's f16 operand planes directly ---------------
// |x| maximum of a whole tensor (non-negative floats order like their bit patterns).
__global__ __launch_bounds__(256) void amax_all_kernel(const float* __restrict__ X, long long n4,
                                                       unsigned* __restrict__ out) {
  float m = 0.f;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    const float4 v = reinterpret_cast<const float4*>(X)[i];
    m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
  }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) m = fmaxf(m, __shfl_xor(m, d));
  __shared__ float red[4];
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned v = __float_as_uint(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])));
    if (v > __hip_atomic_load(out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(out, v);
  }
}

// V planes P[2][16][Cin/16][tiles][16] f16 = split of (B^T d B) * s, s = one power of two for the
// whole tensor: |B^T d B| <= 4 max|x|, so s = 2^(12 - floor(log2 max|x|)) keeps it below 2^15.
// One lane = one tile x 4 channels; lane order (4 channel groups of a 16-channel slab, then tiles):
// a wave writes 16 tiles x 32 B = 512 contiguous bytes per (xi, plane).
template <bool COLMAJOR = false>
__global__ __launch_bounds__(256) void wino_input_h2_kernel(const float* __restrict__ X, WinoGeom g,
                                                            int Cin, const unsigned* __restrict__ amax,
                                                            float* __restrict__ inv_scale,
                                                            unsigned short* __restrict__ Vp,
                                                            unsigned* __restrict__ amax_out) {
  // this layer's output maximum is accumulated by wino_output_kernel, stream-ordered after this
  if (amax_out && blockIdx.x == 0 && threadIdx.x == 0) *amax_out = 0u;
  int e = (int)((*amax >> 23) & 0xff);
  if (*amax == 0 || e == 0xff) e = 127 + 12;
  e = min(max(e, 40), 250);
  const float sc = __uint_as_float((unsigned)(266 - e) << 23);      // 2^(12 - (e - 127))
  const float isc = __uint_as_float((unsigned)(e - 12) << 23);
  const long long total = g.P * (Cin / 4);
  const long long xi_stride = (long long)Cin * g.P;                 // elements between the 16 xi
  const long long plane = 16 * xi_stride;
  for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < total;
       t += (long long)gridDim.x * 256) {
    const int cq = (int)(t & 3);
    const long long p = (t >> 2) % g.P;
    const int s = (int)((t >> 2) / g.P);
    const int c = s * 16 + cq * 4;
    if (s == 0 && cq == 0) inv_scale[p] = isc;
    int n, py, px, ty, tx;
    tile_coords(g, p, n, py, px, ty, tx);
    float4 dd[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int ys = 2 * ty - 1 + i;
      const int y = ys * g.d + py;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int xs = 2 * tx - 1 + j;
        const int x = xs * g.d + px;
        const bool ok = ys >= 0 && xs >= 0 && y < g.H && x < g.W;
        dd[i][j] = ok ? *reinterpret_cast<const float4*>(
                            X + (((long long)n * g.H + y) * g.W + x) * Cin + c)
                      : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
    float4 tt[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {       // B^T d
      tt[0][j] = f4sub(dd[0][j], dd[2][j]);
      tt[1][j] = f4add(dd[1][j], dd[2][j]);
      tt[2][j] = f4sub(dd[2][j], dd[1][j]);
      tt[3][j] = f4sub(dd[1][j], dd[3][j]);
    }
    unsigned short* hi = Vp + ((long long)s * g.P + p) * 16 + cq * 4;
    unsigned short* lo = hi + plane;
    // frequency (row i, column j) lives at index 4 i + j, or - COLMAJOR, the column form of the
    // batch GEMM - at 4 j + i, so that a column's four row blocks are one contiguous K range
#define NAWS_XI(I, J) ((COLMAJOR ? (J) * 4 + (I) : (I) * 4 + (J)) * xi_stride)
#pragma unroll
    for (int i = 0; i < 4; ++i) {       // (.) B
      put_h2(f4sub(tt[i][0], tt[i][2]), sc, hi + NAWS_XI(i, 0), lo + NAWS_XI(i, 0));
      put_h2(f4add(tt[i][1], tt[i][2]), sc, hi + NAWS_XI(i, 1), lo + NAWS_XI(i, 1));
      put_h2(f4sub(tt[i][2], tt[i][1]), sc, hi + NAWS_XI(i, 2), lo + NAWS_XI(i, 2));
      put_h2(f4sub(tt[i][1], tt[i][3]), sc, hi + NAWS_XI(i, 3), lo + NAWS_XI(i, 3));
    }
#undef NAWS_XI
  }
}

// The exact 3 x bf16 split of (B^T d B), written as operand planes P[3][16][Cin/16][tiles][16] by
// the transform itself (x == p1 + p2 + p3 exactly, split3: no scale, nothing to bound) - the
// strict plan's counterpart of wino_input_h2_kernel; replaces wino_input_kernel's fp32 V (16 P Cin
// floats written and read back) + naws_split_bf16x3 over it.
__device__ __forceinline__ void put_x3(float4 v, unsigned short* __restrict__ p1,
                                       unsigned short* __restrict__ p2,
                                       unsigned short* __restrict__ p3) {
  const float t[4] = {v.x, v.y, v.z, v.w};
  unsigned short a[4], b[4], c[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) split3(t[e], a[e], b[e], c[e]);
  u32x2 w;
  w.x = a[0] | ((unsigned)a[1] << 16); w.y = a[2] | ((unsigned)a[3] << 16);
  *reinterpret_cast<u32x2*>(p1) = w;
  w.x = b[0] | ((unsigned)b[1] << 16); w.y = b[2] | ((unsigned)b[3] << 16);
  *reinterpret_cast<u32x2*>(p2) = w;
  w.x = c[0] | ((unsigned)c[1] << 16); w.y = c[2] | ((unsigned)c[3] << 16);
  *reinterpret_cast<u32x2*>(p3) = w;
}

__global__ __launch_bounds__(256) void wino_input_x3_kernel(const float* __restrict__ X, WinoGeom g,
                                                            int Cin, unsigned short* __restrict__ Vp) {
  const long long total = g.P * (Cin / 4);
  const long long xi_stride = (long long)Cin * g.P;                 // elements between the 16 xi
  const long long plane = 16 * xi_stride;
  for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < total;
       t += (long long)gridDim.x * 256) {
    const int cq = (int)(t & 3);
    const long long p = (t >> 2) % g.P;
    const int s = (int)((t >> 2) / g.P);
    const int c = s * 16 + cq * 4;
    int n, py, px, ty, tx;
    tile_coords(g, p, n, py, px, ty, tx);
    float4 dd[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int ys = 2 * ty - 1 + i;
      const int y = ys * g.d + py;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int xs = 2 * tx - 1 + j;
        const int x = xs * g.d + px;
        const bool ok = ys >= 0 && xs >= 0 && y < g.H && x < g.W;
        dd[i][j] = ok ? *reinterpret_cast<const float4*>(
                            X + (((long long)n * g.H + y) * g.W + x) * Cin + c)
                      : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
    float4 tt[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {       // B^T d
      tt[0][j] = f4sub(dd[0][j], dd[2][j]);
      tt[1][j] = f4add(dd[1][j], dd[2][j]);
      tt[2][j] = f4sub(dd[2][j], dd[1][j]);
      tt[3][j] = f4sub(dd[1][j], dd[3][j]);
    }
    unsigned short* p1 = Vp + ((long long)s * g.P + p) * 16 + cq * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {       // (.) B   (the same expressions as wino_input_kernel)
      unsigned short* q = p1 + (i * 4) * xi_stride;
      put_x3(f4sub(tt[i][0], tt[i][2]), q, q + plane, q + 2 * plane);
      q += xi_stride;
      put_x3(f4add(tt[i][1], tt[i][2]), q, q + plane, q + 2 * plane);
      q += xi_stride;
      put_x3(f4sub(tt[i][2], tt[i][1]), q, q + plane, q + 2 * plane);
      q += xi_stride;
      put_x3(f4sub(tt[i][1], tt[i][3]), q, q + plane, q + 2 * plane);
    }
  }
}

// Y tile = A^T m A (+ bias, ReLU);  one lane = one tile x 4 output channels
__global__ __launch_bounds__(256) void wino_output_kernel(const float* __restrict__ M, WinoGeom g,
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
    tile_coords(g, p, n, py, px, ty, tx);
    const float* in = M + p * Cout + c4 * 4;
    float4 m[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) m[i][j] = *reinterpret_cast<const float4*>(in + (i * 4 + j) * slab);
    float4 s[2][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {       // A^T m
      s[0][j] = f4add(f4add(m[0][j], m[1][j]), m[2][j]);
      s[1][j] = f4sub(f4sub(m[1][j], m[2][j]), m[3][j]);
    }
    const float4 b = bias ? *reinterpret_cast<const float4*>(bias + c4 * 4)
                          : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int y = (2 * ty + i) * g.d + py;
      if (y >= g.H) continue;
      float4 o[2];
      o[0] = f4add(f4add(s[i][0], s[i][1]), s[i][2]);   // (.) A
      o[1] = f4sub(f4sub(s[i][1], s[i][2]), s[i][3]);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int x = (2 * tx + j) * g.d + px;
        if (x >= g.W) continue;
        float4 v = f4add(o[j], b);
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

#ifdef NAWS_AB   // the frequency-column form: measured slower (profiles/r04_wino_column_pmc.md); A/B build only
// Output transform of the frequency-column form: the batch GEMM has already taken A^T over the
// rows (S[j][0] = m0j + m1j + m2j, S[j][1] = m1j - m2j - m3j in its accumulators), so a tile reads
// 8 planes instead of 16 and only the column stage (.) A is left.  One lane = one tile x 4 channels.
__global__ __launch_bounds__(256) void wino_output_col_kernel(const float* __restrict__ S, WinoGeom g,
                                                              int Cout, const float* __restrict__ bias,
                                                              int relu, float* __restrict__ Y,
                                                              unsigned* __restrict__ amax_out) {
  const int c4n = Cout / 4;
  const long long total = g.P * c4n;
  const long long slab = g.P * Cout;                  // one (column, row) plane
  float vmax = 0.f;
  for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < total;
       t += (long long)gridDim.x * 256) {
    const int c4 = (int)(t % c4n);
    const long long p = t / c4n;
    int n, py, px, ty, tx;
    tile_coords(g, p, n, py, px, ty, tx);
    const float* in = S + p * Cout + c4 * 4;
    float4 s[2][4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int i = 0; i < 2; ++i) s[i][j] = *reinterpret_cast<const float4*>(in + (j * 2 + i) * slab);
    const float4 b = bias ? *reinterpret_cast<const float4*>(bias + c4 * 4)
                          : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int y = (2 * ty + i) * g.d + py;
      if (y >= g.H) continue;
      float4 o[2];
      o[0] = f4add(f4add(s[i][0], s[i][1]), s[i][2]);   // (.) A
      o[1] = f4sub(f4sub(s[i][1], s[i][2]), s[i][3]);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int x = (2 * tx + j) * g.d + px;
        if (x >= g.W) continue;
        float4 v = f4add(o[j], b);
        if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        *reinterpret_cast<float4*>(Y + (((long long)n * g.H + y) * g.W + x) * Cout + c4 * 4) = v;
        vmax = fmaxf(fmaxf(vmax, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
      }
    }
  }
  if (amax_out) {
    __shared__ float red[4];
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, d));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = vmax;
    __syncthreads();
    if (threadIdx.x == 0) {
      const unsigned v = __float_as_uint(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])));
      if (v > __hip_atomic_load(amax_out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
        atomicMax(amax_out, v);
    }
  }
}

#endif  // NAWS_AB

// ---- fp16x2: the 16 batched GEMMs AND the output transform in one kernel ------------------------
// (VERDICT r2 #4.)  The three-kernel route writes M = V U (16 x P x Cout fp32, 78 MB per conv4/5
// layer-image) and reads it back for A^T M A, and each of its 16 x tiles GEMM tiles has a K loop
// of only Cin / 16 = 16..32 steps: prologue + epilogue dominated (18-28 % MFMA busy).  Here a
// workgroup owns 64 Winograd tiles x 128 output channels for ALL 16 frequencies: one K loop of
// 16 x Cin (the operand planes are laid out so that frequency xi's slabs follow xi-1's: global
// slab index = xi * Cin/16 + s, for V and U alike), a per-frequency accumulator that is folded
//      Y[i][j] += AT[i][a] AT[j][b] * (V_xi U_xi) * (1/scaleV * 1/scaleU[xi][col]),  xi = 4 a + b,
//      AT = [[1, 1, 1, 0], [0, 1, -1, -1]]
// into the four output-position accumulators at each frequency boundary (36 of the 64
// coefficient pairs are non-zero: 2.25 fused multiply-adds per product element, on the VALU
// beside the next frequency's MFMAs), and an epilogue that adds the bias, applies the ReLU,
// writes the 2 x 2 output pixels of each tile straight into the NHWC tensor and reports max|y|.
// M never exists.  Pipeline = gemm_x3_kernel's: 3-stage LDS ring of 32-deep K-steps filled by
// LDS-DMA two steps ahead, counted s_waitcnt + one raw s_barrier per step, same swizzled LDS image.
struct WinoFusedArgs {
  const unsigned short* V;   // planes [2][16][Cin/16][P][16]
  const unsigned short* U;   // planes [2][16][Cin/16][Cout][16]
  const float* invV;         // [P], all equal: 1/scale of V
  const float* scaleU;       // [16][Cout]: 1/scale of U's rows
  const float* bias;
  float* Y;
  unsigned* amax_out;
  WinoGeom g;
  int Cin, Cout, relu;
  long long planeV, planeU;
  int tiles_m, tiles_n;
};

#ifdef NAWS_AB   // the LDS-DMA form of the fusion: measured, not faster (see below); A/B build only
constexpr int WF_BM = 64, WF_BN = 128, WF_KS = 2, WF_STAGES = 3;
constexpr int WF_A_PLANE = WF_BM * 32, WF_B_PLANE = WF_BN * 32, WF_NQ = 2 * WF_KS;
constexpr int WF_STAGE = WF_NQ * (WF_A_PLANE + WF_B_PLANE);            // 24 KB
constexpr int WF_PIECES = WF_NQ * (WF_BM / 32 + WF_BN / 32);           // 1 KB DMA pieces per step
constexpr int WF_G = WF_PIECES / 4;                                    // per wave
constexpr int WF_LDS = WF_STAGES * WF_STAGE + WF_BM * 8 + 16;   // ring | tile table | amax words

template <int XI>
__device__ __forceinline__ void wino_fold(f32x16 (&tmp)[2], f32x16 (&out)[4][2], const float (&c)[2]) {
  constexpr int a = XI / 4, b = XI % 4;
  constexpr int AT[2][4] = {{1, 1, 1, 0}, {0, 1, -1, -1}};
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      constexpr int unused = 0; (void)unused;
      const int sgn = AT[i][a] * AT[jj][b];
      if (sgn == 0) continue;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const float cj = sgn > 0 ? c[j] : -c[j];
#pragma unroll
        for (int e = 0; e < 16; ++e) out[i * 2 + jj][j][e] = fmaf(tmp[j][e], cj, out[i * 2 + jj][j][e]);
      }
    }
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int e = 0; e < 16; ++e) tmp[j][e] = 0.f;
}

__global__ __launch_bounds__(256, 2) void wino_fused_h2_kernel(WinoFusedArgs g) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smx[];
  // one contiguous tile range per XCD (as gemm_x3_kernel): the four channel tiles of a row of
  // Winograd tiles sit on one L2
  const int ntiles = g.tiles_m * g.tiles_n;
  int lid = blockIdx.x;
  {
    const int q = ntiles >> 3, rem = ntiles & 7, xcd = lid & 7, within = lid >> 3;
    lid = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + within;
  }
  const int tm = lid / g.tiles_n, tn = lid - tm * g.tiles_n;
  const int m0 = tm * WF_BM, n0 = tn * WF_BN;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid >> 1, wn = wid & 1;
  const int l31 = lane & 31, h = lane >> 5;
  const int P = (int)g.g.P;

  // output addressing of the workgroup's 64 tiles: element offset of output pixel (0, 0) of
  // the tile and which of its 2 x 2 pixels exist
  int* tab_off = reinterpret_cast<int*>(smx + WF_STAGES * WF_STAGE);
  int* tab_ok = tab_off + WF_BM;
  if (tid < WF_BM) {
    const int p = m0 + tid;
    int off = 0, ok = 0;
    if (p < P) {
      int n, py, px, ty, tx;
      tile_coords(g.g, p, n, py, px, ty, tx);
      const int y0 = 2 * ty * g.g.d + py, x0 = 2 * tx * g.g.d + px;
      off = ((n * g.g.H + y0) * g.g.W + x0) * g.Cout;
      ok = (y0 < g.g.H && x0 < g.g.W ? 1 : 0) | (y0 + g.g.d < g.g.H ? 2 : 0) | (x0 + g.g.d < g.g.W ? 4 : 0);
    }
    tab_off[tid] = off;
    tab_ok[tid] = ok;
  }

  // DMA pieces of this wave: q = wid + 4 k; q < 8: V piece (plane-slab q / 2, 32-row group q % 2),
  // else U piece (plane-slab (q - 8) / 4, row group (q - 8) % 4)
  const int lrow = lane >> 1;
  const int kslot = ((lane & 1) ^ ((lrow >> 3) & 1)) * 8;
  // k < 2: a V piece (q = wid + 4 k < 8), k >= 2: a U piece
  const unsigned short* src[WF_G];
  int dst[WF_G];
  const long long slabV = (long long)P * 16, slabU = (long long)g.Cout * 16;
#pragma unroll
  for (int k = 0; k < WF_G; ++k) {
    const int q = wid + 4 * k;
    if (k < 2) {
      const int pq = q / (WF_BM / 32), rg = q % (WF_BM / 32);
      const int pl = pq / WF_KS, ks = pq % WF_KS;
      src[k] = g.V + pl * g.planeV + ks * slabV + (long long)min(m0 + rg * 32 + lrow, P - 1) * 16 + kslot;
      dst[k] = pq * WF_A_PLANE + rg * 1024;
    } else {
      const int q2 = q - WF_NQ * (WF_BM / 32);
      const int pq = q2 / (WF_BN / 32), rg = q2 % (WF_BN / 32);
      const int pl = pq / WF_KS, ks = pq % WF_KS;
      src[k] = g.U + pl * g.planeU + ks * slabU +
               (long long)min(n0 + rg * 32 + lrow, g.Cout - 1) * 16 + kslot;
      dst[k] = WF_NQ * WF_A_PLANE + pq * WF_B_PLANE + rg * 1024;
    }
  }
  const long long stepV = WF_KS * slabV, stepU = WF_KS * slabU;     // elements per K-step
  auto issue = [&](int st) {                        // the next K-step of the stream, in order
    unsigned char* base = smx + st * WF_STAGE;
#pragma unroll
    for (int k = 0; k < WF_G; ++k) {
      __builtin_amdgcn_global_load_lds(NAWS_GLB_PTR(src[k]), NAWS_LDS_PTR(base + dst[k]), 16, 0, 0);
      src[k] += (k < 2) ? stepV : stepU;
    }
  };

  f32x16 tmp[2], out[4][2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      tmp[j][e] = 0.f;
#pragma unroll
      for (int q = 0; q < 4; ++q) out[q][j][e] = 0.f;
    }
  const int swz = (h ^ ((l31 >> 3) & 1)) * 16;
  const int rd_a = (wm * 32 + l31) * 32 + swz;
  const int rd_b = WF_NQ * WF_A_PLANE + (wn * 64 + l31) * 32 + swz;
  const int S2 = g.Cin / (16 * WF_KS);              // K-steps per frequency
  const int T = 16 * S2;
  const float isc = g.invV[0];
  const int colj = n0 + wn * 64 + l31;              // this lane's column of block j = 0 (j = 1: + 32)
  const int c0i = min(colj, g.Cout - 1), c1i = min(colj + 32, g.Cout - 1);

#pragma unroll
  for (int s = 0; s < WF_STAGES - 1; ++s)
    if (s < T) issue(s);
  int st_cur = 0, st_fill = WF_STAGES - 1;
  int xi = 0, left = S2;
  float cf[2] = {g.scaleU[c0i] * isc, g.scaleU[c1i] * isc};
  for (int t = 0; t < T; ++t) {
    if (t + WF_STAGES - 2 < T) wait_vmcnt<(WF_STAGES - 2) * WF_G>();
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    if (t + WF_STAGES - 1 < T) issue(st_fill);
    const unsigned char* st = smx + st_cur * WF_STAGE;
#pragma unroll
    for (int ks = 0; ks < WF_KS; ++ks) {
      f16x8 a[2], b[2][2];
#pragma unroll
      for (int pl = 0; pl < 2; ++pl) {
        a[pl] = *reinterpret_cast<const f16x8*>(st + rd_a + (pl * WF_KS + ks) * WF_A_PLANE);
#pragma unroll
        for (int j = 0; j < 2; ++j)
          b[pl][j] = *reinterpret_cast<const f16x8*>(st + rd_b + (pl * WF_KS + ks) * WF_B_PLANE + j * 1024);
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) tmp[j] = mfma16(a[0], b[0][j], tmp[j]);
#pragma unroll
      for (int j = 0; j < 2; ++j) tmp[j] = mfma16(a[0], b[1][j], tmp[j]);
#pragma unroll
      for (int j = 0; j < 2; ++j) tmp[j] = mfma16(a[1], b[0][j], tmp[j]);
    }
    st_cur = (st_cur + 1 == WF_STAGES) ? 0 : st_cur + 1;
    st_fill = (st_fill + 1 == WF_STAGES) ? 0 : st_fill + 1;
    if (--left == 0) {
      // frequency xi = 4 a + b is complete: Y[i][j] += AT[i][a] AT[j][b] * tmp * (1 / scales),
      // AT = [[1, 1, 1, 0], [0, 1, -1, -1]].  The coefficients are wave-uniform run-time values
      // (a zero coefficient still costs its multiply-add: 4 instead of 2.25 per element, in
      // exchange for ONE copy of the loop - 16 specialised copies spilt half the accumulators)
      const int a = xi >> 2, b = xi & 3;
      const float ra[2] = {a == 3 ? 0.f : 1.f, a == 0 ? 0.f : (a == 1 ? 1.f : -1.f)};
      const float rb[2] = {b == 3 ? 0.f : 1.f, b == 0 ? 0.f : (b == 1 ? 1.f : -1.f)};
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float cq = ra[q >> 1] * rb[q & 1];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const float c = cq * cf[j];
#pragma unroll
          for (int e = 0; e < 16; ++e) out[q][j][e] = fmaf(tmp[j][e], c, out[q][j][e]);
        }
      }
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) tmp[j][e] = 0.f;
      ++xi;
      left = S2;
      if (xi < 16) {
        cf[0] = g.scaleU[(long long)xi * g.Cout + c0i] * isc;
        cf[1] = g.scaleU[(long long)xi * g.Cout + c1i] * isc;
      }
    }
  }

  // ---- epilogue: bias, ReLU, the tile's 2 x 2 output pixels, max|y|
  float vmax = 0.f;
  const int dW = g.g.d * g.g.W * g.Cout, dX = g.g.d * g.Cout;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int col = colj + j * 32;
    if (col >= g.Cout) continue;
    const float bv = g.bias ? g.bias[col] : 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int row = wm * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
      const int ok = tab_ok[row];
      if (!(ok & 1)) continue;
      float* y00 = g.Y + tab_off[row] + col;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int i = q >> 1, jj = q & 1;
        if ((i && !(ok & 2)) || (jj && !(ok & 4))) continue;
        float v = out[q][j][e] + bv;
        if (g.relu) v = fmaxf(v, 0.f);
        y00[i * dW + jj * dX] = v;
        vmax = fmaxf(vmax, fabsf(v));
      }
    }
  }
  if (g.amax_out) {
    vmax = wave_max(vmax);
    // (no static LDS in this kernel: its dynamic limit is raised to the full 160 KB)
    float* red = reinterpret_cast<float*>(smx + WF_STAGES * WF_STAGE + WF_BM * 8);
    if (lane == 0) red[wid] = vmax;
    __syncthreads();
    if (tid == 0) {
      const unsigned v = __float_as_uint(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])));
      if (v > __hip_atomic_load(g.amax_out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
        atomicMax(g.amax_out, v);
    }
  }
}

#endif  // NAWS_AB

// ---- the same fusion with wave-private weight fragments ------------------------------------------
// The form above moves 24 KB per K-step through LDS-DMA for 12 MFMAs per wave: 6 DMA issues per
// wave and step (100+ cycles each beside MFMAs and LDS reads) cost more than the MFMAs they feed
// (measured: 0.148 ms per conv4_2 image against 0.118 for the GEMM + transform pair it replaces).
// Here the four waves are a 1 x 4 grid - wave wc owns ALL rows of the workgroup's 32 TI Winograd
// tiles x the 32 output channels wc - so a wave's MFMA B operand (32 channels x 16 deep x 2 planes)
// is exactly one 16-byte buffer load per lane and plane straight from U's slab-major planes (1 KB
// contiguous per wave-instruction), used by no other wave: it skips LDS, is prefetched two K-steps
// ahead in a static 4-deep register ring, and needs no barrier.  Only V's 32 TI rows go through
// the (3-stage, LDS-DMA) ring: TI pieces per wave and step instead of 6.
template <int TI, int KS>
__global__ __launch_bounds__(256, 2) void wino_fused_wp_kernel(WinoFusedArgs g) {
  constexpr int BM = 32 * TI, STAGES = 3;
  constexpr int RING = KS >= 4 ? 2 : 4, AHEAD = KS >= 4 ? 1 : 2;    // U fragment ring / prefetch distance
  constexpr int A_PLANE = BM * 32, NQ = 2 * KS, STAGE = NQ * A_PLANE;
  constexpr int GA = NQ * TI / 4;                   // V pieces (1 KB) per wave per K-step
  constexpr int G = GA + NQ;                        // + this wave's U fragments
  static_assert(NQ * TI % 4 == 0, "pieces per wave");
  extern __shared__ __attribute__((aligned(16))) unsigned char smx[];
  const int ntiles = g.tiles_m * g.tiles_n;
  int lid = blockIdx.x;
  {
    const int q = ntiles >> 3, rem = ntiles & 7, xcd = lid & 7, within = lid >> 3;
    lid = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + within;
  }
  const int tm = lid / g.tiles_n, tn = lid - tm * g.tiles_n;
  const int m0 = tm * BM, n0 = tn * 128;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wc = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, h = lane >> 5;
  const int P = (int)g.g.P;

  int* tab_off = reinterpret_cast<int*>(smx + STAGES * STAGE);
  int* tab_ok = tab_off + BM;
  if (tid < BM) {
    const int p = m0 + tid;
    int off = 0, ok = 0;
    if (p < P) {
      int n, py, px, ty, tx;
      tile_coords(g.g, p, n, py, px, ty, tx);
      const int y0 = 2 * ty * g.g.d + py, x0 = 2 * tx * g.g.d + px;
      off = ((n * g.g.H + y0) * g.g.W + x0) * g.Cout;
      ok = (y0 < g.g.H && x0 < g.g.W ? 1 : 0) | (y0 + g.g.d < g.g.H ? 2 : 0) | (x0 + g.g.d < g.g.W ? 4 : 0);
    }
    tab_off[tid] = off;
    tab_ok[tid] = ok;
  }

  // V pieces of this wave: q = wc * GA + k -> (plane-slab q / TI, 32-row group q % TI)
  const int lrow = lane >> 1;
  const int kslot = ((lane & 1) ^ ((lrow >> 3) & 1)) * 8;
  const long long slabV = (long long)P * 16;
  const unsigned short* src[GA];
  int dst[GA];
#pragma unroll
  for (int k = 0; k < GA; ++k) {
    const int q = wc * GA + k;
    const int pq = q / TI, rg = q % TI;
    const int pl = pq / KS, ks = pq % KS;
    src[k] = g.V + pl * g.planeV + ks * slabV + (long long)min(m0 + rg * 32 + lrow, P - 1) * 16 + kslot;
    dst[k] = pq * A_PLANE + rg * 1024;
  }
  const long long stepV = KS * slabV;
  auto issueA = [&](int st) {
    unsigned char* base = smx + st * STAGE;
#pragma unroll
    for (int k = 0; k < GA; ++k) {
      __builtin_amdgcn_global_load_lds(NAWS_GLB_PTR(src[k]), NAWS_LDS_PTR(base + dst[k]), 16, 0, 0);
      src[k] += stepV;
    }
  };
  // U fragments: lane (l31, h) holds k-half h of channel n0 + 32 wc + l31
  const __amdgpu_buffer_rsrc_t rsU = __builtin_amdgcn_make_buffer_rsrc(
      (void*)g.U, 0, (int)(2 * g.planeU * 2), 0x00020000);
  const int voffB = (min(n0 + wc * 32 + l31, g.Cout - 1) * 16 + h * 8) * 2;
  const int slabBytes = g.Cout * 32, planeBytes = (int)(g.planeU * 2);
  f16x8 bq[RING][KS][2];
  auto loadB = [&](auto ring_tag, int step) {
    constexpr int R = decltype(ring_tag)::value;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int pl = 0; pl < 2; ++pl) {
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(
            rsU, voffB, (step * KS + ks) * slabBytes + pl * planeBytes, 0);
        bq[R][ks][pl] = *reinterpret_cast<const f16x8*>(&v);
      }
  };

  f32x16 tmp[TI], out[4][TI];
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      tmp[i][e] = 0.f;
#pragma unroll
      for (int q = 0; q < 4; ++q) out[q][i][e] = 0.f;
    }
  const int rd_a = l31 * 32 + (h ^ ((l31 >> 3) & 1)) * 16;
  const int S2 = g.Cin / (16 * KS);
  const int T = 16 * S2;                            // a multiple of 4 (host: Cin % (64 KS) == 0)
  const float isc = g.invV[0];
  const int col = n0 + wc * 32 + l31;
  const int ci = min(col, g.Cout - 1);

  issueA(0); loadB(std::integral_constant<int, 0>{}, 0);
  issueA(1);
  if constexpr (AHEAD == 2) loadB(std::integral_constant<int, 1>{}, 1);
  int st_cur = 0, st_fill = STAGES - 1;
  int xi = 0, left = S2;
  float cf = g.scaleU[ci] * isc;
  auto step = [&](auto u_tag, int t) {
    constexpr int U = decltype(u_tag)::value;
    // in flight behind step t's operands: V pieces of step t+1 (and, AHEAD 2, its U fragments)
    if (t + 1 < T) wait_vmcnt<(AHEAD == 2 ? G : GA)>();
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    if (t + 2 < T) issueA(st_fill);
    if (t + AHEAD < T) loadB(std::integral_constant<int, (U + AHEAD) % RING>{}, t + AHEAD);
    const unsigned char* st = smx + st_cur * STAGE;
    // every fragment of the step is requested before its first MFMA: one LDS latency per step
    f16x8 a[KS][2][TI];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int pl = 0; pl < 2; ++pl)
#pragma unroll
        for (int i = 0; i < TI; ++i)
          a[ks][pl][i] = *reinterpret_cast<const f16x8*>(st + rd_a + (pl * KS + ks) * A_PLANE + i * 1024);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
      for (int i = 0; i < TI; ++i) tmp[i] = mfma16(a[ks][0][i], bq[U % RING][ks][0], tmp[i]);
#pragma unroll
      for (int i = 0; i < TI; ++i) tmp[i] = mfma16(a[ks][0][i], bq[U % RING][ks][1], tmp[i]);
#pragma unroll
      for (int i = 0; i < TI; ++i) tmp[i] = mfma16(a[ks][1][i], bq[U % RING][ks][0], tmp[i]);
    }
    st_cur = (st_cur + 1 == STAGES) ? 0 : st_cur + 1;
    st_fill = (st_fill + 1 == STAGES) ? 0 : st_fill + 1;
    if (--left == 0) {
      const int a = xi >> 2, b = xi & 3;
      const float ra[2] = {a == 3 ? 0.f : 1.f, a == 0 ? 0.f : (a == 1 ? 1.f : -1.f)};
      const float rb[2] = {b == 3 ? 0.f : 1.f, b == 0 ? 0.f : (b == 1 ? 1.f : -1.f)};
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float c = ra[q >> 1] * rb[q & 1] * cf;
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
          for (int e = 0; e < 16; ++e) out[q][i][e] = fmaf(tmp[i][e], c, out[q][i][e]);
      }
#pragma unroll
      for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) tmp[i][e] = 0.f;
      ++xi;
      left = S2;
      if (xi < 16) cf = g.scaleU[(long long)xi * g.Cout + ci] * isc;
    }
  };
  for (int t = 0; t < T; t += 4) {
    step(std::integral_constant<int, 0>{}, t);
    step(std::integral_constant<int, 1>{}, t + 1);
    step(std::integral_constant<int, 2>{}, t + 2);
    step(std::integral_constant<int, 3>{}, t + 3);
  }

  float vmax = 0.f;
  const int dW = g.g.d * g.g.W * g.Cout, dX = g.g.d * g.Cout;
  if (col < g.Cout) {
    const float bv = g.bias ? g.bias[col] : 0.f;
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
        const int ok = tab_ok[row];
        if (!(ok & 1)) continue;
        float* y00 = g.Y + tab_off[row] + col;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int ii = q >> 1, jj = q & 1;
          if ((ii && !(ok & 2)) || (jj && !(ok & 4))) continue;
          float v = out[q][i][e] + bv;
          if (g.relu) v = fmaxf(v, 0.f);
          y00[ii * dW + jj * dX] = v;
          vmax = fmaxf(vmax, fabsf(v));
        }
      }
  }
  if (g.amax_out) {
    vmax = wave_max(vmax);
    float* red = reinterpret_cast<float*>(smx + STAGES * STAGE + BM * 8);
    if (lane == 0) red[wc] = vmax;
    __syncthreads();
    if (tid == 0) {
      const unsigned v = __float_as_uint(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])));
      if (v > __hip_atomic_load(g.amax_out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
        atomicMax(g.amax_out, v);
    }
  }
}

template <int TI, int KS>
int launch_wino_fused_wp(WinoFusedArgs& a, hipStream_t s) {
  constexpr int BM = 32 * TI;
  a.tiles_m = (int)naws_cdiv(a.g.P, BM);
  a.tiles_n = (int)naws_cdiv(a.Cout, 128);
  const size_t lds = (size_t)3 * (2 * KS * BM * 32) + BM * 8 + 16;
  auto kern = wino_fused_wp_kernel<TI, KS>;
  if (naws_allow_lds(kern) != NAWS_OK) return NAWS_ERR_LAUNCH;
  hipLaunchKernelGGL(kern, dim3((unsigned)(a.tiles_m * a.tiles_n)), dim3(256), lds, s, a);
  return naws_check_launch();
}

// U[xi][o][c] = (G g G^T)[xi] from the reference blob layout [O][I][3][3]
__global__ void wino_weight_kernel(const float* __restrict__ Wt, int Cout, int Cin,
                                   float* __restrict__ U) {
  const long long total = (long long)Cout * Cin;
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total;
       t += (long long)gridDim.x * blockDim.x) {
    const float* gk = Wt + t * 9;
    float g[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) g[i][j] = gk[i * 3 + j];
    float a[4][3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {       // G g
      a[0][j] = g[0][j];
      a[1][j] = 0.5f * (g[0][j] + g[1][j] + g[2][j]);
      a[2][j] = 0.5f * (g[0][j] - g[1][j] + g[2][j]);
      a[3][j] = g[2][j];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {       // (.) G^T
      const float u0 = a[i][0];
      const float u1 = 0.5f * (a[i][0] + a[i][1] + a[i][2]);
      const float u2 = 0.5f * (a[i][0] - a[i][1] + a[i][2]);
      const float u3 = a[i][2];
      U[(i * 4 + 0) * total + t] = u0;
      U[(i * 4 + 1) * total + t] = u1;
      U[(i * 4 + 2) * total + t] = u2;
      U[(i * 4 + 3) * total + t] = u3;
    }
  }
}

}  // namespace

extern "C" int64_t naws_winograd_workspace_floats(int N, int H, int W, int Cin, int Cout,
                                                  int dilation) {
  if (N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || dilation < 1) return 0;
  const WinoGeom g = wino_geom(N, H, W, dilation);
  return 16 * (g.P * ((int64_t)Cin + Cout) + 2 * 4096);
}

extern "C" int naws_winograd_weight_transform(const float* W_oihw, int Cout, int Cin, float* U,
                                              void* stream) {
  if (Cout <= 0 || Cin <= 0) return NAWS_ERR_SHAPE;
  NAWS_REQUIRE_PTR(W_oihw); NAWS_REQUIRE_PTR(U);
  const long long total = (long long)Cout * Cin;
  hipLaunchKernelGGL(wino_weight_kernel, dim3((unsigned)std::min<long long>(naws_cdiv(total, 256), 4096)),
                     dim3(256), 0, (hipStream_t)stream, W_oihw, Cout, Cin, U);
  return naws_check_launch();
}

extern "C" int naws_conv3x3_winograd_nhwc_fwd(const float* X, const float* U, const float* bias,
                                              int N, int H, int W, int Cin, int Cout, int dilation,
                                              int relu, float* workspace, float* Y, void* stream) {
  if (N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return NAWS_ERR_SHAPE;
  if (dilation < 1) return NAWS_ERR_ARG;
  if (Cin % 4 != 0 || Cout % 4 != 0) return NAWS_ERR_UNSUPPORTED;
  NAWS_REQUIRE_PTR(X); NAWS_REQUIRE_PTR(U); NAWS_REQUIRE_PTR(workspace); NAWS_REQUIRE_PTR(Y);
  if ((((uintptr_t)X | (uintptr_t)U | (uintptr_t)workspace | (uintptr_t)Y) & 15) != 0)
    return NAWS_ERR_ARG;
  const WinoGeom g = wino_geom(N, H, W, dilation);
  if (g.P > 0x7fffffffLL) return NAWS_ERR_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  const long long pad = wino_pad();
  const long long slabV = g.P * Cin + pad, slabM = g.P * Cout + pad;
  float* V = workspace;
  float* Mb = workspace + 16 * slabV;
  {
    const long long total = g.P * (Cin / 4);
    hipLaunchKernelGGL(wino_input_kernel, dim3((unsigned)std::min<long long>(naws_cdiv(total, 256), 256 * 16)),
                       dim3(256), 0, s, X, g, Cin, slabV, V);
    int rc = naws_check_launch();
    if (rc != NAWS_OK) return rc;
  }
  int rc = naws_gemm_f32(0, 1, (int)g.P, Cout, Cin, V, Cin, U, Cin, Mb, Cout, 16, slabV,
                         (int64_t)Cout * Cin, slabM, NAWS_EPI_NONE, nullptr, 0, nullptr, 0, 1.0f,
                         0.0f, 0, 0, stream);
  if (rc != NAWS_OK) return rc;
  {
    const long long total = g.P * (Cout / 4);
    hipLaunchKernelGGL(wino_output_kernel, dim3((unsigned)std::min<long long>(naws_cdiv(total, 256), 256 * 16)),
                       dim3(256), 0, s, Mb, g, Cout, slabM, bias, relu, Y, (unsigned*)nullptr);
  }
  return naws_check_launch();
}

// Same convolution with the 16 batched GEMMs on the bf16 matrix cores in the exact 3xbf16 split
// (csrc/gemm_x3.hip): V is split into planes after the input transform, U3 is the split of the
// transformed weight (naws_split_bf16x3 of U viewed [16][Cout][Cin]).
extern "C" int64_t naws_winograd_f32x3_workspace_floats(int N, int H, int W, int Cin, int Cout,
                                                        int dilation) {
  if (N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || dilation < 1) return 0;
  const WinoGeom g = wino_geom(N, H, W, dilation);
  return naws_winograd_workspace_floats(N, H, W, Cin, Cout, dilation) + 24 * g.P * (int64_t)Cin + 64;
}

extern "C" int naws_conv3x3_winograd_nhwc_f32x3_fwd(const float* X, const void* U3,
                                                    const float* bias, int N, int H, int W, int Cin,
                                                    int Cout, int dilation, int relu,
                                                    float* workspace, float* Y, void* stream) {
  if (N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return NAWS_ERR_SHAPE;
  if (dilation < 1) return NAWS_ERR_ARG;
  if (Cin % 16 != 0 || Cout % 4 != 0) return NAWS_ERR_UNSUPPORTED;
  NAWS_REQUIRE_PTR(X); NAWS_REQUIRE_PTR(U3); NAWS_REQUIRE_PTR(workspace); NAWS_REQUIRE_PTR(Y);
  if ((((uintptr_t)X | (uintptr_t)U3 | (uintptr_t)workspace | (uintptr_t)Y) & 15) != 0)
    return NAWS_ERR_ARG;
  const WinoGeom g = wino_geom(N, H, W, dilation);
  if (g.P > 0x7fffffffLL) return NAWS_ERR_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  const long long pad = wino_pad();
  const long long slabV = g.P * Cin + pad, slabM = g.P * Cout + pad;
  float* Mb = workspace + 16 * slabV;
  float* Vp = Mb + 16 * slabM;
  Vp += (16 - ((uintptr_t)Vp & 63) / 4) & 15;            // 64-byte aligned planes
  {
    // the input transform writes the three bf16 operand planes itself (the first 16 slabV floats
    // of the workspace, once the fp32 V of this layout's first version, are unused)
    const long long total = g.P * (Cin / 4);
    hipLaunchKernelGGL(wino_input_x3_kernel, dim3((unsigned)std::min<long long>(naws_cdiv(total, 256), 256 * 16)),
                       dim3(256), 0, s, X, g, Cin, (unsigned short*)Vp);
    int rc = naws_check_launch();
    if (rc != NAWS_OK) return rc;
  }
  int rc;
  rc = naws_gemm_f32x3_nt((int)g.P, Cout, Cin, Vp, g.P * 16, 16 * g.P * Cin, U3, (int64_t)Cout * 16,
                          (int64_t)16 * Cout * Cin, Mb, Cout, 16, g.P * Cin, (int64_t)Cout * Cin,
                          slabM, NAWS_EPI_NONE, nullptr, 0, nullptr, 0, 1.0f, 0.0f, 0, 0, stream);
  if (rc != NAWS_OK) return rc;
  {
    const long long total = g.P * (Cout / 4);
    hipLaunchKernelGGL(wino_output_kernel, dim3((unsigned)std::min<long long>(naws_cdiv(total, 256), 256 * 16)),
                       dim3(256), 0, s, Mb, g, Cout, slabM, bias, relu, Y, (unsigned*)nullptr);
  }
  return naws_check_launch();
}

// Same convolution with the 16 batched GEMMs on the f16 matrix cores in the row-scaled 2 x f16
// split (csrc/gemm_x3.hip, naws_gemm_f32_f16x2_nt): the input transform writes the scaled hi / lo
// planes itself (no fp32 V round trip); U2 / scaleU = naws_split_f16x2 of the transformed weight
// viewed [16][Cout][Cin] (planes [2][16][Cin/16][Cout][16], scales [2][16][Cout], scaleU = [1]).
extern "C" int64_t naws_winograd_f16x2_workspace_floats(int N, int H, int W, int Cin, int Cout,
                                                        int dilation) {
  if (N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || dilation < 1) return 0;
  const WinoGeom g = wino_geom(N, H, W, dilation);
  return 16 * g.P * ((int64_t)Cin + Cout) + g.P + 64;
}

extern "C" int naws_conv3x3_winograd_nhwc_f16x2_fwd(const float* X, const void* U2,
                                                    const float* scaleU, const float* bias, int N,
                                                    int H, int W, int Cin, int Cout, int dilation,
                                                    int relu, float* workspace, float* Y,
                                                    const uint32_t* amax_in, uint32_t* amax_out,
                                                    void* stream) {
  if (N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return NAWS_ERR_SHAPE;
  if (dilation < 1) return NAWS_ERR_ARG;
  if (Cin % 32 != 0 || Cout % 4 != 0) return NAWS_ERR_UNSUPPORTED;
  NAWS_REQUIRE_PTR(X); NAWS_REQUIRE_PTR(U2); NAWS_REQUIRE_PTR(scaleU);
  NAWS_REQUIRE_PTR(workspace); NAWS_REQUIRE_PTR(Y);
  if ((((uintptr_t)X | (uintptr_t)U2 | (uintptr_t)workspace | (uintptr_t)Y) & 15) != 0)
    return NAWS_ERR_ARG;
  const WinoGeom g = wino_geom(N, H, W, dilation);
  if (g.P > 0x7fffffffLL) return NAWS_ERR_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  const long long slabM = g.P * Cout;
  unsigned short* Vp = (unsigned short*)workspace;                   // 16 * P * Cin floats
  float* Mb = workspace + 16 * g.P * Cin;
  float* invA = Mb + 16 * slabM;                                     // P floats
  const unsigned* amax = (const unsigned*)amax_in;
  {
    if (!amax) {           // no bound handed over by the producer of X: take the maximum here
      unsigned* own = (unsigned*)(invA + g.P);
      if (hipMemsetAsync(own, 0, sizeof(unsigned), s) != hipSuccess) return NAWS_ERR_LAUNCH;
      const long long n4 = (long long)N * H * W * Cin / 4;
      hipLaunchKernelGGL(amax_all_kernel, dim3((unsigned)std::min<long long>(naws_cdiv(n4, 256 * 8), 2048)),
                         dim3(256), 0, s, X, n4, own);
      amax = own;
    }
    const long long total = g.P * (Cin / 4);
    hipLaunchKernelGGL(wino_input_h2_kernel<false>, dim3((unsigned)std::min<long long>(naws_cdiv(total, 256), 256 * 16)),
                       dim3(256), 0, s, X, g, Cin, amax, invA, Vp, (unsigned*)amax_out);
    int rc = naws_check_launch();
    if (rc != NAWS_OK) return rc;
  }
  // The fused GEMM + output-transform kernel where it wins (tools/ab_wino.py, interleaved, one
  // image per launch as the engine runs them; ms incl. the 0.035 ms input transform):
  //     layer shape            three kernels   fused (32 tiles x 128 ch, 64-deep K-steps)
  //     256 -> 256, 150x250        0.219           0.169
  //     256 -> 512,  75x125        0.111           0.102
  //     512 -> 512,  75x125        0.149           0.175     (every fused form: 0.17-0.21)
  // A fused workgroup walks all 16 frequencies, so U's planes (16 Cin Cout 4 B) are streamed by
  // every row of tiles: at 512 x 512 they are 16.8 MB, four times an XCD's L2, and come from the
  // Infinity Cache each time - the three-kernel route keeps whole frequencies (1 MB of U) on one
  // XCD.  Rule: fuse while U's planes stay within 8.4 MB.  Knob "wino": 0 = this rule, 1 = never
  // fuse, 3 = always (A/B build: 2 / 4 / 5 / 6 select the other fused forms).
  const int wv = naws_knob(NAWS_KNOB_WINO);
  const bool fits = (long long)N * H * W * Cout < 0x7fffffffLL && 16 * g.P * Cin < 0x7fffffffLL &&
                    (long long)64 * Cout * Cin < 0x7fffffffLL && Cin % 256 == 0;
  if (wv != 1 && fits && (wv != 0 || (long long)Cin * Cout <= 256 * 512)) {
    WinoFusedArgs a{};
    a.V = Vp; a.U = (const unsigned short*)U2; a.invV = invA; a.scaleU = scaleU; a.bias = bias;
    a.Y = Y; a.amax_out = (unsigned*)amax_out; a.g = g; a.Cin = Cin; a.Cout = Cout; a.relu = relu;
    a.planeV = 16 * g.P * Cin; a.planeU = (long long)16 * Cout * Cin;
#ifdef NAWS_AB
    if (wv == 2) {
      a.tiles_m = (int)naws_cdiv(g.P, WF_BM); a.tiles_n = (int)naws_cdiv(Cout, WF_BN);
      if (naws_allow_lds(wino_fused_h2_kernel) != NAWS_OK) return NAWS_ERR_LAUNCH;
      hipLaunchKernelGGL(wino_fused_h2_kernel, dim3((unsigned)(a.tiles_m * a.tiles_n)), dim3(256),
                         WF_LDS, s, a);
      return naws_check_launch();
    }
    if (wv == 4) return launch_wino_fused_wp<1, 2>(a, s);
    if (wv == 5) return launch_wino_fused_wp<2, 2>(a, s);
    if (wv == 7) return launch_wino_fused_wp<2, 4>(a, s);
#endif
    return launch_wino_fused_wp<1, 4>(a, s);
  }
  int rc = naws_gemm_f32_f16x2_nt((int)g.P, Cout, Cin, Vp, g.P * 16, 16 * g.P * Cin, invA, U2,
                                  (int64_t)Cout * 16, (int64_t)16 * Cout * Cin, scaleU, Mb, Cout, 16,
                                  g.P * Cin, (int64_t)Cout * Cin, slabM, 0, Cout, NAWS_EPI_NONE,
                                  nullptr, 0, nullptr, 0, 1.0f, 0.0f, 0, 0, stream);
  if (rc != NAWS_OK) return rc;
  {
    const long long total = g.P * (Cout / 4);
    hipLaunchKernelGGL(wino_output_kernel, dim3((unsigned)std::min<long long>(naws_cdiv(total, 256), 256 * 16)),
                       dim3(256), 0, s, Mb, g, Cout, slabM, bias, relu, Y, (unsigned*)amax_out);
  }
  return naws_check_launch();
}

#ifdef NAWS_AB
// The frequency-COLUMN form of naws_conv3x3_winograd_nhwc_f16x2_fwd (same operator, same arguments,
// reference detectron/modeling/VGG16.py:24-46): the batch GEMM runs one K loop of 4 Cin per
// (tile block, channel block, column j) and takes the A^T stage over the rows in two accumulator
// sets, so M shrinks from 16 to 8 planes and the tiles' prologues / epilogues from 16 to 4 per
// position.  U2 = naws_split_f16x2 of the transformed weight regrouped [4 j][Cout][4 i x Cin]
// (planes [2][4][4 Cin / 16][Cout][16], scaleU [4][Cout]: one scale per (column, channel) row).
// Results differ from the 16-plane form in the last bits (the row sums are taken in the MFMA
// accumulators).  Cin % 32 == 0, Cout % 4 == 0.
extern "C" int naws_conv3x3_winograd_nhwc_f16x2_col_fwd(const float* X, const void* U2,
                                                        const float* scaleU, const float* bias,
                                                        int N, int H, int W, int Cin, int Cout,
                                                        int dilation, int relu, float* workspace,
                                                        float* Y, const uint32_t* amax_in,
                                                        uint32_t* amax_out, void* stream) {
  if (N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return NAWS_ERR_SHAPE;
  if (dilation < 1) return NAWS_ERR_ARG;
  if (Cin % 32 != 0 || Cout % 4 != 0) return NAWS_ERR_UNSUPPORTED;
  NAWS_REQUIRE_PTR(X); NAWS_REQUIRE_PTR(U2); NAWS_REQUIRE_PTR(scaleU);
  NAWS_REQUIRE_PTR(workspace); NAWS_REQUIRE_PTR(Y); NAWS_REQUIRE_PTR(amax_in);
  if ((((uintptr_t)X | (uintptr_t)U2 | (uintptr_t)workspace | (uintptr_t)Y) & 15) != 0)
    return NAWS_ERR_ARG;
  const WinoGeom g = wino_geom(N, H, W, dilation);
  if (g.P > 0x7fffffffLL) return NAWS_ERR_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  unsigned short* Vp = (unsigned short*)workspace;                   // 16 * P * Cin floats
  float* Sb = workspace + 16 * g.P * Cin;                            // 8 * P * Cout floats
  float* invA = Sb + 16 * g.P * Cout;                                // P floats (layout of the 16-plane form)
  {
    const long long total = g.P * (Cin / 4);
    hipLaunchKernelGGL(wino_input_h2_kernel<true>, dim3((unsigned)std::min<long long>(naws_cdiv(total, 256), 256 * 16)),
                       dim3(256), 0, s, X, g, Cin, (const unsigned*)amax_in, invA, Vp, (unsigned*)amax_out);
    int rc = naws_check_launch();
    if (rc != NAWS_OK) return rc;
  }
  int rc = naws_wino_col_gemm_impl((int)g.P, Cout, Cin, Vp, invA, U2, scaleU, Sb, s);
  if (rc != NAWS_OK) return rc;
  {
    const long long total = g.P * (Cout / 4);
    hipLaunchKernelGGL(wino_output_col_kernel, dim3((unsigned)std::min<long long>(naws_cdiv(total, 256), 256 * 16)),
                       dim3(256), 0, s, Sb, g, Cout, bias, relu, Y, (unsigned*)amax_out);
  }
  return naws_check_launch();
}
#endif  // NAWS_AB

// Shared helpers for the gfx950 kernels behind include/naws.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <algorithm>
#include "naws.h"

#define NAWS_WAVE 64

extern thread_local int g_naws_last_hip_error;

static inline int naws_check_launch() {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    g_naws_last_hip_error = (int)e;
    return NAWS_ERR_LAUNCH;
  }
  return NAWS_OK;
}

// Launch attributes are per (kernel, device): a kernel that needs more than 64 KB of dynamic LDS
// has hipFuncAttributeMaxDynamicSharedMemorySize raised once on EVERY device it is launched on
// (table in misc_ops.hip, keyed by the kernel's host pointer, one bit per device ordinal;
// naws_launch_state_reset() forgets it).  Returns NAWS_OK or NAWS_ERR_LAUNCH.
int naws_allow_lds_impl(const void* kernel, int bytes);
template <typename K>
static inline int naws_allow_lds(K kernel, int bytes = 160 * 1024) {
  return naws_allow_lds_impl(reinterpret_cast<const void*>(kernel), bytes);
}

// Winograd frequency-column batch GEMM (gemm_x3.hip, used by winograd.hip): for each of the 4
// columns j, S0_j = sum over rows i = 0, 1, 2 and S1_j = (i = 1) - (i = 2) - (i = 3) of V_ij U_ij^T
// in ONE K loop of 4 Cin.  V2: planes [2][4 j][4 Cin / 16][P][16] (column-major frequency order),
// invV [P]; U2: planes [2][4 j][4 Cin / 16][Cout][16], scaleU [4][Cout]; S: [4 j][2][P][Cout] fp32.
int naws_wino_col_gemm_impl(int P, int Cout, int Cin, const void* V2, const float* invV,
                            const void* U2, const float* scaleU, float* S, hipStream_t stream);

// Tuning knobs of the A/B tools (naws_set_variant; defaults below).  None changes a result.
enum NawsKnob { NAWS_KNOB_GEMM = 0, NAWS_KNOB_X3, NAWS_KNOB_H2, NAWS_KNOB_CONV_RING,
                NAWS_KNOB_CONV_BN, NAWS_KNOB_ROI_NW, NAWS_KNOB_WINO, NAWS_KNOB_SPLIT, NAWS_KNOB_SGD_WGS,
                NAWS_KNOB_COUNT };
int naws_knob(int knob);

#define NAWS_REQUIRE_PTR(p) \
  do {                      \
    if ((p) == nullptr) return NAWS_ERR_NULL; \
  } while (0)

static inline int64_t naws_cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Wave-level reductions over all 64 lanes (result valid in every lane).
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// Counter-based keep decision for Dropout: one 64-bit mix of (seed, index).
// keep  <=>  u >= ratio with u uniform in [0,1) on 24 bits.
__device__ __host__ __forceinline__ uint32_t naws_hash_u32(uint64_t seed, uint64_t idx) {
  uint64_t z = seed + 0x9E3779B97F4A7C15ull * (idx + 1);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z = z ^ (z >> 31);
  return (uint32_t)(z >> 40);  // 24 bits
}
__device__ __host__ __forceinline__ bool naws_keep(uint64_t seed, uint64_t idx, uint32_t thr24) {
  return naws_hash_u32(seed, idx) >= thr24;
}
static inline uint32_t naws_drop_threshold(float ratio) {
  double t = (double)ratio * 16777216.0;
  if (t < 0) t = 0;
  if (t > 16777216.0) t = 16777216.0;
  return (uint32_t)t;
}

// fp16x2 operand scaling: from the bit pattern of an upper bound of max|x| to (scale, 1/scale),
// scale = 2^(14 - floor(log2 bound)) so that bound * scale lies in [2^14, 2^15).  NaN / inf / zero
// bounds keep scale 1 (a NaN then propagates through the f16 conversion); the scale is capped at
// 2^101 (bounds below 2^-87 sit lower in the f16 range: absolute floor 2^-126).
__device__ __forceinline__ void naws_f16x2_scales(unsigned bound_bits, float& s, float& inv) {
  int e = (int)((bound_bits >> 23) & 0xff);
  if (bound_bits == 0 || e == 0xff) e = 127 + 14;
  e = min(max(e, 40), 250);
  s = __uint_as_float((unsigned)(268 - e) << 23);
  inv = __uint_as_float((unsigned)(e - 14) << 23);
}

// ---- ACM weight-decay momentum SGD, one element ------------------------------------------------------
// One element of the update, every product and sum rounded on its own (no FMA contraction): the
// reference's CPU operator is scalar C++ built for generic x86-64 (math::Scale, math::Axpy, then
// `lr * g + momentum * m`, acm_weightdecay_momentum_sgd_op.h:79-109), and the oracle restates it
// with -ffp-contract=off; both SGD kernels below go through this one function, so they agree
// with each other bit for bit.  a = the accumulated gradient (acm + g).
__device__ __forceinline__ void sgd_elem(float a, float& m, float& p, float scale, float wd, float LR,
                                         float momentum, int nesterov) {
#pragma clang fp contract(off)          // (the build's -ffp-contract=fast-honor-pragmas honours this)
  float t = a * scale;                                 // Normalize
  t = t + wd * p;                                      // Regularize (Axpy)
  if (!nesterov) {
    const float adj = LR * t + momentum * m;
    m = adj;
    p = p - adj;
  } else {
    const float mi = m;
    const float mi_new = momentum * mi + LR * t;
    m = mi_new;
    p = p - ((1.0f + momentum) * mi_new - momentum * mi);
  }
}

// ---- |C| maxima reported by a GEMM epilogue ---------------------------------------------------------
// The fp16x2 operand split needs max|x| per row (NT operand) and / or per column (transposed
// operand) of a matrix a GEMM has just produced.  Instead of re-reading the matrix, the producing
// kernel folds the values it stores into two device vectors of bit patterns (non-negative floats
// order like unsigned words): registers -> wave shuffles -> one guarded atomicMax per row / column
// per wave.  NaNs are skipped exactly as fmaxf skips them in the stand-alone pass (amax_kernel), so
// the maxima - and therefore the planes - are bit-identical to the two-pass route.
struct NawsAmax {
  unsigned* rowmax;      // [batch][nseg][M]: max over the columns of segment col / seg_cols (nullable)
  unsigned* colmax;      // [batch][N] (nullable)
  const float* colmul;   // colmax is taken over |C[m][n] * colmul[m]| (nullable = 1)
  int seg_cols;          // columns per rowmax segment (a multiple of the widest tile, or >= N)
  long long sRow, sCol;  // elements between batch items
};

__device__ __forceinline__ void naws_atomic_max_bits(unsigned* p, float v) {
  const unsigned b = __float_as_uint(v);
  if (b > __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(p, b);
}

// acc[i][j] = 32x32 blocks in the v_mfma_f32_32x32x* C/D layout (col = lane & 31,
// row = (e & 3) + 8 (e >> 2) + 4 (lane >> 5)), holding the values as stored.
template <int TI, int TJ, typename V16>
__device__ __forceinline__ void naws_tile_amax_32(const V16 (&acc)[TI][TJ], int row_base, int col_base,
                                                  int M, int N, int lane, const NawsAmax& a,
                                                  long long bz) {
  const int l31 = lane & 31, h = lane >> 5;
  float colm[TJ];
#pragma unroll
  for (int j = 0; j < TJ; ++j) colm[j] = 0.f;
  unsigned* rowmax = a.rowmax ? a.rowmax + bz * a.sRow + (long long)(col_base / a.seg_cols) * M : nullptr;
#pragma unroll
  for (int i = 0; i < TI; ++i) {
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int row = row_base + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
      const float cmul = (a.colmul && row < M) ? fabsf(a.colmul[row]) : 1.f;
      float rm = 0.f;
#pragma unroll
      for (int j = 0; j < TJ; ++j) {
        const int col = col_base + j * 32 + l31;
        if (row < M && col < N) {
          const float av = fabsf(acc[i][j][e]);
          rm = fmaxf(rm, av);
          colm[j] = fmaxf(colm[j], av * cmul);
        }
      }
      if (rowmax) {
#pragma unroll
        for (int d = 16; d > 0; d >>= 1) rm = fmaxf(rm, __shfl_xor(rm, d));
        if (l31 == 0 && row < M && rm > 0.f) naws_atomic_max_bits(rowmax + row, rm);
      }
    }
  }
  if (a.colmax) {
#pragma unroll
    for (int j = 0; j < TJ; ++j) {
      const float cm = fmaxf(colm[j], __shfl_xor(colm[j], 32));
      const int col = col_base + j * 32 + l31;
      if (h == 0 && col < N && cm > 0.f) naws_atomic_max_bits(a.colmax + bz * a.sCol + col, cm);
    }
  }
}

// acc[i][j] = TRANSPOSED 16x16 blocks (MFMA issued with the operands swapped): row = lane & 15,
// col = (lane >> 4) * 4 + e.
template <int TI, int TJ, typename V4>
__device__ __forceinline__ void naws_tile_amax_16t(const V4 (&acc)[TI][TJ], int row_base, int col_base,
                                                   int M, int N, int lane, const NawsAmax& a,
                                                   long long bz) {
  const int l15 = lane & 15, kg = lane >> 4;
  float colm[TJ][4];
#pragma unroll
  for (int j = 0; j < TJ; ++j)
#pragma unroll
    for (int e = 0; e < 4; ++e) colm[j][e] = 0.f;
  unsigned* rowmax = a.rowmax ? a.rowmax + bz * a.sRow + (long long)(col_base / a.seg_cols) * M : nullptr;
#pragma unroll
  for (int i = 0; i < TI; ++i) {
    const int row = row_base + i * 16 + l15;
    const float cmul = (a.colmul && row < M) ? fabsf(a.colmul[row]) : 1.f;
    float rm = 0.f;
#pragma unroll
    for (int j = 0; j < TJ; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int col = col_base + j * 16 + kg * 4 + e;
        if (row < M && col < N) {
          const float av = fabsf(acc[i][j][e]);
          rm = fmaxf(rm, av);
          colm[j][e] = fmaxf(colm[j][e], av * cmul);
        }
      }
    if (rowmax) {
      rm = fmaxf(rm, __shfl_xor(rm, 16));
      rm = fmaxf(rm, __shfl_xor(rm, 32));
      if (kg == 0 && row < M && rm > 0.f) naws_atomic_max_bits(rowmax + row, rm);
    }
  }
  if (a.colmax) {
#pragma unroll
    for (int j = 0; j < TJ; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float cm = colm[j][e];
#pragma unroll
        for (int d = 8; d > 0; d >>= 1) cm = fmaxf(cm, __shfl_xor(cm, d));
        const int col = col_base + j * 16 + kg * 4 + e;
        if (l15 == 0 && col < N && cm > 0.f) naws_atomic_max_bits(a.colmax + bz * a.sCol + col, cm);
      }
  }
}

// acc[i][j] = 16x16 blocks in the v_mfma_f32_16x16x* C/D layout (col = lane & 15,
// row = (lane >> 4) * 4 + e).
template <int TI, int TJ, typename V4>
__device__ __forceinline__ void naws_tile_amax_16(const V4 (&acc)[TI][TJ], int row_base, int col_base,
                                                  int M, int N, int lane, const NawsAmax& a,
                                                  long long bz) {
  const int l15 = lane & 15, kg = lane >> 4;
  float colm[TJ];
#pragma unroll
  for (int j = 0; j < TJ; ++j) colm[j] = 0.f;
  unsigned* rowmax = a.rowmax ? a.rowmax + bz * a.sRow + (long long)(col_base / a.seg_cols) * M : nullptr;
#pragma unroll
  for (int i = 0; i < TI; ++i) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int row = row_base + i * 16 + kg * 4 + e;
      const float cmul = (a.colmul && row < M) ? fabsf(a.colmul[row]) : 1.f;
      float rm = 0.f;
#pragma unroll
      for (int j = 0; j < TJ; ++j) {
        const int col = col_base + j * 16 + l15;
        if (row < M && col < N) {
          const float av = fabsf(acc[i][j][e]);
          rm = fmaxf(rm, av);
          colm[j] = fmaxf(colm[j], av * cmul);
        }
      }
      if (rowmax) {
#pragma unroll
        for (int d = 8; d > 0; d >>= 1) rm = fmaxf(rm, __shfl_xor(rm, d));
        if (l15 == 0 && row < M && rm > 0.f) naws_atomic_max_bits(rowmax + row, rm);
      }
    }
  }
  if (a.colmax) {
#pragma unroll
    for (int j = 0; j < TJ; ++j) {
      float cm = fmaxf(colm[j], __shfl_xor(colm[j], 16));
      cm = fmaxf(cm, __shfl_xor(cm, 32));
      const int col = col_base + j * 16 + l15;
      if (kg == 0 && col < N && cm > 0.f) naws_atomic_max_bits(a.colmax + bz * a.sCol + col, cm);
    }
  }
}

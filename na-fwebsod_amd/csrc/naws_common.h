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

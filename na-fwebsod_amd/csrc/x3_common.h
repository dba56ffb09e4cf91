// Shared device helpers of the split-operand MFMA kernels (gemm_x3.hip, conv_x3.hip).
#pragma once
#include <stdlib.h>
#include "naws_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x16 mfma16(bf16x8 a, bf16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mfma16(f16x8 a, f16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}
template <bool F16> struct OperandVec { typedef bf16x8 type; };
template <> struct OperandVec<true> { typedef f16x8 type; };

#define NAWS_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define NAWS_GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_vm_lgkm0() {       // + this wave's LDS writes are done
  asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N) : "memory");
}

__device__ __forceinline__ void split3(float a, unsigned short& p1, unsigned short& p2,
                                       unsigned short& p3) {
  const __bf16 h1 = (__bf16)a;
  float r = a - (float)h1;
  if (!(fabsf(a) <= 3.4028234e38f)) r = 0.f;        // inf / NaN live in plane 1 only
  const __bf16 h2 = (__bf16)r;
  const __bf16 h3 = (__bf16)(r - (float)h2);
  p1 = *reinterpret_cast<const unsigned short*>(&h1);
  p2 = *reinterpret_cast<const unsigned short*>(&h2);
  p3 = *reinterpret_cast<const unsigned short*>(&h3);
}

// Four fp32 values -> scaled f16 hi / lo planes (8 bytes each): the element step of every
// transform that writes fp16x2 operand planes itself (winograd.hip, winograd4.hip).
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void put_h2(float4 v, float sc, unsigned short* __restrict__ hi,
                                       unsigned short* __restrict__ lo) {
  const float t[4] = {v.x * sc, v.y * sc, v.z * sc, v.w * sc};
  unsigned short h[4], l[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const _Float16 a = (_Float16)t[e];
    float r = t[e] - (float)a;
    if (!(fabsf(t[e]) <= 65504.f)) r = 0.f;
    const _Float16 b = (_Float16)r;
    h[e] = *reinterpret_cast<const unsigned short*>(&a);
    l[e] = *reinterpret_cast<const unsigned short*>(&b);
  }
  u32x2 wh, wl;
  wh.x = h[0] | ((unsigned)h[1] << 16); wh.y = h[2] | ((unsigned)h[3] << 16);
  wl.x = l[0] | ((unsigned)l[1] << 16); wl.y = l[2] | ((unsigned)l[3] << 16);
  *reinterpret_cast<u32x2*>(hi) = wh;
  *reinterpret_cast<u32x2*>(lo) = wl;
}

__device__ __forceinline__ void f16x2_scales(unsigned amax_bits, float& s, float& inv) {
  naws_f16x2_scales(amax_bits, s, inv);      // naws_common.h
}

}  // namespace

// fp32 GEMM on the bf16 matrix cores: the "3 x bf16" split.
//
// gfx950 has no fast fp32 MFMA path (v_mfma_f32_32x32x2_f32: 157 TFLOP/s) but a 16x faster
// bf16 one (v_mfma_f32_32x32x16_bf16: 2.5 PFLOP/s, fp32 accumulate).  An fp32 number is EXACTLY
// the sum of three bf16 numbers (8 + 8 + 8 significand bits, round-to-nearest residues):
//      a = a1 + a2 + a3,   a1 = bf16(a), a2 = bf16(a - a1), a3 = bf16(a - a1 - a2)
// so  a*b = sum_{p,q} a_p b_q  and keeping the six terms with p + q <= 4
//      a1b1 + (a1b2 + a2b1) + (a1b3 + a2b2 + a3b1)
// drops only terms below 2^-26 |ab| - a quarter of an fp32 ulp of the product, less than the
// rounding an fp32 FMA chain commits anyway.  Each bf16 x bf16 product is exact (16-bit
// significand) and is accumulated in fp32 by the MFMA, exactly like the fp32 MFMA accumulates.
// Six bf16 MFMA passes cost 6/16 of one fp32 MFMA pass: an fp32-accurate GEMM at up to 2.67x the
// fp32 MFMA peak.  tests/test_gpu_x3.py pins the claim: against a float64 product the error of
// this kernel is not larger than that of the fp32-MFMA kernel (gemm_f32.hip) on the same data.
//
// Non-finite operands: NaN propagates as in fp32; an infinite operand gives NaN instead of inf
// whenever the other factor has an exactly-zero low plane (inf * 0 in a cross term) - a state
// the fp32 path has also left the realm of useful numbers in.
//
// Operands arrive pre-split ("planes": P[3][rows][K] bf16, K-contiguous, K % 16 == 0, made by
// naws_split_bf16x3, which also transposes for the dW = dY^T X form).  With 48 MFMAs (1536
// cycles) per wave per 16-deep K-step the staging pipeline can be the simplest correct one:
// LDS-DMA (global_load_lds_dwordx4, no VGPR round trip) of K-step t+1 into the other LDS stage
// while step t multiplies, one barrier per step; two 4-wave workgroups per CU (72 KB LDS each)
// drift against each other and keep the MFMA pipe fed.
//
// replaces: Caffe2 FC / FCGradient for fc6 / fc7 (reference detectron/modeling/wsl_heads.py:
// 674-679, webly_heads.py:490-498), same as gemm_f32.hip.
#include <stdlib.h>
#include "naws_common.h"

static int g_x3_variant = -1;

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

struct XArgs {
  const unsigned short* A;   // planes [3][K/16][rowsA][16]
  const unsigned short* B;   // planes [3][K/16][rowsB][16]
  float* C;
  int M, N, K;
  int ldc;
  long long planeA, planeB;  // elements between planes
  long long slabA, slabB;    // elements between 16-deep K slabs (= rows * 16)
  long long sA, sB, sC, sBias;
  const float* bias;
  const float* aux;
  int ldaux;
  float alpha;
  unsigned drop_thr;
  float drop_scale;
  unsigned long long seed;
  int epilogue, accumulate;
  int tiles_m, tiles_n;
  const float* rs;           // fp16x2: per-row / per-column power-of-two factors undoing the
  const float* cs;           //         operand scaling (null otherwise)
  long long sRs, sCs;
  NawsAmax am;               // |C| maxima for the consumer's operand split (naws_common.h)
};

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

// Operand planes are stored K-slab-major, P[plane][k/16][row][k%16]: the 16-deep K-step of a
// BM-row tile is ONE contiguous BM*32-byte run per plane, so every LDS-DMA wave-instruction
// reads 1 KB of consecutive HBM bytes.  LDS image of a plane-stage: [row][2 slots of 16 B]; the
// slot of k-half h of row r is h ^ ((r >> 3) & 1), which spreads the 16 lanes of a ds_read_b128
// group over all 64 banks.  The DMA writes lane-linearly, so the swizzle is applied to the
// global source address (it only permutes 16-byte halves inside one 32-byte row record).
//
// Pipeline: ring of STAGES LDS stages, DMA runs STAGES-1 K-steps ahead.  Per step: counted
// s_waitcnt vmcnt (retires this wave's pieces of step t, leaves the younger steps in flight),
// one raw s_barrier (everybody's pieces landed; everybody is done reading the stage about to be
// refilled), issue step t+STAGES-1, then 6 x TI x TJ MFMAs on step t.
//
// NPL = 3, KS = 1: the fp32x3 GEMM (6 MFMA terms per 16-deep K slab).  NPL = 1, KS = 4: the same
// pipeline as a plain bf16 GEMM for the bf16 plan (one plane, 64-deep K-steps of 4 slabs, so a
// step still carries 32 MFMAs per wave between barriers).  F16 (NPL = 2): the fp16x2 GEMM, see
// the note above naws_split_f16x2 - two f16 planes per operand, 3 MFMA terms per slab.
template <int BM, int BN, int WM, int WN, int STAGES, int NPL = 3, int KS = 1, bool F16 = false>
__global__ __launch_bounds__(64 * WM * WN, (WM * WN >= 8 ? 1 : 2)) void gemm_x3_kernel(XArgs g) {
  static_assert(!F16 || NPL == 2, "fp16x2 uses two planes");
  typedef typename OperandVec<F16>::type vec_t;
  constexpr int NT = 64 * WM * WN;
  constexpr int WTM = BM / WM, WTN = BN / WN;
  constexpr int TI = WTM / 32, TJ = WTN / 32;
  constexpr int A_PLANE = BM * 32, B_PLANE = BN * 32;       // bytes per plane-slab per stage
  constexpr int NQ = NPL * KS;                              // plane-slabs per operand per stage
  constexpr int STAGE = NQ * (A_PLANE + B_PLANE);
  constexpr int PIECE_ROWS = NT / 2;                        // rows one DMA round covers
  constexpr int PA = BM / PIECE_ROWS, PB = BN / PIECE_ROWS;
  constexpr int G = NQ * (PA + PB);                         // DMA instructions per thread per step
  static_assert(BM % PIECE_ROWS == 0 && BN % PIECE_ROWS == 0, "tile vs workgroup");
  static_assert((STAGES - 2) * G <= 63, "vmcnt range");
  extern __shared__ __attribute__((aligned(16))) unsigned char smx[];

  const int ntiles = g.tiles_m * g.tiles_n;
  int lid = blockIdx.x;
  {
    const int q = ntiles >> 3, rem = ntiles & 7, xcd = lid & 7, within = lid >> 3;
    lid = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + within;
  }
  constexpr int GM = 8;
  const int per_group = GM * g.tiles_n;
  const int grp = lid / per_group;
  const int first_m = grp * GM;
  const int gsz = min(g.tiles_m - first_m, GM);
  const int tm = first_m + (lid % per_group) % gsz;
  const int tn = (lid % per_group) / gsz;
  const int m0 = tm * BM, n0 = tn * BN;

  const long long bz = blockIdx.z;
  const unsigned short* A = g.A + bz * g.sA;
  const unsigned short* B = g.B + bz * g.sB;
  float* C = g.C + bz * g.sC;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid / WN, wn = wid % WN;
  const int l31 = lane & 31, h = lane >> 5;

  // DMA sources: thread -> (row = piece*PIECE_ROWS + tid/2, physical slot = tid&1); rows past
  // the edge re-read the last row (their products land in accumulator rows never stored)
  const int lrow = tid >> 1;
  const int kslot = ((tid & 1) ^ ((lrow >> 3) & 1)) * 8;
  const unsigned short* srcA[PA];
  const unsigned short* srcB[PB];
#pragma unroll
  for (int p = 0; p < PA; ++p)
    srcA[p] = A + (long long)min(m0 + p * PIECE_ROWS + lrow, g.M - 1) * 16 + kslot;
#pragma unroll
  for (int p = 0; p < PB; ++p)
    srcB[p] = B + (long long)min(n0 + p * PIECE_ROWS + lrow, g.N - 1) * 16 + kslot;

  auto issue = [&](int t, int st) {
    unsigned char* base = smx + st * STAGE + wid * 1024;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int pl = q / KS, ks = q % KS;
      const long long ka = (long long)(t * KS + ks) * g.slabA, kb = (long long)(t * KS + ks) * g.slabB;
#pragma unroll
      for (int p = 0; p < PA; ++p)
        __builtin_amdgcn_global_load_lds(NAWS_GLB_PTR(srcA[p] + pl * g.planeA + ka),
                                         NAWS_LDS_PTR(base + q * A_PLANE + p * (NT * 16)), 16, 0, 0);
#pragma unroll
      for (int p = 0; p < PB; ++p)
        __builtin_amdgcn_global_load_lds(NAWS_GLB_PTR(srcB[p] + pl * g.planeB + kb),
                                         NAWS_LDS_PTR(base + NQ * A_PLANE + q * B_PLANE + p * (NT * 16)),
                                         16, 0, 0);
    }
  };

  f32x16 acc[TI][TJ];
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int j = 0; j < TJ; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int rd_a = (wm * WTM + l31) * 32 + ((h ^ ((l31 >> 3) & 1)) * 16);
  const int rd_b = NQ * A_PLANE + (wn * WTN + l31) * 32 + ((h ^ ((l31 >> 3) & 1)) * 16);

  const int T = g.K / (16 * KS);
#pragma unroll
  for (int s = 0; s < STAGES - 1; ++s)
    if (s < T) issue(s, s);
  int st_cur = 0, st_fill = STAGES - 1;
  for (int t = 0; t < T; ++t) {
    if (t + STAGES - 2 < T) wait_vmcnt<(STAGES - 2) * G>();
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    if (t + STAGES - 1 < T) issue(t + STAGES - 1, st_fill);
    const unsigned char* st = smx + st_cur * STAGE;
#define NAWS_X3_TERM(P, Q)                                                                      \
  _Pragma("unroll") for (int i = 0; i < TI; ++i) _Pragma("unroll") for (int j = 0; j < TJ; ++j) \
      acc[i][j] = mfma16(a[P][i], b[Q][j], acc[i][j]);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      vec_t a[NPL][TI], b[NPL][TJ];
#pragma unroll
      for (int pl = 0; pl < NPL; ++pl) {
#pragma unroll
        for (int i = 0; i < TI; ++i)
          a[pl][i] = *reinterpret_cast<const vec_t*>(st + rd_a + (pl * KS + ks) * A_PLANE + i * 1024);
#pragma unroll
        for (int j = 0; j < TJ; ++j)
          b[pl][j] = *reinterpret_cast<const vec_t*>(st + rd_b + (pl * KS + ks) * B_PLANE + j * 1024);
      }
      // consecutive MFMAs go to different accumulators
      NAWS_X3_TERM(0, 0)
      if constexpr (NPL >= 2) {
        NAWS_X3_TERM(0, 1)
        NAWS_X3_TERM(1, 0)
      }
      if constexpr (NPL == 3) {
        NAWS_X3_TERM(1, 1)
        NAWS_X3_TERM(0, 2)
        NAWS_X3_TERM(2, 0)
      }
    }
#undef NAWS_X3_TERM
    st_cur = (st_cur + 1 == STAGES) ? 0 : st_cur + 1;
    st_fill = (st_fill + 1 == STAGES) ? 0 : st_fill + 1;
  }

  const float* bias = g.bias ? g.bias + bz * g.sBias : nullptr;
  const float* aux = g.aux ? g.aux + bz * g.sC : nullptr;
  const int epi = g.epilogue;
  if constexpr (F16) {
    // undo the row scaling first (powers of two: exact).  Accumulator register e of a lane in
    // half h is row (e&3) + 8(e>>2) + 4h of the 32x32 block: lane l fetches the factor of row l,
    // v_readlane hands each register its row's factor (one load per block instead of 16).
    const float* rs = g.rs + bz * g.sRs;
#pragma unroll
    for (int i = 0; i < TI; ++i) {
      const int mine = __float_as_int(rs[min(m0 + wm * WTM + i * 32 + l31, g.M - 1)]);
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int l0 = (e & 3) + 8 * (e >> 2);
        const float r0 = __int_as_float(__builtin_amdgcn_readlane(mine, l0));
        const float r1 = __int_as_float(__builtin_amdgcn_readlane(mine, l0 + 4));
        const float r = h ? r1 : r0;
#pragma unroll
        for (int j = 0; j < TJ; ++j) acc[i][j][e] *= r;
      }
    }
  }
#pragma unroll
  for (int j = 0; j < TJ; ++j) {
    const int col = n0 + wn * WTN + j * 32 + l31;
    if (col >= g.N) continue;
    const float bv = (bias && epi >= NAWS_EPI_BIAS && epi <= NAWS_EPI_BIAS_RELU_DROP) ? bias[col] : 0.f;
    float cscale = 1.f;
    if constexpr (F16) cscale = g.cs[bz * g.sCs + col];
#pragma unroll
    for (int i = 0; i < TI; ++i) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = m0 + wm * WTM + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
        if (row >= g.M) continue;
        float v = acc[i][j][e];
        if constexpr (F16) v *= cscale;                 // power of two: exact
        v += bv;
        if (epi == NAWS_EPI_BIAS_RELU || epi == NAWS_EPI_BIAS_RELU_DROP) v = fmaxf(v, 0.f);
        if (epi == NAWS_EPI_BIAS_RELU_DROP) {
          const unsigned long long idx =
              (unsigned long long)bz * g.M * g.N + (unsigned long long)row * g.N + col;
          v = naws_keep(g.seed, idx, g.drop_thr) ? v * g.drop_scale : 0.f;
        } else if (epi == NAWS_EPI_GATE_POS) {
          v = (aux[row * g.ldaux + col] > 0.f) ? v * g.alpha : 0.f;
        }
        const int idx = row * g.ldc + col;      // < 2^31, checked on the host
        if (g.accumulate) v += C[idx];
        C[idx] = v;
        acc[i][j][e] = v;
      }
    }
  }
  if (g.am.rowmax || g.am.colmax)
    naws_tile_amax_32<TI, TJ>(acc, m0 + wm * WTM, n0 + wn * WTN, g.M, g.N, lane, g.am, bz);
}

template <int BM, int BN, int WM, int WN, int STAGES, int NPL = 3, int KS = 1, bool F16 = false>
int launch_x3(XArgs& g, int batch, hipStream_t s) {
  g.tiles_m = (int)naws_cdiv(g.M, BM);
  g.tiles_n = (int)naws_cdiv(g.N, BN);
  const size_t lds = (size_t)STAGES * NPL * KS * (BM + BN) * 32;
  auto kern = gemm_x3_kernel<BM, BN, WM, WN, STAGES, NPL, KS, F16>;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  dim3 grid((unsigned)(g.tiles_m * g.tiles_n), 1, batch);
  hipLaunchKernelGGL(kern, grid, dim3(64 * WM * WN), lds, s, g);
  return naws_check_launch();
}

// ---- the same pipeline on the 16x16x32 MFMA shape ------------------------------------------------
// MI355X_MICROARCH.md 'DVFS give-back' item 7: in power-limited 16-bit MFMA loops the chip holds a
// higher clock on v_mfma_f32_16x16x32 than on 32x32x16 at equal cycles per flop (measured there:
// 1.12-1.14x the flop/s with every operand re-read from LDS).  The fc6 GEMMs are exactly that
// regime (PMC: 1.5 GHz under the 32x32x16 form).  Same operand planes, same LDS-DMA ring, same
// bytes read from LDS per flop; differences:
//   * a 32-deep MFMA K = two 16-deep slabs: lane l reads row (l & 15) of slab (l >> 5), k-half
//     (l >> 4) & 1; with THIS lane->address map the un-swizzled [row][2 x 16 B] image is the
//     conflict-free one for ds_read_b128's 16-lane groups (rows r and r+8 of a group carry opposite
//     k-halves), so the DMA source keeps its natural order;
//   * accumulator block = 16x16, 4 registers per lane: row (l >> 4) * 4 + e, column l & 15.
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 mfma32(bf16x8 a, bf16x8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mfma32(f16x8 a, f16x8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}

template <int BM, int BN, int WM, int WN, int STAGES, int NPL, int KS, bool F16>
__global__ __launch_bounds__(64 * WM * WN, (WM * WN >= 8 ? 1 : 2)) void gemm_x3_m16_kernel(XArgs g) {
  static_assert(KS % 2 == 0, "a 16x16x32 MFMA spans two 16-deep slabs");
  static_assert(NPL <= 2, "one- or two-plane operands");
  typedef typename OperandVec<F16>::type vec_t;
  constexpr int NT = 64 * WM * WN;
  constexpr int WTM = BM / WM, WTN = BN / WN;
  constexpr int TI = WTM / 16, TJ = WTN / 16;
  constexpr int IH = TI >= 8 ? 2 : 1, TIH = TI / IH;        // A fragments are read in row halves
  constexpr int A_PLANE = BM * 32, B_PLANE = BN * 32;
  constexpr int NQ = NPL * KS;
  constexpr int STAGE = NQ * (A_PLANE + B_PLANE);
  constexpr int PIECE_ROWS = NT / 2;
  constexpr int PA = BM / PIECE_ROWS, PB = BN / PIECE_ROWS;
  constexpr int G = NQ * (PA + PB);
  static_assert(BM % PIECE_ROWS == 0 && BN % PIECE_ROWS == 0, "tile vs workgroup");
  static_assert((STAGES - 2) * G <= 63, "vmcnt range");
  extern __shared__ __attribute__((aligned(16))) unsigned char smx[];

  const int ntiles = g.tiles_m * g.tiles_n;
  int lid = blockIdx.x;
  {
    const int q = ntiles >> 3, rem = ntiles & 7, xcd = lid & 7, within = lid >> 3;
    lid = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + within;
  }
  constexpr int GM = 8;
  const int per_group = GM * g.tiles_n;
  const int grp = lid / per_group;
  const int first_m = grp * GM;
  const int gsz = min(g.tiles_m - first_m, GM);
  const int tm = first_m + (lid % per_group) % gsz;
  const int tn = (lid % per_group) / gsz;
  const int m0 = tm * BM, n0 = tn * BN;

  const long long bz = blockIdx.z;
  const unsigned short* A = g.A + bz * g.sA;
  const unsigned short* B = g.B + bz * g.sB;
  float* C = g.C + bz * g.sC;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid / WN, wn = wid % WN;
  const int l15 = lane & 15, kg = lane >> 4;

  const int lrow = tid >> 1;
  const int kslot = (tid & 1) * 8;
  const unsigned short* srcA[PA];
  const unsigned short* srcB[PB];
#pragma unroll
  for (int p = 0; p < PA; ++p)
    srcA[p] = A + (long long)min(m0 + p * PIECE_ROWS + lrow, g.M - 1) * 16 + kslot;
#pragma unroll
  for (int p = 0; p < PB; ++p)
    srcB[p] = B + (long long)min(n0 + p * PIECE_ROWS + lrow, g.N - 1) * 16 + kslot;

  auto issue = [&](int t, int st) {
    unsigned char* base = smx + st * STAGE + wid * 1024;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int pl = q / KS, ks = q % KS;
      const long long ka = (long long)(t * KS + ks) * g.slabA, kb = (long long)(t * KS + ks) * g.slabB;
#pragma unroll
      for (int p = 0; p < PA; ++p)
        __builtin_amdgcn_global_load_lds(NAWS_GLB_PTR(srcA[p] + pl * g.planeA + ka),
                                         NAWS_LDS_PTR(base + q * A_PLANE + p * (NT * 16)), 16, 0, 0);
#pragma unroll
      for (int p = 0; p < PB; ++p)
        __builtin_amdgcn_global_load_lds(NAWS_GLB_PTR(srcB[p] + pl * g.planeB + kb),
                                         NAWS_LDS_PTR(base + NQ * A_PLANE + q * B_PLANE + p * (NT * 16)),
                                         16, 0, 0);
    }
  };

  f32x4 acc[TI][TJ];
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int j = 0; j < TJ; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.f;

  // lane -> (row l15, slab kg >> 1 of the pair, k-half kg & 1)
  const int rd_a = (wm * WTM + l15) * 32 + (kg & 1) * 16 + (kg >> 1) * A_PLANE;
  const int rd_b = NQ * A_PLANE + (wn * WTN + l15) * 32 + (kg & 1) * 16 + (kg >> 1) * B_PLANE;

  const int T = g.K / (16 * KS);
#pragma unroll
  for (int s = 0; s < STAGES - 1; ++s)
    if (s < T) issue(s, s);
  int st_cur = 0, st_fill = STAGES - 1;
  for (int t = 0; t < T; ++t) {
    if (t + STAGES - 2 < T) wait_vmcnt<(STAGES - 2) * G>();
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    if (t + STAGES - 1 < T) issue(t + STAGES - 1, st_fill);
    const unsigned char* st = smx + st_cur * STAGE;
#pragma unroll
    for (int kk = 0; kk < KS / 2; ++kk) {
      vec_t b[NPL][TJ];
#pragma unroll
      for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
        for (int j = 0; j < TJ; ++j)
          b[pl][j] = *reinterpret_cast<const vec_t*>(st + rd_b + (pl * KS + 2 * kk) * B_PLANE + j * 512);
#pragma unroll
      for (int ih = 0; ih < IH; ++ih) {
        vec_t a[NPL][TIH];
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
          for (int i = 0; i < TIH; ++i)
            a[pl][i] = *reinterpret_cast<const vec_t*>(st + rd_a + (pl * KS + 2 * kk) * A_PLANE +
                                                       (ih * TIH + i) * 512);
#define NAWS_M16_TERM(P, Q)                                                                      \
  _Pragma("unroll") for (int i = 0; i < TIH; ++i) _Pragma("unroll") for (int j = 0; j < TJ; ++j) \
      acc[ih * TIH + i][j] = mfma32(a[P][i], b[Q][j], acc[ih * TIH + i][j]);
        NAWS_M16_TERM(0, 0)
        if constexpr (NPL == 2) {
          NAWS_M16_TERM(0, 1)
          NAWS_M16_TERM(1, 0)
        }
#undef NAWS_M16_TERM
      }
    }
    st_cur = (st_cur + 1 == STAGES) ? 0 : st_cur + 1;
    st_fill = (st_fill + 1 == STAGES) ? 0 : st_fill + 1;
  }

  const float* bias = g.bias ? g.bias + bz * g.sBias : nullptr;
  const float* aux = g.aux ? g.aux + bz * g.sC : nullptr;
  const int epi = g.epilogue;
#pragma unroll
  for (int i = 0; i < TI; ++i) {
    const int row0 = m0 + wm * WTM + i * 16 + kg * 4;
    float rsv[4] = {1.f, 1.f, 1.f, 1.f};
    if constexpr (F16) {
      const float* rs = g.rs + bz * g.sRs;
#pragma unroll
      for (int e = 0; e < 4; ++e) rsv[e] = rs[min(row0 + e, g.M - 1)];
    }
#pragma unroll
    for (int j = 0; j < TJ; ++j) {
      const int col = n0 + wn * WTN + j * 16 + l15;
      if (col >= g.N) continue;
      const float bv = (bias && epi >= NAWS_EPI_BIAS && epi <= NAWS_EPI_BIAS_RELU_DROP) ? bias[col] : 0.f;
      float cscale = 1.f;
      if constexpr (F16) cscale = g.cs[bz * g.sCs + col];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int row = row0 + e;
        if (row >= g.M) continue;
        float v = acc[i][j][e];
        if constexpr (F16) v = v * rsv[e] * cscale;       // powers of two: exact, in this order
        v += bv;
        if (epi == NAWS_EPI_BIAS_RELU || epi == NAWS_EPI_BIAS_RELU_DROP) v = fmaxf(v, 0.f);
        if (epi == NAWS_EPI_BIAS_RELU_DROP) {
          const unsigned long long idx =
              (unsigned long long)bz * g.M * g.N + (unsigned long long)row * g.N + col;
          v = naws_keep(g.seed, idx, g.drop_thr) ? v * g.drop_scale : 0.f;
        } else if (epi == NAWS_EPI_GATE_POS) {
          v = (aux[row * g.ldaux + col] > 0.f) ? v * g.alpha : 0.f;
        }
        const int idx = row * g.ldc + col;
        if (g.accumulate) v += C[idx];
        C[idx] = v;
        acc[i][j][e] = v;
      }
    }
  }
  if (g.am.rowmax || g.am.colmax)
    naws_tile_amax_16<TI, TJ>(acc, m0 + wm * WTM, n0 + wn * WTN, g.M, g.N, lane, g.am, bz);
}

template <int BM, int BN, int WM, int WN, int STAGES, int NPL, int KS, bool F16>
int launch_x3_m16(XArgs& g, int batch, hipStream_t s) {
  g.tiles_m = (int)naws_cdiv(g.M, BM);
  g.tiles_n = (int)naws_cdiv(g.N, BN);
  const size_t lds = (size_t)STAGES * NPL * KS * (BM + BN) * 32;
  auto kern = gemm_x3_m16_kernel<BM, BN, WM, WN, STAGES, NPL, KS, F16>;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  dim3 grid((unsigned)(g.tiles_m * g.tiles_n), 1, batch);
  hipLaunchKernelGGL(kern, grid, dim3(64 * WM * WN), lds, s, g);
  return naws_check_launch();
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

// X fp32 [rows][ld] -> P[3][slabs][outer][16] through 64 x 64 LDS tiles.
//   TRANS == false: outer = rows, K = cols;  TRANS == true: outer = cols, K = rows.
// Each workgroup writes, per plane, 4 runs of 64 * 32 contiguous bytes.
template <bool TRANS, int NPL = 3>
__global__ __launch_bounds__(256) void split3_kernel(const float* __restrict__ X, int rows, int cols,
                                                     int ld, int outer, int slabs, long long sx,
                                                     long long plane, long long sp,
                                                     unsigned short* __restrict__ P) {
  __shared__ float tile[64][65];
  const float* Xb = X + blockIdx.z * sx;
  unsigned short* Pb = P + blockIdx.z * sp;
  const int c0 = blockIdx.x * 64, r0 = blockIdx.y * 64;
  const int tc = threadIdx.x & 63, tr = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int r = r0 + i * 4 + tr, c = c0 + tc;
    tile[i * 4 + tr][tc] = (r < rows && c < cols) ? Xb[(long long)r * ld + c] : 0.f;
  }
  __syncthreads();
  const int o0 = TRANS ? c0 : r0;          // first outer index of this tile
  const int k0 = TRANS ? r0 : c0;          // first K index (multiple of 64)
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int u = threadIdx.x + it * 256;  // (slab s, outer o, half hh), hh fastest
    const int hh = u & 1, o = (u >> 1) & 63, s = u >> 7;
    if (o0 + o >= outer || k0 / 16 + s >= slabs) continue;
    unsigned short q[3][8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int kk = s * 16 + hh * 8 + e;
      split3(TRANS ? tile[kk][o] : tile[o][kk], q[0][e], q[1][e], q[2][e]);
    }
    const long long dst = ((long long)(k0 / 16 + s) * outer + (o0 + o)) * 16 + hh * 8;
#pragma unroll
    for (int pl = 0; pl < NPL; ++pl) {
      u32x4 w;
      w.x = q[pl][0] | ((unsigned)q[pl][1] << 16);
      w.y = q[pl][2] | ((unsigned)q[pl][3] << 16);
      w.z = q[pl][4] | ((unsigned)q[pl][5] << 16);
      w.w = q[pl][6] | ((unsigned)q[pl][7] << 16);
      *reinterpret_cast<u32x4*>(Pb + pl * plane + dst) = w;
    }
  }
}

// ---- fp16x2: fp32 operands as two scaled f16 planes ---------------------------------------------
// Per outer index o (a row of the NT operand = one output row / column of the GEMM):
//      s_o  = 2^(14 - floor(log2 amax_o))              (amax_o * s_o in [2^14, 2^15))
//      hi   = f16(x * s_o),   lo = f16(x * s_o - hi)   (the residue is exact in fp32)
// |x*s - hi - lo| <= max(2^-22 |x*s|, 2^-25): 22+ significand bits for every element within
// 2^-18 of its row's maximum, an absolute floor of 2^-39 of the row maximum below that.  The GEMM
// keeps hi*hi + hi*lo + lo*hi (the dropped lo*lo is < 2^-22 |ab|), accumulates in fp32 and
// multiplies the accumulator by 1/(s_row * s_col) - exact.  Three MFMA passes instead of the six
// of the bf16 split, at an operand error far below the fp32 accumulation error of a K >= 1000
// dot product (profiles/r01_x3_accuracy.md has the measured comparison).
// amax_kernel: |x| maxima per outer index (non-negative floats order like their bit patterns).
constexpr int AMAX_CT = 16;
template <bool TRANS>
__global__ __launch_bounds__(256) void amax_kernel(const float* __restrict__ X, int rows, int cols,
                                                   int ld, int outer, long long sx,
                                                   unsigned* __restrict__ amax,
                                                   const float* __restrict__ rowmul) {
  __shared__ unsigned red[4][64];
  const float* Xb = X + blockIdx.z * sx;
  unsigned* out = amax + (long long)blockIdx.z * outer;
  const int c0 = blockIdx.x * 64, r0 = blockIdx.y * 64;
  const int tc = threadIdx.x & 63, tr = threadIdx.x >> 6;
  if (TRANS) {            // outer = columns: each thread folds 16 rows of its column
    float m = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int r = r0 + i * 4 + tr, c = c0 + tc;
      if (r < rows && c < cols)
        m = fmaxf(m, fabsf(Xb[(long long)r * ld + c] * (rowmul ? rowmul[r] : 1.f)));
    }
    red[tr][tc] = __float_as_uint(m);
    __syncthreads();
    if (tr == 0 && c0 + tc < cols) {
      const unsigned v = max(max(red[0][tc], red[1][tc]), max(red[2][tc], red[3][tc]));
      if (v) atomicMax(out + c0 + tc, v);
    }
  } else {                // outer = rows: blockIdx.x covers AMAX_CT column tiles; a thread folds its
                          // column of each, then the wave folds the 64 lanes (16 rows per wave)
    float m[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) m[i] = 0.f;
    for (int ct = 0; ct < AMAX_CT; ++ct) {
      const int c = (blockIdx.x * AMAX_CT + ct) * 64 + tc;
      if (c >= cols) break;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int r = r0 + i * 4 + tr;
        if (r < rows) m[i] = fmaxf(m[i], fabsf(Xb[(long long)r * ld + c] * (rowmul ? rowmul[r] : 1.f)));
      }
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      float v = m[i];
#pragma unroll
      for (int d = 32; d > 0; d >>= 1) v = fmaxf(v, __shfl_xor(v, d));
      const int r = r0 + i * 4 + tr;
      if (tc == 0 && r < rows && v > 0.f) atomicMax(out + r, __float_as_uint(v));
    }
  }
}

__device__ __forceinline__ void f16x2_scales(unsigned amax_bits, float& s, float& inv) {
  naws_f16x2_scales(amax_bits, s, inv);      // naws_common.h
}

template <bool TRANS>
__global__ __launch_bounds__(256) void split2h_kernel(const float* __restrict__ X, int rows, int cols,
                                                      int ld, int outer, int slabs, long long sx,
                                                      long long plane, long long sp,
                                                      const unsigned* __restrict__ amax,
                                                      float* __restrict__ inv_scale,
                                                      unsigned short* __restrict__ P,
                                                      const float* __restrict__ rowmul) {
  __shared__ float tile[64][65];
  const float* Xb = X + blockIdx.z * sx;
  unsigned short* Pb = P + blockIdx.z * sp;
  const unsigned* am = amax + (long long)blockIdx.z * outer;
  float* inv = inv_scale + (long long)blockIdx.z * outer;
  const int c0 = blockIdx.x * 64, r0 = blockIdx.y * 64;
  const int tc = threadIdx.x & 63, tr = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int r = r0 + i * 4 + tr, c = c0 + tc;
    tile[i * 4 + tr][tc] = (r < rows && c < cols) ? Xb[(long long)r * ld + c] * (rowmul ? rowmul[r] : 1.f) : 0.f;
  }
  __syncthreads();
  const int o0 = TRANS ? c0 : r0;
  const int k0 = TRANS ? r0 : c0;
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int u = threadIdx.x + it * 256;
    const int hh = u & 1, o = (u >> 1) & 63, s = u >> 7;
    if (o0 + o >= outer || k0 / 16 + s >= slabs) continue;
    float sc, isc;
    f16x2_scales(am[o0 + o], sc, isc);
    if (k0 == 0 && s == 0 && hh == 0) inv[o0 + o] = isc;
    unsigned short q[2][8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int kk = s * 16 + hh * 8 + e;
      const float t = (TRANS ? tile[kk][o] : tile[o][kk]) * sc;
      const _Float16 hi = (_Float16)t;
      float r = t - (float)hi;
      if (!(fabsf(t) <= 65504.f)) r = 0.f;             // NaN / overflow live in the hi plane only
      const _Float16 lo = (_Float16)r;
      q[0][e] = *reinterpret_cast<const unsigned short*>(&hi);
      q[1][e] = *reinterpret_cast<const unsigned short*>(&lo);
    }
    const long long dst = ((long long)(k0 / 16 + s) * outer + (o0 + o)) * 16 + hh * 8;
#pragma unroll
    for (int pl = 0; pl < 2; ++pl) {
      u32x4 w;
      w.x = q[pl][0] | ((unsigned)q[pl][1] << 16);
      w.y = q[pl][2] | ((unsigned)q[pl][3] << 16);
      w.z = q[pl][4] | ((unsigned)q[pl][5] << 16);
      w.w = q[pl][6] | ((unsigned)q[pl][7] << 16);
      *reinterpret_cast<u32x4*>(Pb + pl * plane + dst) = w;
    }
  }
}

// One pass over X writing BOTH operand forms of the fp16x2 GEMM from maxima that the producer of
// X has already reported (NawsAmax): the row-scaled planes Pn[2][batch][kn/16][rows][16] (X as an NT
// operand, K = cols) and / or the column-scaled transposed planes Pt[2][batch][kt/16][cols][16]
// (X^T as an NT operand, K = rows; of diag(rowmul) X when rowmul is given).  Replaces
// amax + split + amax<T> + split<T> (4 reads of X) by one read.
__device__ __forceinline__ void split2h_pack(const float (&t)[8], float sc, u32x4& hi4, u32x4& lo4) {
  unsigned short q[2][8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float v = t[e] * sc;
    const _Float16 hi = (_Float16)v;
    float r = v - (float)hi;
    if (!(fabsf(v) <= 65504.f)) r = 0.f;             // NaN / overflow live in the hi plane only
    const _Float16 lo = (_Float16)r;
    q[0][e] = *reinterpret_cast<const unsigned short*>(&hi);
    q[1][e] = *reinterpret_cast<const unsigned short*>(&lo);
  }
  hi4.x = q[0][0] | ((unsigned)q[0][1] << 16); hi4.y = q[0][2] | ((unsigned)q[0][3] << 16);
  hi4.z = q[0][4] | ((unsigned)q[0][5] << 16); hi4.w = q[0][6] | ((unsigned)q[0][7] << 16);
  lo4.x = q[1][0] | ((unsigned)q[1][1] << 16); lo4.y = q[1][2] | ((unsigned)q[1][3] << 16);
  lo4.z = q[1][4] | ((unsigned)q[1][5] << 16); lo4.w = q[1][6] | ((unsigned)q[1][7] << 16);
}

__global__ __launch_bounds__(256) void split2h_dual_kernel(
    const float* __restrict__ X, int rows, int cols, int ld, long long sx,
    const unsigned* __restrict__ rowmax, const unsigned* __restrict__ colmax,
    const float* __restrict__ rowmul, unsigned short* __restrict__ Pn, float* __restrict__ inv_n,
    int slabs_n, unsigned short* __restrict__ Pt, float* __restrict__ inv_t, int slabs_t, int batch) {
  __shared__ float tile[64][65];
  const int bz = blockIdx.z;
  const float* Xb = X + bz * sx;
  const int c0 = blockIdx.x * 64, r0 = blockIdx.y * 64;
  const int tc = threadIdx.x & 63, tr = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int r = r0 + i * 4 + tr, c = c0 + tc;
    tile[i * 4 + tr][tc] = (r < rows && c < cols) ? Xb[(long long)r * ld + c] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int u = threadIdx.x + it * 256;           // (slab s, outer o, half hh), hh fastest
    const int hh = u & 1, o = (u >> 1) & 63, s = u >> 7;
    if (Pn && r0 + o < rows && c0 / 16 + s < slabs_n) {
      float sc, isc;
      f16x2_scales(rowmax[(long long)bz * rows + r0 + o], sc, isc);
      if (c0 == 0 && s == 0 && hh == 0) inv_n[(long long)bz * rows + r0 + o] = isc;
      float t[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) t[e] = tile[o][s * 16 + hh * 8 + e];
      u32x4 hi4, lo4;
      split2h_pack(t, sc, hi4, lo4);
      const long long sp = (long long)slabs_n * 16 * rows, plane = (long long)batch * sp;
      const long long dst = bz * sp + ((long long)(c0 / 16 + s) * rows + (r0 + o)) * 16 + hh * 8;
      *reinterpret_cast<u32x4*>(Pn + dst) = hi4;
      *reinterpret_cast<u32x4*>(Pn + plane + dst) = lo4;
    }
    if (Pt && c0 + o < cols && r0 / 16 + s < slabs_t) {
      float sc, isc;
      f16x2_scales(colmax[(long long)bz * cols + c0 + o], sc, isc);
      if (r0 == 0 && s == 0 && hh == 0) inv_t[(long long)bz * cols + c0 + o] = isc;
      float t[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int kk = s * 16 + hh * 8 + e;
        t[e] = tile[kk][o] * ((rowmul && r0 + kk < rows) ? rowmul[r0 + kk] : 1.f);
      }
      u32x4 hi4, lo4;
      split2h_pack(t, sc, hi4, lo4);
      const long long sp = (long long)slabs_t * 16 * cols, plane = (long long)batch * sp;
      const long long dst = bz * sp + ((long long)(r0 / 16 + s) * cols + (c0 + o)) * 16 + hh * 8;
      *reinterpret_cast<u32x4*>(Pt + dst) = hi4;
      *reinterpret_cast<u32x4*>(Pt + plane + dst) = lo4;
    }
  }
}

// ---- 3x3 convolution (NHWC fp32 activations) as an fp32x3 implicit GEMM -------------------------
// A = activations gathered straight from the fp32 NHWC tensor (branch-free buffer loads, halo ->
// 0), split into the three bf16 planes in registers (8 VALU ops per element, < 5 % of the MFMA
// time of a step) and written into the same swizzled LDS image the DMA path uses; B = weight
// planes [3][9*Cin/16][Cout][16] (naws_split_bf16x3 of the packed [Cout][3][3][Cin] weight) by
// LDS-DMA.  A K-step is 16 channels of one tap.  Two LDS stages: step t+1's activations are
// loaded at the top of step t and split/written after its MFMAs.
struct CArgs {
  const float* X;            // NHWC
  const unsigned short* B;   // weight planes
  const float* bias;
  float* Y;                  // NHWC
  int M, Cout, Cin, H, W, dil, relu;
  long long planeB, slabB;
  unsigned bytesX;
  int tiles_m, tiles_n;
  // fp16x2 form of the halo kernel
  const float* scaleB;       // 1/scale per output channel (naws_split_f16x2 of the weight)
  const unsigned* amax_in;   // bit pattern of an upper bound b of max|X| ...
  float in_mul, in_add;      // ... the bound used is b * in_mul + in_add
  unsigned* amax_out;        // receives the bit pattern of max|Y| (nullable)
  int pool;                  // 1: write maxpool2x2/stride 2 of the output instead of the output
};

template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(64 * WM * WN, (WM * WN >= 8 ? 1 : 2)) void conv_x3_kernel(CArgs g) {
  constexpr int NT = 64 * WM * WN, NW = WM * WN;
  constexpr int WTM = BM / WM, WTN = BN / WN;
  constexpr int TI = WTM / 32, TJ = WTN / 32;
  constexpr int A_PLANE = BM * 32, B_PLANE = BN * 32;
  constexpr int STAGE = 3 * (A_PLANE + B_PLANE);
  constexpr int UA = BM * 2 / NT;                 // (row, 8-channel half) units per thread
  constexpr int BPIECES = 3 * BN / 32;            // 1 KB DMA pieces of the weight stage
  static_assert(BM * 2 % NT == 0, "tile vs workgroup");
  extern __shared__ __attribute__((aligned(16))) unsigned char smx[];

  const int ntiles = g.tiles_m * g.tiles_n;
  int lid = blockIdx.x;
  {
    const int q = ntiles >> 3, rem = ntiles & 7, xcd = lid & 7, within = lid >> 3;
    lid = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + within;
  }
  // all Cout tiles of one pixel tile are neighbours: the gathered activations are shared in L2
  const int tm = lid / g.tiles_n, tn = lid % g.tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid / WN, wn = wid % WN;
  const int l31 = lane & 31, h = lane >> 5;

  const __amdgpu_buffer_rsrc_t rsX =
      __builtin_amdgcn_make_buffer_rsrc((void*)g.X, 0, (int)g.bytesX, 0x00020000);
  constexpr unsigned OOB = 0xFFFFFFF0u;
  int py[UA], px[UA];
  unsigned abase[UA];
  int awr[UA];
#pragma unroll
  for (int i = 0; i < UA; ++i) {
    const int u = tid + i * NT;
    const int row = u >> 1, half = u & 1;
    const int gm = m0 + row;
    if (gm < g.M) {
      px[i] = gm % g.W; py[i] = (gm / g.W) % g.H;
      abase[i] = ((unsigned)gm * (unsigned)g.Cin + half * 8) * 4u;
    } else { px[i] = 0; py[i] = 0; abase[i] = OOB; }
    awr[i] = row * 32 + ((half ^ ((row >> 3) & 1)) * 16);
  }
  const int bslot = ((lane & 1) ^ ((lane >> 4) & 1)) * 8;

  u32x4 ra[UA][2];
  auto loadA = [&](int t) {
    const int k0 = t * 16;
    const int tap = k0 / g.Cin, c0 = k0 - tap * g.Cin;
    const int dy = (tap / 3 - 1) * g.dil, dx = (tap % 3 - 1) * g.dil;
    const int delta = ((dy * g.W + dx) * g.Cin + c0) * 4;
#pragma unroll
    for (int i = 0; i < UA; ++i) {
      const int yy = py[i] + dy, xx = px[i] + dx;
      const bool ok = (abase[i] != OOB) && yy >= 0 && yy < g.H && xx >= 0 && xx < g.W;
      const unsigned off = ok ? (unsigned)((int)abase[i] + delta) : OOB;
      ra[i][0] = __builtin_amdgcn_raw_buffer_load_b128(rsX, (int)off, 0, 0);
      ra[i][1] = __builtin_amdgcn_raw_buffer_load_b128(rsX, ok ? (int)(off + 16) : (int)OOB, 0, 0);
    }
  };
  auto storeA = [&](int st) {
    unsigned char* base = smx + st * STAGE;
#pragma unroll
    for (int i = 0; i < UA; ++i) {
      unsigned short q[3][8];
      const unsigned w[8] = {ra[i][0].x, ra[i][0].y, ra[i][0].z, ra[i][0].w,
                             ra[i][1].x, ra[i][1].y, ra[i][1].z, ra[i][1].w};
#pragma unroll
      for (int e = 0; e < 8; ++e) split3(__uint_as_float(w[e]), q[0][e], q[1][e], q[2][e]);
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) {
        u32x4 v;
        v.x = q[pl][0] | ((unsigned)q[pl][1] << 16);
        v.y = q[pl][2] | ((unsigned)q[pl][3] << 16);
        v.z = q[pl][4] | ((unsigned)q[pl][5] << 16);
        v.w = q[pl][6] | ((unsigned)q[pl][7] << 16);
        *reinterpret_cast<u32x4*>(base + pl * A_PLANE + awr[i]) = v;
      }
    }
  };
  auto issueB = [&](int t, int st) {
    unsigned char* base = smx + st * STAGE + 3 * A_PLANE;
#pragma unroll
    for (int r = 0; r < (BPIECES + NW - 1) / NW; ++r) {
      const int piece = wid + r * NW;             // wave-uniform
      if (piece < BPIECES) {
        const int pl = piece / (BN / 32), rb = piece % (BN / 32);
        // weight rows past Cout re-read the last row (their columns are never stored)
        const int wrow = min(n0 + rb * 32 + (lane >> 1), g.Cout - 1);
        __builtin_amdgcn_global_load_lds(
            NAWS_GLB_PTR(g.B + pl * g.planeB + t * g.slabB + (long long)wrow * 16 + bslot),
            NAWS_LDS_PTR(base + pl * B_PLANE + rb * 1024), 16, 0, 0);
      }
    }
  };

  f32x16 acc[TI][TJ];
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int j = 0; j < TJ; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int rd_a = (wm * WTM + l31) * 32 + ((h ^ ((l31 >> 3) & 1)) * 16);
  const int rd_b = 3 * A_PLANE + (wn * WTN + l31) * 32 + ((h ^ ((l31 >> 3) & 1)) * 16);

  const int T = 9 * g.Cin / 16;
  issueB(0, 0);
  loadA(0);
  storeA(0);
  for (int t = 0; t < T; ++t) {
    __syncthreads();                     // stage t&1 complete (DMA drained + LDS writes visible)
    const bool more = t + 1 < T;
    if (more) { issueB(t + 1, (t + 1) & 1); loadA(t + 1); }
    const unsigned char* st = smx + (t & 1) * STAGE;
    bf16x8 a[3][TI], b[3][TJ];
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
      for (int i = 0; i < TI; ++i)
        a[pl][i] = *reinterpret_cast<const bf16x8*>(st + rd_a + pl * A_PLANE + i * 1024);
#pragma unroll
      for (int j = 0; j < TJ; ++j)
        b[pl][j] = *reinterpret_cast<const bf16x8*>(st + rd_b + pl * B_PLANE + j * 1024);
    }
#define NAWS_X3_TERM(P, Q)                                                                      \
  _Pragma("unroll") for (int i = 0; i < TI; ++i) _Pragma("unroll") for (int j = 0; j < TJ; ++j) \
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[P][i], b[Q][j], acc[i][j], 0, 0, 0);
    NAWS_X3_TERM(0, 0)
    NAWS_X3_TERM(0, 1)
    NAWS_X3_TERM(1, 0)
    NAWS_X3_TERM(1, 1)
    NAWS_X3_TERM(0, 2)
    NAWS_X3_TERM(2, 0)
#undef NAWS_X3_TERM
    if (more) storeA((t + 1) & 1);
  }

#pragma unroll
  for (int j = 0; j < TJ; ++j) {
    const int col = n0 + wn * WTN + j * 32 + l31;
    if (col >= g.Cout) continue;
    const float bv = g.bias ? g.bias[col] : 0.f;
#pragma unroll
    for (int i = 0; i < TI; ++i) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = m0 + wm * WTM + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
        if (row >= g.M) continue;
        float v = acc[i][j][e] + bv;
        if (g.relu) v = fmaxf(v, 0.f);
        g.Y[(long long)row * g.Cout + col] = v;
      }
    }
  }
}

template <int BM, int BN, int WM, int WN>
int launch_conv_x3(CArgs& g, hipStream_t s) {
  g.tiles_m = (int)naws_cdiv(g.M, BM);
  g.tiles_n = (int)naws_cdiv(g.Cout, BN);
  const size_t lds = (size_t)2 * 3 * (BM + BN) * 32;
  auto kern = conv_x3_kernel<BM, BN, WM, WN>;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)(g.tiles_m * g.tiles_n)), dim3(64 * WM * WN), lds, s, g);
  return naws_check_launch();
}


// Epilogue of the halo-tile kernels: un-scale, bias, ReLU, optional fused 2x2 max pool, store, and
// the workgroup's max|Y| for the next layer's operand scale.  acc[i][j]: tile row 2 * wid + i,
// channel block j (32x32x16 MFMA accumulator layout).
template <int BN, bool F16>
__device__ __forceinline__ void halo_epilogue(const CArgs& g, f32x16 (&acc)[2][BN / 32], int img,
                                              int ty0, int tx0, int n0, int wid, int lane,
                                              float iscA, unsigned char* smx) {
  constexpr int TJ = BN / 32, TI = 2;
  const int tid = threadIdx.x, l31 = lane & 31, h = lane >> 5;
  float vmax = 0.f;
  bool pooled = false;
  if constexpr (F16) pooled = g.pool != 0;
  if (pooled) {
    // max-pool 2x2 / stride 2 fused: a lane's accumulators hold the four pixels of a window
    // (rows 2*wid, 2*wid + 1 of the tile = i; register pairs (e, e+1) = adjacent columns), and
    // max commutes exactly with the monotone epilogue (x * 2^k + b, ReLU)
    const int Ho = g.H / 2, Wo = g.W / 2;
    const int yo = (ty0 + 2 * wid) >> 1;
#pragma unroll
    for (int j = 0; j < TJ; ++j) {
      const int col = n0 + j * 32 + l31;
      if (col >= g.Cout) continue;
      const float bv = g.bias ? g.bias[col] : 0.f;
      const float un = iscA * g.scaleB[col];
#pragma unroll
      for (int e = 0; e < 16; e += 2) {
        const int xo = (tx0 + (e & 3) + 8 * (e >> 2) + 4 * h) >> 1;
        if (yo >= Ho || xo >= Wo) continue;
        float v = -3.4028234e38f;
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
          for (int d = 0; d < 2; ++d) {
            float t = acc[i][j][e + d] * un + bv;
            if (g.relu) t = fmaxf(t, 0.f);
            v = fmaxf(v, t);
          }
        g.Y[((long long)(img * Ho + yo) * Wo + xo) * g.Cout + col] = v;
        vmax = fmaxf(vmax, fabsf(v));
      }
    }
  } else {
#pragma unroll
  for (int j = 0; j < TJ; ++j) {
    const int col = n0 + j * 32 + l31;
    if (col >= g.Cout) continue;
    const float bv = g.bias ? g.bias[col] : 0.f;
    float un = 1.f;
    if constexpr (F16) un = iscA * g.scaleB[col];        // powers of two: exact
#pragma unroll
    for (int i = 0; i < TI; ++i) {
      const int y = ty0 + 2 * wid + i;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int x = tx0 + (e & 3) + 8 * (e >> 2) + 4 * h;
        if (y >= g.H || x >= g.W) continue;
        float v = acc[i][j][e];
        if constexpr (F16) v *= un;
        v += bv;
        if (g.relu) v = fmaxf(v, 0.f);
        g.Y[((long long)(img * g.H + y) * g.W + x) * g.Cout + col] = v;
        if constexpr (F16) vmax = fmaxf(vmax, fabsf(v));
      }
    }
  }
  }
  if constexpr (F16) {
    if (g.amax_out) {
      float* red = reinterpret_cast<float*>(smx);     // the operand stages are no longer read
#pragma unroll
      for (int d = 32; d > 0; d >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, d));
      __syncthreads();
      if (lane == 0) red[wid] = vmax;
      __syncthreads();
      if (tid == 0) {
        const unsigned v = __float_as_uint(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])));
        if (v > __hip_atomic_load(g.amax_out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
          atomicMax(g.amax_out, v);
      }
    }
  }
}

// ---- 3x3 conv, fp32x3, with the input halo tile staged once per channel slab -------------------
// The linear-pixel kernel above gathers every tap's activations again: 9 x 64 B per output pixel
// per 16-channel slab.  For the wide shallow layers (Cout <= 128: conv1_2, conv2_1, conv2_2) that
// gather, not the MFMA, sets the pace.  Here a workgroup owns an 8-row x 32-column pixel tile:
// for each 16-channel slab the (8+2) x (32+2) halo is gathered, split and written to LDS ONCE
// (340 pixels instead of 9 x 256), and the nine taps read their A fragments from it at shifted
// rows - a wave's 32-lane fragment is one image row of the tile, so a tap is just a row offset
// (dy * 34 + dx) into the halo image.  Same LDS row format and bank swizzle as everywhere else;
// weights per (tap, slab) by LDS-DMA, double buffered.  dilation 1, stride 1.
// F16: the fp16x2 form - activations scaled by one power of two per tensor (from an upper bound
// of max|X| handed in by the producer of X), split into f16 hi / lo planes in registers; weight
// planes from naws_split_f16x2 (per-output-channel scales); 3 MFMA terms instead of 6; the
// accumulator is un-scaled in the epilogue, which also reports max|Y| for the next layer.
template <int BN, bool F16 = false, int DIL = 1>
__global__ __launch_bounds__(256, (BN <= 64 ? 3 : 2)) void conv_x3_halo_kernel(CArgs g) {
  constexpr int NPL = F16 ? 2 : 3;
  typedef typename OperandVec<F16>::type vec_t;
  // BN = 64: one halo stage (refilled behind an extra barrier every 9th step) keeps the
  // workgroup at 45 KB of LDS, so three of them share a CU; BN = 128: two halo stages
  constexpr int ASTAGES = BN <= 64 ? 1 : 2;
  // DIL = 2 (conv5_x): the taps sit 2 pixels apart, the halo is (8+4) x (32+4) = 432 pixels
  constexpr int TH = 8, TW = 32, HWD = TW + 2 * DIL, HPIX = (TH + 2 * DIL) * HWD;   // 340 halo pixels
  constexpr int A_ROWS = (HPIX + 7) / 8 * 8;
  constexpr int A_PLANE = A_ROWS * 32, A_STAGE = NPL * A_PLANE;
  constexpr int B_PLANE = BN * 32, B_STAGE = NPL * B_PLANE;
  constexpr int TJ = BN / 32, TI = 2;
  constexpr int UR = (HPIX * 2 + 255) / 256;                                // staging rounds (3)
  constexpr int BPIECES = NPL * BN / 32;
  extern __shared__ __attribute__((aligned(16))) unsigned char smx[];
  unsigned char* smA = smx;
  unsigned char* smB = smx + ASTAGES * A_STAGE;

  const int tiles_x = (g.W + TW - 1) / TW, tiles_y = (g.H + TH - 1) / TH;
  int lid = blockIdx.x;
  const int tn = lid % g.tiles_n;
  lid /= g.tiles_n;
  const int tx0 = (lid % tiles_x) * TW;
  const int ty0 = ((lid / tiles_x) % tiles_y) * TH;
  const int img = lid / (tiles_x * tiles_y);
  const int n0 = tn * BN;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, h = lane >> 5;
  const __amdgpu_buffer_rsrc_t rsX =
      __builtin_amdgcn_make_buffer_rsrc((void*)g.X, 0, (int)g.bytesX, 0x00020000);
  constexpr unsigned OOB = 0xFFFFFFF0u;
  float scA = 1.f, iscA = 1.f;
  if constexpr (F16) {
    const float bound = __uint_as_float(*g.amax_in) * g.in_mul + g.in_add;
    f16x2_scales(__float_as_uint(bound), scA, iscA);
  }

  unsigned abase[UR];
  int awr[UR];
#pragma unroll
  for (int r = 0; r < UR; ++r) {
    const int u = tid + r * 256;
    const int hp = u >> 1, half = u & 1;
    const int y = ty0 - DIL + hp / HWD, x = tx0 - DIL + hp % HWD;
    const bool ok = hp < HPIX && y >= 0 && y < g.H && x >= 0 && x < g.W;
    abase[r] = ok ? ((unsigned)((img * g.H + y) * g.W + x) * (unsigned)g.Cin + half * 8) * 4u : OOB;
    awr[r] = hp < HPIX ? hp * 32 + ((half ^ ((hp >> 3) & 1)) * 16) : -1;
  }
  u32x4 ra[UR][2];
  auto loadA = [&](int slab) {
#pragma unroll
    for (int r = 0; r < UR; ++r) {
      const unsigned off = abase[r] != OOB ? abase[r] + (unsigned)slab * 64u : OOB;
      ra[r][0] = __builtin_amdgcn_raw_buffer_load_b128(rsX, (int)off, 0, 0);
      ra[r][1] = __builtin_amdgcn_raw_buffer_load_b128(rsX, abase[r] != OOB ? (int)(off + 16) : (int)OOB, 0, 0);
    }
  };
  auto storeA = [&](int st) {
    unsigned char* base = smA + st * A_STAGE;
#pragma unroll
    for (int r = 0; r < UR; ++r) {
      if (awr[r] < 0) continue;
      unsigned short q[3][8];
      const unsigned w[8] = {ra[r][0].x, ra[r][0].y, ra[r][0].z, ra[r][0].w,
                             ra[r][1].x, ra[r][1].y, ra[r][1].z, ra[r][1].w};
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        if constexpr (F16) {
          const float t = __uint_as_float(w[e]) * scA;
          const _Float16 hi = (_Float16)t;
          float rr = t - (float)hi;
          if (!(fabsf(t) <= 65504.f)) rr = 0.f;
          const _Float16 lo = (_Float16)rr;
          q[0][e] = *reinterpret_cast<const unsigned short*>(&hi);
          q[1][e] = *reinterpret_cast<const unsigned short*>(&lo);
        } else {
          split3(__uint_as_float(w[e]), q[0][e], q[1][e], q[2][e]);
        }
      }
#pragma unroll
      for (int pl = 0; pl < NPL; ++pl) {
        u32x4 v;
        v.x = q[pl][0] | ((unsigned)q[pl][1] << 16);
        v.y = q[pl][2] | ((unsigned)q[pl][3] << 16);
        v.z = q[pl][4] | ((unsigned)q[pl][5] << 16);
        v.w = q[pl][6] | ((unsigned)q[pl][7] << 16);
        *reinterpret_cast<u32x4*>(base + pl * A_PLANE + awr[r]) = v;
      }
    }
  };
  const int bslot = ((lane & 1) ^ ((lane >> 4) & 1)) * 8;
  auto issueB = [&](int kslab, int st) {
    unsigned char* base = smB + st * B_STAGE;
#pragma unroll
    for (int r = 0; r < (BPIECES + 3) / 4; ++r) {
      const int piece = wid + r * 4;
      if (piece < BPIECES) {
        const int pl = piece / (BN / 32), rb = piece % (BN / 32);
        const int wrow = min(n0 + rb * 32 + (lane >> 1), g.Cout - 1);
        __builtin_amdgcn_global_load_lds(
            NAWS_GLB_PTR(g.B + pl * g.planeB + (long long)kslab * g.slabB + (long long)wrow * 16 + bslot),
            NAWS_LDS_PTR(base + pl * B_PLANE + rb * 1024), 16, 0, 0);
      }
    }
  };

  f32x16 acc[TI][TJ];
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int j = 0; j < TJ; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int S = g.Cin / 16, T = 9 * S;
  const int rd_b = l31 * 32 + ((h ^ ((l31 >> 3) & 1)) * 16);
  issueB(0, 0);                  // (tap 0, slab 0)
  loadA(0);
  storeA(0);
  for (int kk = 0; kk < T; ++kk) {
    const int slab = kk / 9, tap = kk - slab * 9;
    __syncthreads();             // halo stage + weight stage of this step are complete
    if (kk + 1 < T) {
      const int s1 = (kk + 1) / 9, t1 = (kk + 1) - s1 * 9;
      issueB(t1 * S + s1, (kk + 1) & 1);
    }
    if (tap == 0 && slab + 1 < S) loadA(slab + 1);
    const unsigned char* sa = smA + (slab & (ASTAGES - 1)) * A_STAGE;
    const unsigned char* sb = smB + (kk & 1) * B_STAGE;
    const int dy = tap / 3 - 1, dx = tap % 3 - 1;
    vec_t a[NPL][TI], b[NPL][TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i) {
      const int hp = (2 * wid + i + DIL + dy * DIL) * HWD + (l31 + DIL + dx * DIL);
      const int off = hp * 32 + ((h ^ ((hp >> 3) & 1)) * 16);
#pragma unroll
      for (int pl = 0; pl < NPL; ++pl)
        a[pl][i] = *reinterpret_cast<const vec_t*>(sa + pl * A_PLANE + off);
    }
#pragma unroll
    for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
      for (int j = 0; j < TJ; ++j)
        b[pl][j] = *reinterpret_cast<const vec_t*>(sb + pl * B_PLANE + rd_b + j * 1024);
#define NAWS_X3_TERM(P, Q)                                                                      \
  _Pragma("unroll") for (int i = 0; i < TI; ++i) _Pragma("unroll") for (int j = 0; j < TJ; ++j) \
      acc[i][j] = mfma16(a[P][i], b[Q][j], acc[i][j]);
    NAWS_X3_TERM(0, 0)
    NAWS_X3_TERM(0, 1)
    NAWS_X3_TERM(1, 0)
    if constexpr (!F16) {
      NAWS_X3_TERM(1, 1)
      NAWS_X3_TERM(0, 2)
      NAWS_X3_TERM(2, 0)
    }
#undef NAWS_X3_TERM
    if (tap == 8 && slab + 1 < S) {
      if (ASTAGES == 1) __syncthreads();   // every wave is done with the only halo stage
      storeA((slab + 1) & (ASTAGES - 1));
    }
  }

  halo_epilogue<BN, F16>(g, acc, img, ty0, tx0, n0, wid, lane, iscA, smx);
}

template <int BN, bool F16 = false, int DIL = 1>
int launch_conv_x3_halo(CArgs& g, int N, hipStream_t s) {
  g.tiles_n = (int)naws_cdiv(g.Cout, BN);
  const long long tiles = (long long)N * naws_cdiv(g.H, 8) * naws_cdiv(g.W, 32) * g.tiles_n;
  if (tiles > 0x7fffffffLL) return NAWS_ERR_UNSUPPORTED;
  constexpr int A_ROWS = ((8 + 2 * DIL) * (32 + 2 * DIL) + 7) / 8 * 8;
  constexpr int NPL = F16 ? 2 : 3;
  const size_t lds = (size_t)(BN <= 64 ? 1 : 2) * NPL * A_ROWS * 32 + (size_t)2 * NPL * BN * 32;
  auto kern = conv_x3_halo_kernel<BN, F16, DIL>;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(256), lds, s, g);
  return naws_check_launch();
}

}  // namespace

template <int NPL>
static int split_planes(const float* X, int batch, int rows, int cols, int ld, int64_t strideX,
                        int transpose, int kpad, int kmult, void* P, void* stream) {
  if (batch <= 0 || rows <= 0 || cols <= 0 || ld < cols) return NAWS_ERR_SHAPE;
  NAWS_REQUIRE_PTR(X); NAWS_REQUIRE_PTR(P);
  const int kdim = transpose ? rows : cols;
  if (kpad != (kdim + kmult - 1) / kmult * kmult) return NAWS_ERR_ARG;
  if (((uintptr_t)P & 15) != 0) return NAWS_ERR_ARG;
  const int outer = transpose ? cols : rows;
  const long long sp = (long long)kpad * outer;              // one batch item of one plane
  const long long plane = (long long)batch * sp;             // P[NPL][batch][kpad/16][outer][16]
  const long long gy = naws_cdiv(transpose ? kpad : rows, 64);
  if (batch > 65535 || gy > 65535) return NAWS_ERR_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  if (!transpose) {
    dim3 grid((unsigned)naws_cdiv(kpad, 64), (unsigned)gy, batch);
    hipLaunchKernelGGL((split3_kernel<false, NPL>), grid, dim3(256), 0, s, X, rows, cols, ld, outer,
                       kpad / 16, (long long)strideX, plane, sp, (unsigned short*)P);
  } else {
    dim3 grid((unsigned)naws_cdiv(cols, 64), (unsigned)gy, batch);
    hipLaunchKernelGGL((split3_kernel<true, NPL>), grid, dim3(256), 0, s, X, rows, cols, ld, outer,
                       kpad / 16, (long long)strideX, plane, sp, (unsigned short*)P);
  }
  return naws_check_launch();
}

extern "C" int naws_split_bf16x3(const float* X, int batch, int rows, int cols, int ld,
                                 int64_t strideX, int transpose, int kpad, void* P, void* stream) {
  return split_planes<3>(X, batch, rows, cols, ld, strideX, transpose, kpad, 16, P, stream);
}

extern "C" int naws_to_bf16_slab(const float* X, int batch, int rows, int cols, int ld,
                                 int64_t strideX, int transpose, int kpad, void* P, void* stream) {
  return split_planes<1>(X, batch, rows, cols, ld, strideX, transpose, kpad, 64, P, stream);
}


extern "C" int naws_gemm_f32x3_nt(int M, int N, int K, const void* A3, int64_t slabA,
                                  int64_t planeA, const void* B3, int64_t slabB, int64_t planeB,
                                  float* C, int ldc, int batch, int64_t strideA, int64_t strideB,
                                  int64_t strideC, int epilogue, const float* bias,
                                  int64_t strideBias, const float* aux, int ldaux, float alpha,
                                  float drop_ratio, uint64_t seed, int accumulate, void* stream) {
  if (M <= 0 || N <= 0 || K <= 0 || batch <= 0) return NAWS_ERR_SHAPE;
  NAWS_REQUIRE_PTR(A3); NAWS_REQUIRE_PTR(B3); NAWS_REQUIRE_PTR(C);
  if (epilogue < NAWS_EPI_NONE || epilogue > NAWS_EPI_GATE_POS) return NAWS_ERR_ARG;
  if (epilogue == NAWS_EPI_GATE_POS && aux == nullptr) return NAWS_ERR_NULL;
  if (epilogue == NAWS_EPI_BIAS_RELU_DROP && !(drop_ratio >= 0.f && drop_ratio < 1.f)) return NAWS_ERR_ARG;
  if (slabA < (int64_t)M * 16 || slabB < (int64_t)N * 16 || ldc < N) return NAWS_ERR_SHAPE;
  if (K % 16 != 0 || slabA % 8 != 0 || slabB % 8 != 0 || planeA % 8 != 0 || planeB % 8 != 0 ||
      strideA % 8 != 0 || strideB % 8 != 0)
    return NAWS_ERR_ARG;
  if ((((uintptr_t)A3 | (uintptr_t)B3) & 15) != 0) return NAWS_ERR_ARG;
  if (batch > 65535) return NAWS_ERR_UNSUPPORTED;
  if ((long long)(M - 1) * ldc + N > 0x7fffffffLL ||
      (aux && (long long)(M - 1) * ldaux + N > 0x7fffffffLL))
    return NAWS_ERR_UNSUPPORTED;
  XArgs g{};
  g.A = (const unsigned short*)A3; g.B = (const unsigned short*)B3; g.C = C;
  g.M = M; g.N = N; g.K = K; g.ldc = ldc;
  g.planeA = planeA; g.planeB = planeB; g.slabA = slabA; g.slabB = slabB;
  g.sA = strideA; g.sB = strideB; g.sC = strideC; g.sBias = strideBias;
  g.bias = bias; g.aux = aux; g.ldaux = ldaux; g.alpha = alpha;
  g.drop_thr = naws_drop_threshold(drop_ratio);
  g.drop_scale = (float)(1.0 / (1.0 - (double)drop_ratio));
  g.seed = seed; g.epilogue = epilogue; g.accumulate = accumulate;
  hipStream_t s = (hipStream_t)stream;
  if (g_x3_variant < 0) {
    const char* e = getenv("NAWS_X3_VARIANT");
    g_x3_variant = e ? atoi(e) : 0;
  }
  if (N <= 64 || M <= 128) return launch_x3<128, 128, 2, 2, 2>(g, batch, s);
  // short K (the Winograd batch GEMMs): prologue and epilogue are a large share of a tile's life,
  // two 4-wave workgroups per CU overlap one's epilogue with the other's K loop
  if (K <= 1024 && g_x3_variant != 4) {
    if (g_x3_variant == 5) return launch_x3<128, 128, 2, 2, 2>(g, batch, s);
    return launch_x3<256, 128, 2, 2, 2>(g, batch, s);
  }
  switch (g_x3_variant) {
    case 1: return launch_x3<256, 128, 2, 2, 2>(g, batch, s);
    case 2: return launch_x3<256, 256, 2, 4, 2>(g, batch, s);
    case 3: return launch_x3<256, 128, 2, 2, 3>(g, batch, s);
    default: return launch_x3<256, 256, 2, 4, 3>(g, batch, s);
  }
}

// P[2][batch][kpad/16][outer][16] f16 + scales[2][batch][outer]: [0] = |x| maxima (bit patterns,
// scratch), [1] = 1/scale per outer index, handed to naws_gemm_f32_f16x2_nt.
// As naws_split_f16x2 on X' = diag(rowmul) X (rowmul: one factor per source row, shared by the
// batch items; powers of two keep it exact): the form dY takes in dW = dY^T X when X's planes
// carry per-row scales of their own (naws_roi_pool_f_f16x2_fwd + naws_f16_planes_transpose).
extern "C" int naws_split_f16x2_kscaled(const float* X, int batch, int rows, int cols, int ld,
                                        int64_t strideX, int transpose, int kpad, void* P,
                                        float* scales, const float* rowmul, void* stream) {
  if (batch <= 0 || rows <= 0 || cols <= 0 || ld < cols) return NAWS_ERR_SHAPE;
  NAWS_REQUIRE_PTR(X); NAWS_REQUIRE_PTR(P); NAWS_REQUIRE_PTR(scales);
  const int kdim = transpose ? rows : cols;
  if (kpad != (kdim + 31) / 32 * 32) return NAWS_ERR_ARG;
  if (((uintptr_t)P & 15) != 0) return NAWS_ERR_ARG;
  const int outer = transpose ? cols : rows;
  const long long sp = (long long)kpad * outer;
  const long long plane = (long long)batch * sp;
  const long long gy = naws_cdiv(transpose ? kpad : rows, 64);
  const long long gy_src = naws_cdiv(rows, 64);
  if (batch > 65535 || gy > 65535 || gy_src > 65535) return NAWS_ERR_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  unsigned* amax = (unsigned*)scales;
  float* inv = scales + (long long)batch * outer;
  if (hipMemsetAsync(amax, 0, sizeof(unsigned) * (size_t)batch * outer, s) != hipSuccess)
    return NAWS_ERR_LAUNCH;
  dim3 gsrc((unsigned)naws_cdiv(cols, 64), (unsigned)gy_src, batch);
  if (!transpose) {
    dim3 grow((unsigned)naws_cdiv(cols, 64 * AMAX_CT), (unsigned)gy_src, batch);
    hipLaunchKernelGGL((amax_kernel<false>), grow, dim3(256), 0, s, X, rows, cols, ld, outer,
                       (long long)strideX, amax, rowmul);
    dim3 grid((unsigned)naws_cdiv(kpad, 64), (unsigned)gy, batch);
    hipLaunchKernelGGL((split2h_kernel<false>), grid, dim3(256), 0, s, X, rows, cols, ld, outer,
                       kpad / 16, (long long)strideX, plane, sp, (const unsigned*)amax, inv,
                       (unsigned short*)P, rowmul);
  } else {
    hipLaunchKernelGGL((amax_kernel<true>), gsrc, dim3(256), 0, s, X, rows, cols, ld, outer,
                       (long long)strideX, amax, rowmul);
    dim3 grid((unsigned)naws_cdiv(cols, 64), (unsigned)gy, batch);
    hipLaunchKernelGGL((split2h_kernel<true>), grid, dim3(256), 0, s, X, rows, cols, ld, outer,
                       kpad / 16, (long long)strideX, plane, sp, (const unsigned*)amax, inv,
                       (unsigned short*)P, rowmul);
  }
  return naws_check_launch();
}

// naws_split_f16x2_dual: see split2h_dual_kernel.  rowmax [batch][rows] / colmax [batch][cols]: bit
// patterns of max|x| per row / per column of diag(rowmul) X (as reported by a *_amax GEMM).  Pn /
// Pt nullable (at least one); scales_n [2][batch][rows] / scales_t [2][batch][cols]: [1] receives
// 1/scale ([0] is free for the caller - typically it IS the rowmax / colmax vector).
extern "C" int naws_split_f16x2_dual(const float* X, int batch, int rows, int cols, int ld,
                                     int64_t strideX, const uint32_t* rowmax, const uint32_t* colmax,
                                     const float* rowmul, void* Pn, float* scales_n, int kpad_n,
                                     void* Pt, float* scales_t, int kpad_t, void* stream) {
  if (batch <= 0 || rows <= 0 || cols <= 0 || ld < cols) return NAWS_ERR_SHAPE;
  NAWS_REQUIRE_PTR(X);
  if (!Pn && !Pt) return NAWS_ERR_NULL;
  if (Pn) {
    NAWS_REQUIRE_PTR(rowmax); NAWS_REQUIRE_PTR(scales_n);
    if (kpad_n != (cols + 31) / 32 * 32 || ((uintptr_t)Pn & 15) != 0) return NAWS_ERR_ARG;
  }
  if (Pt) {
    NAWS_REQUIRE_PTR(colmax); NAWS_REQUIRE_PTR(scales_t);
    if (kpad_t != (rows + 31) / 32 * 32 || ((uintptr_t)Pt & 15) != 0) return NAWS_ERR_ARG;
  }
  const long long gy = naws_cdiv(rows, 64);
  if (batch > 65535 || gy > 65535) return NAWS_ERR_UNSUPPORTED;
  dim3 grid((unsigned)naws_cdiv(cols, 64), (unsigned)gy, batch);
  hipLaunchKernelGGL(split2h_dual_kernel, grid, dim3(256), 0, (hipStream_t)stream, X, rows, cols, ld,
                     (long long)strideX, (const unsigned*)rowmax, (const unsigned*)colmax, rowmul,
                     (unsigned short*)Pn, Pn ? scales_n + (long long)batch * rows : nullptr,
                     kpad_n / 16, (unsigned short*)Pt, Pt ? scales_t + (long long)batch * cols : nullptr,
                     kpad_t / 16, batch);
  return naws_check_launch();
}

extern "C" int naws_split_f16x2(const float* X, int batch, int rows, int cols, int ld,
                                int64_t strideX, int transpose, int kpad, void* P, float* scales,
                                void* stream) {
  return naws_split_f16x2_kscaled(X, batch, rows, cols, ld, strideX, transpose, kpad, P, scales,
                                  nullptr, stream);
}

static int g_h2_variant = -1;

extern "C" int naws_gemm_f32_f16x2_nt_amax(int M, int N, int K, const void* A2, int64_t slabA,
                                           int64_t planeA, const float* scaleA, const void* B2,
                                           int64_t slabB, int64_t planeB, const float* scaleB,
                                           float* C, int ldc, int batch, int64_t strideA,
                                           int64_t strideB, int64_t strideC, int64_t strideScaleA,
                                           int64_t strideScaleB, int epilogue, const float* bias,
                                           int64_t strideBias, const float* aux, int ldaux,
                                           float alpha, float drop_ratio, uint64_t seed,
                                           int accumulate, uint32_t* rowmax, int rowmax_seg_cols,
                                           uint32_t* colmax, const float* colmax_rowmul,
                                           void* stream) {
  if (M <= 0 || N <= 0 || K <= 0 || batch <= 0) return NAWS_ERR_SHAPE;
  NAWS_REQUIRE_PTR(A2); NAWS_REQUIRE_PTR(B2); NAWS_REQUIRE_PTR(C);
  NAWS_REQUIRE_PTR(scaleA); NAWS_REQUIRE_PTR(scaleB);
  if (epilogue < NAWS_EPI_NONE || epilogue > NAWS_EPI_GATE_POS) return NAWS_ERR_ARG;
  if (epilogue == NAWS_EPI_GATE_POS && aux == nullptr) return NAWS_ERR_NULL;
  if (epilogue == NAWS_EPI_BIAS_RELU_DROP && !(drop_ratio >= 0.f && drop_ratio < 1.f)) return NAWS_ERR_ARG;
  if (slabA < (int64_t)M * 16 || slabB < (int64_t)N * 16 || ldc < N) return NAWS_ERR_SHAPE;
  if (K % 32 != 0 || slabA % 8 != 0 || slabB % 8 != 0 || planeA % 8 != 0 || planeB % 8 != 0 ||
      strideA % 8 != 0 || strideB % 8 != 0)
    return NAWS_ERR_ARG;
  if ((((uintptr_t)A2 | (uintptr_t)B2) & 15) != 0) return NAWS_ERR_ARG;
  if (batch > 65535) return NAWS_ERR_UNSUPPORTED;
  if ((long long)(M - 1) * ldc + N > 0x7fffffffLL ||
      (aux && (long long)(M - 1) * ldaux + N > 0x7fffffffLL))
    return NAWS_ERR_UNSUPPORTED;
  XArgs g{};
  g.A = (const unsigned short*)A2; g.B = (const unsigned short*)B2; g.C = C;
  g.M = M; g.N = N; g.K = K; g.ldc = ldc;
  g.planeA = planeA; g.planeB = planeB; g.slabA = slabA; g.slabB = slabB;
  g.sA = strideA; g.sB = strideB; g.sC = strideC; g.sBias = strideBias;
  g.bias = bias; g.aux = aux; g.ldaux = ldaux; g.alpha = alpha;
  g.drop_thr = naws_drop_threshold(drop_ratio);
  g.drop_scale = (float)(1.0 / (1.0 - (double)drop_ratio));
  g.seed = seed; g.epilogue = epilogue; g.accumulate = accumulate;
  g.rs = scaleA; g.cs = scaleB; g.sRs = strideScaleA; g.sCs = strideScaleB;
  if (rowmax || colmax) {
    // a wave's columns must lie in one rowmax segment: segments are multiples of the widest tile
    const int seg = rowmax_seg_cols > 0 ? rowmax_seg_cols : N;
    if (rowmax && seg < N && seg % 256 != 0) return NAWS_ERR_ARG;
    const int nseg = (int)naws_cdiv(N, seg);
    g.am.rowmax = rowmax; g.am.colmax = colmax; g.am.colmul = colmax_rowmul;
    g.am.seg_cols = seg >= N ? 0x40000000 : seg;
    g.am.sRow = (long long)nseg * M; g.am.sCol = N;
  }
  hipStream_t s = (hipStream_t)stream;
  if (g_h2_variant < 0) {
    const char* e = getenv("NAWS_H2_VARIANT");
    g_h2_variant = e ? atoi(e) : 0;
  }
  if (N <= 64 || M <= 128) return launch_x3<128, 128, 2, 2, 2, 2, 2, true>(g, batch, s);
  // fewer 256x256 tiles than CUs (the column remainder of a wgrad split into whole waves)
  if (naws_cdiv(M, 256) * naws_cdiv(N, 256) * batch < 256 && g_h2_variant != 5)
    return launch_x3<128, 128, 2, 2, 2, 2, 2, true>(g, batch, s);
  // short K (the Winograd batch GEMMs): two 4-wave workgroups per CU overlap one's prologue /
  // epilogue with the other's K loop
  if (K <= 1024 && g_h2_variant != 5) {
    if (g_h2_variant == 6) return launch_x3<256, 128, 2, 2, 2, 2, 1, true>(g, batch, s);
    if (g_h2_variant == 12) return launch_x3<128, 128, 2, 2, 4, 2, 1, true>(g, batch, s);
    if (g_h2_variant == 14) return launch_x3<128, 128, 2, 2, 3, 2, 2, true>(g, batch, s);
    if (g_h2_variant == 15) return launch_x3_m16<128, 128, 2, 2, 2, 2, 2, true>(g, batch, s);
    if (g_h2_variant == 17) return launch_x3<128, 128, 2, 2, 2, 2, 2, true>(g, batch, s);
    // 16-deep K-steps on a ring of 3 stages (48 KB: still two workgroups per CU): the DMA runs two
    // steps ahead instead of one, which is what a 16..32-step tile needs (tools/ab_h2.py, the
    // Winograd batch GEMM of conv4_2: 0.101 vs 0.114 ms with 32-deep steps and 2 stages)
    return launch_x3<128, 128, 2, 2, 3, 2, 1, true>(g, batch, s);
  }
  switch (g_h2_variant) {
    case 1: return launch_x3<256, 256, 2, 4, 3, 2, 1, true>(g, batch, s);
    case 2: return launch_x3<256, 256, 2, 4, 4, 2, 1, true>(g, batch, s);
    case 3: return launch_x3<256, 128, 2, 2, 2, 2, 2, true>(g, batch, s);
    case 4: return launch_x3<256, 128, 2, 2, 3, 2, 2, true>(g, batch, s);
    case 7: return launch_x3_m16<256, 256, 2, 4, 2, 2, 2, true>(g, batch, s);
    case 10: return launch_x3_m16<256, 256, 4, 2, 2, 2, 2, true>(g, batch, s);
    case 11: return launch_x3_m16<256, 128, 2, 2, 3, 2, 2, true>(g, batch, s);
    case 9: return launch_x3<256, 256, 2, 4, 2, 2, 2, true>(g, batch, s);
    // 16x16x32 MFMAs, 4 x 2 waves (64 x 128 per wave): the chip holds a higher clock on this shape
    // in the power-limited long-K GEMMs (tools/ab_h2.py, interleaved: fc6 fwd 3.62 vs 3.83 ms)
    default: return launch_x3_m16<256, 256, 4, 2, 2, 2, 2, true>(g, batch, s);
  }
}

extern "C" int naws_gemm_f32_f16x2_nt(int M, int N, int K, const void* A2, int64_t slabA,
                                      int64_t planeA, const float* scaleA, const void* B2,
                                      int64_t slabB, int64_t planeB, const float* scaleB, float* C,
                                      int ldc, int batch, int64_t strideA, int64_t strideB,
                                      int64_t strideC, int64_t strideScaleA, int64_t strideScaleB,
                                      int epilogue, const float* bias, int64_t strideBias,
                                      const float* aux, int ldaux, float alpha, float drop_ratio,
                                      uint64_t seed, int accumulate, void* stream) {
  return naws_gemm_f32_f16x2_nt_amax(M, N, K, A2, slabA, planeA, scaleA, B2, slabB, planeB, scaleB,
                                     C, ldc, batch, strideA, strideB, strideC, strideScaleA,
                                     strideScaleB, epilogue, bias, strideBias, aux, ldaux, alpha,
                                     drop_ratio, seed, accumulate, nullptr, 0, nullptr, nullptr,
                                     stream);
}

extern "C" int naws_conv3x3_nhwc_f32x3_fwd(const float* X, const void* W3, const float* bias,
                                           int N, int H, int W, int Cin, int Cout, int dilation,
                                           int relu, float* Y, void* stream) {
  if (N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return NAWS_ERR_SHAPE;
  if (dilation < 1) return NAWS_ERR_ARG;
  if (Cin % 16 != 0 || Cout % 4 != 0) return NAWS_ERR_UNSUPPORTED;
  NAWS_REQUIRE_PTR(X); NAWS_REQUIRE_PTR(W3); NAWS_REQUIRE_PTR(Y);
  if (!bias && relu) return NAWS_ERR_ARG;
  if ((((uintptr_t)X | (uintptr_t)W3) & 15) != 0) return NAWS_ERR_ARG;
  const long long pix = (long long)N * H * W;
  if (pix > 0x7fffffffLL || pix * Cin * 4 > 0xFFFFFF00LL) return NAWS_ERR_UNSUPPORTED;
  CArgs g{};
  g.X = X; g.B = (const unsigned short*)W3; g.bias = bias; g.Y = Y;
  g.M = (int)pix; g.Cout = Cout; g.Cin = Cin; g.H = H; g.W = W; g.dil = dilation; g.relu = relu;
  g.slabB = (long long)Cout * 16;
  g.planeB = (long long)9 * Cin * Cout;
  g.bytesX = (unsigned)(pix * Cin * 4);
  hipStream_t s = (hipStream_t)stream;
  if (g_x3_variant < 0) {
    const char* e = getenv("NAWS_X3_VARIANT");
    g_x3_variant = e ? atoi(e) : 0;
  }
  // wide shallow layers: the halo-tile kernel (input gathered once per channel slab, not per tap)
  if (dilation == 1 && Cout <= 128 && Cout % 32 == 0 && g_x3_variant != 6) {
    if (Cout <= 64) return launch_conv_x3_halo<64>(g, N, s);
    return launch_conv_x3_halo<128>(g, N, s);
  }
  if (Cout <= 64) return launch_conv_x3<256, 64, 4, 1>(g, s);
  if (Cout <= 128 || naws_cdiv(pix, 256) * naws_cdiv(Cout, 256) < 256) {
    if (naws_cdiv(pix, 256) * naws_cdiv(Cout, 128) < 512) return launch_conv_x3<128, 128, 2, 2>(g, s);
    return launch_conv_x3<256, 128, 2, 2>(g, s);
  }
  return launch_conv_x3<256, 256, 2, 4>(g, s);
}

// |x| maximum of a tensor, as a bit pattern (non-negative floats order like unsigned words).
namespace {
__global__ __launch_bounds__(256) void amax_word_kernel(const float* __restrict__ X, long long n,
                                                        unsigned* __restrict__ out) {
  float m = 0.f;
  // scalar head up to the first 16-byte boundary, float4 body, scalar tail
  const long long head = min((long long)(((16 - ((uintptr_t)X & 15)) & 15) / 4), n);
  const long long n4 = (n - head) / 4;
  const float* Xb = X + head;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    const float4 v = reinterpret_cast<const float4*>(Xb)[i];
    m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
  }
  if (blockIdx.x == 0) {
    if (threadIdx.x < head) m = fmaxf(m, fabsf(X[threadIdx.x]));
    const long long t0 = head + n4 * 4;
    if (t0 + threadIdx.x < n && threadIdx.x < 4) m = fmaxf(m, fabsf(X[t0 + threadIdx.x]));
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
}  // namespace

extern "C" int naws_amax_f32(const float* X, int64_t n, uint32_t* out, void* stream) {
  if (n <= 0) return NAWS_ERR_SHAPE;
  NAWS_REQUIRE_PTR(X); NAWS_REQUIRE_PTR(out);
  if (((uintptr_t)X & 3) != 0) return NAWS_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  if (hipMemsetAsync(out, 0, sizeof(uint32_t), s) != hipSuccess) return NAWS_ERR_LAUNCH;
  hipLaunchKernelGGL(amax_word_kernel, dim3((unsigned)std::min<long long>(naws_cdiv(n / 4 + 1, 256 * 8), 2048)),
                     dim3(256), 0, s, X, (long long)n, (unsigned*)out);
  return naws_check_launch();
}

// fp16x2 form of the shallow-layer convolution (the halo-tile kernel): W2 / scaleW =
// naws_split_f16x2 of the packed weight viewed [Cout][9*Cin]; the activation scale comes from
// *amax_in * in_mul + in_add (an upper bound of max|X|); dilation 1, Cout <= 128, Cout % 32 == 0.
extern "C" int naws_conv3x3_nhwc_f16x2_fwd(const float* X, const void* W2, const float* scaleW,
                                           const float* bias, int N, int H, int W, int Cin,
                                           int Cout, int dilation, int relu, float* Y,
                                           const uint32_t* amax_in, float in_mul, float in_add,
                                           uint32_t* amax_out, int amax_out_zeroed, int pool2,
                                           void* stream) {
  if (N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return NAWS_ERR_SHAPE;
  if (dilation != 1 && dilation != 2) return NAWS_ERR_UNSUPPORTED;
  if (dilation == 2 && pool2) return NAWS_ERR_ARG;
  if (Cin % 16 != 0 || (9 * Cin) % 32 != 0 || Cout % 32 != 0 || (Cout > 128 && Cout % 128 != 0))
    return NAWS_ERR_UNSUPPORTED;
  NAWS_REQUIRE_PTR(X); NAWS_REQUIRE_PTR(W2); NAWS_REQUIRE_PTR(scaleW); NAWS_REQUIRE_PTR(Y);
  NAWS_REQUIRE_PTR(amax_in);
  if (!bias && relu) return NAWS_ERR_ARG;
  if (!(in_mul > 0.f) || !(in_add >= 0.f) || amax_in == amax_out) return NAWS_ERR_ARG;
  if ((((uintptr_t)X | (uintptr_t)W2) & 15) != 0) return NAWS_ERR_ARG;
  const long long pix = (long long)N * H * W;
  if (pix > 0x7fffffffLL || pix * Cin * 4 > 0xFFFFFF00LL) return NAWS_ERR_UNSUPPORTED;
  CArgs g{};
  g.X = X; g.B = (const unsigned short*)W2; g.bias = bias; g.Y = Y;
  g.M = (int)pix; g.Cout = Cout; g.Cin = Cin; g.H = H; g.W = W; g.dil = dilation; g.relu = relu;
  g.slabB = (long long)Cout * 16;
  g.planeB = (long long)9 * Cin * Cout;
  g.bytesX = (unsigned)(pix * Cin * 4);
  g.scaleB = scaleW; g.amax_in = (const unsigned*)amax_in; g.in_mul = in_mul; g.in_add = in_add;
  g.amax_out = (unsigned*)amax_out;
  g.pool = pool2 ? 1 : 0;
  if (pool2 && (H < 2 || W < 2)) return NAWS_ERR_SHAPE;
  hipStream_t s = (hipStream_t)stream;
  // (a chain of layers zeroes all its words with one fill and passes amax_out_zeroed = 1: a
  // memset per layer is a 6 us kernel plus a launch gap in a dependent chain)
  if (amax_out && !amax_out_zeroed && hipMemsetAsync(amax_out, 0, sizeof(uint32_t), s) != hipSuccess)
    return NAWS_ERR_LAUNCH;
  // 64-wide channel tiles (3 workgroups per CU) where 128-wide ones leave CUs idle or the layer is
  // deep: measured per layer at 2 images (tools/kernel_bench.py --what x3): conv3_x 0.137 / 0.259
  // vs 0.156 / 0.289 ms, conv4_2 0.282 vs 0.297; conv2_2 (608 tiles of 128) keeps 128
  bool bn64 = Cout <= 64;
  if (!bn64 && Cout % 64 == 0) {
    const long long t128 = (long long)N * naws_cdiv(H, 8) * naws_cdiv(W, 32) * naws_cdiv(Cout, 128);
    bn64 = Cin >= 128 && Cout >= 256 && t128 < 4 * 512;
    const char* e = getenv("NAWS_CONV_BN");            // A/B knob (tools/kernel_bench.py)
    if (e) bn64 = atoi(e) == 64;
  }
  if (dilation == 2)
    return bn64 ? launch_conv_x3_halo<64, true, 2>(g, N, s) : launch_conv_x3_halo<128, true, 2>(g, N, s);
  return bn64 ? launch_conv_x3_halo<64, true>(g, N, s) : launch_conv_x3_halo<128, true>(g, N, s);
}

extern "C" int naws_gemm_bf16_slab_nt(int M, int N, int K, const void* A, int64_t slabA,
                                      const void* B, int64_t slabB, float* C, int ldc, int batch,
                                      int64_t strideA, int64_t strideB, int64_t strideC,
                                      int epilogue, const float* bias, int64_t strideBias,
                                      const float* aux, int ldaux, float alpha, float drop_ratio,
                                      uint64_t seed, int accumulate, void* stream) {
  if (M <= 0 || N <= 0 || K <= 0 || batch <= 0) return NAWS_ERR_SHAPE;
  NAWS_REQUIRE_PTR(A); NAWS_REQUIRE_PTR(B); NAWS_REQUIRE_PTR(C);
  if (epilogue < NAWS_EPI_NONE || epilogue > NAWS_EPI_GATE_POS) return NAWS_ERR_ARG;
  if (epilogue == NAWS_EPI_GATE_POS && aux == nullptr) return NAWS_ERR_NULL;
  if (epilogue == NAWS_EPI_BIAS_RELU_DROP && !(drop_ratio >= 0.f && drop_ratio < 1.f)) return NAWS_ERR_ARG;
  if (slabA < (int64_t)M * 16 || slabB < (int64_t)N * 16 || ldc < N) return NAWS_ERR_SHAPE;
  if (K % 64 != 0 || slabA % 8 != 0 || slabB % 8 != 0 || strideA % 8 != 0 || strideB % 8 != 0)
    return NAWS_ERR_ARG;
  if ((((uintptr_t)A | (uintptr_t)B) & 15) != 0) return NAWS_ERR_ARG;
  if (batch > 65535) return NAWS_ERR_UNSUPPORTED;
  if ((long long)(M - 1) * ldc + N > 0x7fffffffLL ||
      (aux && (long long)(M - 1) * ldaux + N > 0x7fffffffLL))
    return NAWS_ERR_UNSUPPORTED;
  XArgs g{};
  g.A = (const unsigned short*)A; g.B = (const unsigned short*)B; g.C = C;
  g.M = M; g.N = N; g.K = K; g.ldc = ldc;
  g.slabA = slabA; g.slabB = slabB;
  g.sA = strideA; g.sB = strideB; g.sC = strideC; g.sBias = strideBias;
  g.bias = bias; g.aux = aux; g.ldaux = ldaux; g.alpha = alpha;
  g.drop_thr = naws_drop_threshold(drop_ratio);
  g.drop_scale = (float)(1.0 / (1.0 - (double)drop_ratio));
  g.seed = seed; g.epilogue = epilogue; g.accumulate = accumulate;
  hipStream_t s = (hipStream_t)stream;
  if (N <= 64 || M <= 128) return launch_x3<128, 128, 2, 2, 2, 1, 4>(g, batch, s);
  if (g_h2_variant < 0) {
    const char* e = getenv("NAWS_H2_VARIANT");
    g_h2_variant = e ? atoi(e) : 0;
  }
  if (g_h2_variant == 9) return launch_x3<256, 256, 2, 4, 2, 1, 4>(g, batch, s);
  return launch_x3_m16<256, 256, 4, 2, 2, 1, 4, false>(g, batch, s);
}

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
#include "x3_common.h"

#define g_x3_variant naws_knob(NAWS_KNOB_X3)

namespace {

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
  int tiles_m, tiles_n, batch;
  int vec4;                  // every row of C / aux and bias / cs start 16-byte aligned: float4 epilogue
  const float* rs;           // fp16x2: per-row / per-column power-of-two factors undoing the
  const float* cs;           //         operand scaling (null otherwise)
  long long sRs, sCs;
  NawsAmax am;               // |C| maxima for the consumer's operand split (naws_common.h)
  // COL2 (the Winograd frequency-column form, winograd.hip): K = 4 blocks of col_steps K-steps;
  // C receives blocks 0 + 1 + 2, C + sC2 receives blocks 1 - 2 - 3
  long long sC2;
  int col_steps;
  // SGD form (naws_gemm_bf16_slab_nt_sgd): C is never written; the tile's products are the
  // gradient of param[M][ldp] and go straight into the update
  float* mom;
  float* param;
  int ldp;
  const float* lr;           // device scalar: the base learning rate
  float lr_mult, wd, momentum, gscale;
  int nesterov, first;
  unsigned short* P;         // param's bf16 operand plane [N/16][prows][16]
  int prows;
  // Dropout counter of element (row, col) = row * drop_ld + drop_c0 + col (0 / 0: the launch's own
  // N and column 0).  A launch that produces a COLUMN RANGE of a wider activation (fc6 forward
  // cut along the weight rows, naws_gemm_f32_f16x2_nt_cols) draws the masks of the full-width launch.
  int drop_ld, drop_c0;
};

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
template <int BM, int BN, int WM, int WN, int STAGES, int NPL = 3, int KS = 1, bool F16 = false,
          bool COL2 = false>
__global__ __launch_bounds__(64 * WM * WN, (WM * WN >= 8 ? 1 : 2)) void gemm_x3_kernel(XArgs g) {
  static_assert(!F16 || NPL == 2, "fp16x2 uses two planes");
  static_assert(!COL2 || (F16 && KS == 1), "the column form is an fp16x2 batch GEMM");
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

  // Workgroups go to the 8 XCDs round-robin by launch index.  The (batch item, tile) space is cut
  // into 8 contiguous ranges, one per XCD: within a batch item an XCD's L2 sees neighbouring
  // tiles, and a batched launch (the 16 Winograd frequencies) keeps whole batch items on one XCD,
  // so every operand is fetched into ONE L2 instead of all eight (PMC: 301 MB fetched per launch
  // for 95 MB of operands when each frequency's tiles were spread over the XCDs)
  const int ntiles = g.tiles_m * g.tiles_n;
  int lid = blockIdx.x;
  {
    const int total = ntiles * g.batch;
    const int q = total >> 3, rem = total & 7, xcd = lid & 7, within = lid >> 3;
    lid = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + within;
  }
  const long long bz = lid / ntiles;
  lid -= (int)bz * ntiles;
  constexpr int GM = 8;
  const int per_group = GM * g.tiles_n;
  const int grp = lid / per_group;
  const int first_m = grp * GM;
  const int gsz = min(g.tiles_m - first_m, GM);
  const int tm = first_m + (lid % per_group) % gsz;
  const int tn = (lid % per_group) / gsz;
  const int m0 = tm * BM, n0 = tn * BN;

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
  f32x16 acc1[COL2 ? TI : 1][COL2 ? TJ : 1];     // COL2: the second Winograd row (blocks 1 - 2 - 3)
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int j = 0; j < TJ; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        acc[i][j][e] = 0.f;
        if constexpr (COL2) acc1[i][j][e] = 0.f;
      }

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
#define NAWS_X3_TERM_TO(ACC, AF, P, Q)                                                          \
  _Pragma("unroll") for (int i = 0; i < TI; ++i) _Pragma("unroll") for (int j = 0; j < TJ; ++j) \
      ACC[i][j] = mfma16(AF[P][i], b[Q][j], ACC[i][j]);
#define NAWS_X3_TERM(P, Q) NAWS_X3_TERM_TO(acc, a, P, Q)
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
      if constexpr (COL2) {
        // frequency block of this K-step (wave-uniform): row 0 of A^T takes blocks 0, 1, 2, row 1
        // takes 1, -2, -3 - the same fragments feed both accumulator sets, the sign is a flip of
        // the A fragments' sign bits (hi and lo planes alike: exact)
        const int blk = t / g.col_steps;
        if (blk <= 2) {
          NAWS_X3_TERM(0, 0)
          NAWS_X3_TERM(0, 1)
          NAWS_X3_TERM(1, 0)
        }
        if (blk == 1) {
          NAWS_X3_TERM_TO(acc1, a, 0, 0)
          NAWS_X3_TERM_TO(acc1, a, 0, 1)
          NAWS_X3_TERM_TO(acc1, a, 1, 0)
        } else if (blk >= 2) {
          vec_t an[NPL][TI];
#pragma unroll
          for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
            for (int i = 0; i < TI; ++i) {
              u32x4 w = *reinterpret_cast<const u32x4*>(&a[pl][i]);
              w.x ^= 0x80008000u; w.y ^= 0x80008000u; w.z ^= 0x80008000u; w.w ^= 0x80008000u;
              an[pl][i] = *reinterpret_cast<const vec_t*>(&w);
            }
          NAWS_X3_TERM_TO(acc1, an, 0, 0)
          NAWS_X3_TERM_TO(acc1, an, 0, 1)
          NAWS_X3_TERM_TO(acc1, an, 1, 0)
        }
      } else {
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
    }
#undef NAWS_X3_TERM
#undef NAWS_X3_TERM_TO
    st_cur = (st_cur + 1 == STAGES) ? 0 : st_cur + 1;
    st_fill = (st_fill + 1 == STAGES) ? 0 : st_fill + 1;
  }

  const float* bias = g.bias ? g.bias + bz * g.sBias : nullptr;
  const float* aux = g.aux ? g.aux + bz * g.sC : nullptr;
  const int epi = g.epilogue;
#define NAWS_EPI_ACC acc
#define NAWS_EPI_C C
#include "gemm_x3_epilogue.inc"
#undef NAWS_EPI_ACC
#undef NAWS_EPI_C
  if constexpr (COL2) {            // (A/B build only) the second accumulator set
    float* C2 = C + g.sC2;
#define NAWS_EPI_ACC acc1
#define NAWS_EPI_C C2
#include "gemm_x3_epilogue.inc"
#undef NAWS_EPI_ACC
#undef NAWS_EPI_C
  }
  if (g.am.rowmax || g.am.colmax)
    naws_tile_amax_32<TI, TJ>(acc, m0 + wm * WTM, n0 + wn * WTN, g.M, g.N, lane, g.am, bz);
}

template <int BM, int BN, int WM, int WN, int STAGES, int NPL = 3, int KS = 1, bool F16 = false,
          bool COL2 = false>
int launch_x3(XArgs& g, int batch, hipStream_t s) {
  g.tiles_m = (int)naws_cdiv(g.M, BM);
  g.tiles_n = (int)naws_cdiv(g.N, BN);
  const size_t lds = (size_t)STAGES * NPL * KS * (BM + BN) * 32;
  auto kern = gemm_x3_kernel<BM, BN, WM, WN, STAGES, NPL, KS, F16, COL2>;
  if (naws_allow_lds(kern) != NAWS_OK) return NAWS_ERR_LAUNCH;
  g.batch = batch;
  if ((long long)g.tiles_m * g.tiles_n * batch > 0x7fffffffLL) return NAWS_ERR_UNSUPPORTED;
  dim3 grid((unsigned)(g.tiles_m * g.tiles_n * batch), 1, 1);
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

template <int BM, int BN, int WM, int WN, int STAGES, int NPL, int KS, bool F16, bool SGD = false>
__global__ __launch_bounds__(64 * WM * WN, (WM * WN >= 8 ? 1 : 2)) void gemm_x3_m16_kernel(XArgs g) {
  static_assert(KS % 2 == 0, "a 16x16x32 MFMA spans two 16-deep slabs");
  static_assert(!SGD || (NPL == 1 && !F16), "the update epilogue: the bf16 plan's one-plane form");
  static_assert(!F16 || NPL <= 2, "f16 operands have one or two planes");
  typedef typename OperandVec<F16>::type vec_t;
  constexpr int NT = 64 * WM * WN;
  constexpr int WTM = BM / WM, WTN = BN / WN;
  constexpr int TI = WTM / 16, TJ = WTN / 16;
  constexpr int IH = TI >= 8 ? 2 : 1, TIH = TI / IH;        // A fragments are read in row halves
  constexpr int A_PLANE = BM * 32, B_PLANE = BN * 32;
  constexpr int NQ = NPL * KS;
  constexpr int STAGE = NQ * (A_PLANE + B_PLANE);
  constexpr int PIECE_ROWS = NT / 2;
  // DMA pieces (1 KB = 32 rows of one plane-slab).  Tiles whose sides are multiples of the
  // workgroup's PIECE_ROWS use the round form (wave w takes rows p * PIECE_ROWS + 32 w of every
  // plane-slab); other tiles (256 x 128 on 8 waves) deal the flat piece list round-robin:
  // piece q = wid + NW k, the first NQ BM / 32 of them A pieces.
  constexpr bool ROUND = BM % PIECE_ROWS == 0 && BN % PIECE_ROWS == 0;
  constexpr int NW = WM * WN;
  constexpr int PA = ROUND ? BM / PIECE_ROWS : 1, PB = ROUND ? BN / PIECE_ROWS : 1;
  constexpr int NPA = NQ * BM / 32, NPB = NQ * BN / 32;      // pieces per step
  constexpr int G = ROUND ? NQ * (PA + PB) : (NPA + NPB) / NW;
  static_assert(ROUND || (NPA % NW == 0 && NPB % NW == 0), "tile vs workgroup");
  static_assert((STAGES - 2) * G <= 63, "vmcnt range");
  extern __shared__ __attribute__((aligned(16))) unsigned char smx[];

  // Workgroups go to the 8 XCDs round-robin by launch index.  The (batch item, tile) space is cut
  // into 8 contiguous ranges, one per XCD: within a batch item an XCD's L2 sees neighbouring
  // tiles, and a batched launch (the 16 Winograd frequencies) keeps whole batch items on one XCD,
  // so every operand is fetched into ONE L2 instead of all eight (PMC: 301 MB fetched per launch
  // for 95 MB of operands when each frequency's tiles were spread over the XCDs)
  const int ntiles = g.tiles_m * g.tiles_n;
  int lid = blockIdx.x;
  {
    const int total = ntiles * g.batch;
    const int q = total >> 3, rem = total & 7, xcd = lid & 7, within = lid >> 3;
    lid = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + within;
  }
  const long long bz = lid / ntiles;
  lid -= (int)bz * ntiles;
  constexpr int GM = 8;
  const int per_group = GM * g.tiles_n;
  const int grp = lid / per_group;
  const int first_m = grp * GM;
  const int gsz = min(g.tiles_m - first_m, GM);
  const int tm = first_m + (lid % per_group) % gsz;
  const int tn = (lid % per_group) / gsz;
  const int m0 = tm * BM, n0 = tn * BN;

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

  // flat form: this wave's pieces
  constexpr int GF = ROUND ? 1 : G;
  const unsigned short* fsrc[GF];
  int fdst[GF];
  if constexpr (!ROUND) {
    const int r32 = lane >> 1;
#pragma unroll
    for (int kq = 0; kq < GF; ++kq) {
      const int q = wid + NW * kq;
      if (kq < NPA / NW) {
        const int pq = q / (BM / 32), rg = q % (BM / 32);
        fsrc[kq] = A + (long long)(pq / KS) * g.planeA + (long long)(pq % KS) * g.slabA +
                   (long long)min(m0 + rg * 32 + r32, g.M - 1) * 16 + (lane & 1) * 8;
        fdst[kq] = pq * A_PLANE + rg * 1024;
      } else {
        const int q2 = q - NPA;
        const int pq = q2 / (BN / 32), rg = q2 % (BN / 32);
        fsrc[kq] = B + (long long)(pq / KS) * g.planeB + (long long)(pq % KS) * g.slabB +
                   (long long)min(n0 + rg * 32 + r32, g.N - 1) * 16 + (lane & 1) * 8;
        fdst[kq] = NQ * A_PLANE + pq * B_PLANE + rg * 1024;
      }
    }
  }
  auto issue = [&](int t, int st) {
    if constexpr (!ROUND) {
      unsigned char* base = smx + st * STAGE;
#pragma unroll
      for (int kq = 0; kq < GF; ++kq) {
        const long long adv = (long long)(t * KS) * (kq < NPA / NW ? g.slabA : g.slabB);
        __builtin_amdgcn_global_load_lds(NAWS_GLB_PTR(fsrc[kq] + adv), NAWS_LDS_PTR(base + fdst[kq]),
                                         16, 0, 0);
      }
      return;
    }
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
      acc[ih * TIH + i][j] = mfma32(b[Q][j], a[P][i], acc[ih * TIH + i][j]);
        NAWS_M16_TERM(0, 0)
        if constexpr (NPL >= 2) {
          NAWS_M16_TERM(0, 1)
          NAWS_M16_TERM(1, 0)
        }
        if constexpr (NPL == 3) {                     // the exact 3 x bf16 split: six terms
          NAWS_M16_TERM(1, 1)
          NAWS_M16_TERM(0, 2)
          NAWS_M16_TERM(2, 0)
        }
#undef NAWS_M16_TERM
      }
    }
    st_cur = (st_cur + 1 == STAGES) ? 0 : st_cur + 1;
    st_fill = (st_fill + 1 == STAGES) ? 0 : st_fill + 1;
  }

  // The MFMAs above ran with the operands swapped (B fragment first): an accumulator block is the
  // TRANSPOSED 16x16 block of C, i.e. lane (l15, kg) holds row l15, columns kg * 4 + e - four
  // consecutive columns of one row (same products, same k order: bit-identical to the un-swapped
  // form, which holds four rows of one column).  The epilogue therefore moves 16 bytes per lane
  // (C, aux, bias, column factors) wherever the layout allows (g.vec4), 4x fewer memory
  // instructions than the 4-byte form: the aux-reading fc7 dgrad 0.72 -> see DESIGN 0a.
  if constexpr (SGD) {
    // ---- the update in place of the store (one process, no gradient exchange in between;
    // gemm_btr.hip's SGD form for the bf16 plan): the product is the gradient element, then
    // acm_sgd_planes_kernel<2>'s element work - sgd_elem, momentum and parameter written back,
    // the updated weight rounded to the bf16 operand plane (a lane's four columns = 8 bytes; a
    // fragment's 16 rows x 32 bytes = one contiguous 512-byte run of the K-slab).  All of a row
    // group's loads are issued before its arithmetic and stores.
    const float LR = g.lr[0] * g.lr_mult;
#pragma unroll
    for (int i = 0; i < TI; ++i) {
      const int row = m0 + wm * WTM + i * 16 + l15;
      const bool row_on = row < g.M;
      const int rr = row_on ? row : g.M - 1;
      f32x4 pw[TJ], pm[TJ];
#pragma unroll
      for (int j = 0; j < TJ; ++j) {
        const int col = n0 + wn * WTN + j * 16 + kg * 4;
        const bool on = row_on && col < g.N;
        const long long o = (long long)rr * g.ldp + (on ? col : 0);
        pw[j] = *reinterpret_cast<const f32x4*>(g.param + o);
        if (!g.first) pm[j] = *reinterpret_cast<const f32x4*>(g.mom + o);
        else pm[j] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int j = 0; j < TJ; ++j) {
        const int col = n0 + wn * WTN + j * 16 + kg * 4;
        if (!row_on || col >= g.N) continue;          // N % 16 == 0: four columns in or out together
        const f32x4 v = acc[i][j];
        const long long o = (long long)row * g.ldp + col;
        f32x4 p = pw[j], m = pm[j];
        unsigned short q[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float me = m[e], pe = p[e];
          sgd_elem(v[e], me, pe, g.gscale, g.wd, LR, g.momentum, g.nesterov);
          m[e] = me; p[e] = pe;
          const __bf16 h = (__bf16)pe;
          q[e] = *reinterpret_cast<const unsigned short*>(&h);
        }
        *reinterpret_cast<f32x4*>(g.mom + o) = m;
        *reinterpret_cast<f32x4*>(g.param + o) = p;
        const long long po = ((long long)(col >> 4) * g.prows + row) * 16 + (col & 15);
        *reinterpret_cast<uint2*>(g.P + po) =
            make_uint2(q[0] | ((unsigned)q[1] << 16), q[2] | ((unsigned)q[3] << 16));
      }
    }
    return;
  }
  const float* bias = g.bias ? g.bias + bz * g.sBias : nullptr;
  const float* aux = g.aux ? g.aux + bz * g.sC : nullptr;
  const int epi = g.epilogue;
  const bool has_bias = bias && epi >= NAWS_EPI_BIAS && epi <= NAWS_EPI_BIAS_RELU_DROP;
#pragma unroll
  for (int i = 0; i < TI; ++i) {
    const int row = m0 + wm * WTM + i * 16 + l15;
    float rsv = 1.f;
    if constexpr (F16) rsv = (g.rs + bz * g.sRs)[min(row, g.M - 1)];
#pragma unroll
    for (int j = 0; j < TJ; ++j) {
      const int col = n0 + wn * WTN + j * 16 + kg * 4;
      f32x4 v = acc[i][j];
      if (row < g.M && col < g.N) {
        const bool full = g.vec4 && col + 3 < g.N;
        f32x4 bv = {0.f, 0.f, 0.f, 0.f}, cs = {1.f, 1.f, 1.f, 1.f}, ax = {0.f, 0.f, 0.f, 0.f},
              old = {0.f, 0.f, 0.f, 0.f};
        float* cp = C + (long long)row * g.ldc + col;
        const float* ap = aux ? aux + (long long)row * g.ldaux + col : nullptr;
        if (full) {
          if (has_bias) bv = *reinterpret_cast<const f32x4*>(bias + col);
          if constexpr (F16) cs = *reinterpret_cast<const f32x4*>(g.cs + bz * g.sCs + col);
          if (epi == NAWS_EPI_GATE_POS) ax = *reinterpret_cast<const f32x4*>(ap);
          if (g.accumulate) old = *reinterpret_cast<const f32x4*>(cp);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            if (col + e >= g.N) continue;
            if (has_bias) bv[e] = bias[col + e];
            if constexpr (F16) cs[e] = g.cs[bz * g.sCs + col + e];
            if (epi == NAWS_EPI_GATE_POS) ax[e] = ap[e];
            if (g.accumulate) old[e] = cp[e];
          }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float t = v[e];
          if constexpr (F16) t = t * rsv * cs[e];          // powers of two: exact, in this order
          t += bv[e];
          if (epi == NAWS_EPI_BIAS_RELU || epi == NAWS_EPI_BIAS_RELU_DROP) t = fmaxf(t, 0.f);
          if (epi == NAWS_EPI_BIAS_RELU_DROP) {
            const unsigned long long dld = g.drop_ld ? g.drop_ld : g.N;
            const unsigned long long idx =
                (unsigned long long)bz * g.M * dld + (unsigned long long)row * dld + (g.drop_c0 + col + e);
            t = naws_keep(g.seed, idx, g.drop_thr) ? t * g.drop_scale : 0.f;
          } else if (epi == NAWS_EPI_GATE_POS) {
            t = (ax[e] > 0.f) ? t * g.alpha : 0.f;
          }
          if (g.accumulate) t += old[e];
          v[e] = t;
        }
        if (full) {
          *reinterpret_cast<f32x4*>(cp) = v;
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (col + e < g.N) cp[e] = v[e];
        }
      }
      acc[i][j] = v;
    }
  }
  if (g.am.rowmax || g.am.colmax)
    naws_tile_amax_16t<TI, TJ>(acc, m0 + wm * WTM, n0 + wn * WTN, g.M, g.N, lane, g.am, bz);
}

template <int BM, int BN, int WM, int WN, int STAGES, int NPL, int KS, bool F16, bool SGD = false>
int launch_x3_m16(XArgs& g, int batch, hipStream_t s) {
  auto a16 = [](const void* p) { return ((uintptr_t)p & 15) == 0; };
  g.vec4 = g.ldc % 4 == 0 && a16(g.C) && g.sC % 4 == 0 &&
           (!g.aux || (g.ldaux % 4 == 0 && a16(g.aux))) &&
           (!g.bias || (a16(g.bias) && g.sBias % 4 == 0)) &&
           (!g.cs || (a16(g.cs) && g.sCs % 4 == 0));
  g.tiles_m = (int)naws_cdiv(g.M, BM);
  g.tiles_n = (int)naws_cdiv(g.N, BN);
  const size_t lds = (size_t)STAGES * NPL * KS * (BM + BN) * 32;
  auto kern = gemm_x3_m16_kernel<BM, BN, WM, WN, STAGES, NPL, KS, F16, SGD>;
  if (naws_allow_lds(kern) != NAWS_OK) return NAWS_ERR_LAUNCH;
  g.batch = batch;
  if ((long long)g.tiles_m * g.tiles_n * batch > 0x7fffffffLL) return NAWS_ERR_UNSUPPORTED;
  dim3 grid((unsigned)(g.tiles_m * g.tiles_n * batch), 1, 1);
  hipLaunchKernelGGL(kern, grid, dim3(64 * WM * WN), lds, s, g);
  return naws_check_launch();
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
    int slabs_n, unsigned short* __restrict__ Pt, float* __restrict__ inv_t, int slabs_t, int batch,
    const int* __restrict__ cond, int cond_value, int prows_n) {
  __shared__ float tile[64][65];
  if (cond && *cond != cond_value) return;          // (naws_split_f16x2_rows_if)
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
      // (prows_n: the planes' own row count - a launch may cover a row range of them)
      const long long sp = (long long)slabs_n * 16 * prows_n, plane = (long long)batch * sp;
      const long long dst = bz * sp + ((long long)(c0 / 16 + s) * prows_n + (r0 + o)) * 16 + hh * 8;
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


// The same pass with 16-byte traffic on every leg (X 16-byte aligned, ld and cols multiples of 4):
// a thread fetches four float4s of the 64 x 64 tile, and owns ONE 16-element unit per operand form
// - (row o, K-slab s) of Pn, (column o, K-slab s) of Pt, o = lane, s = wave - so that a wave's
// stores are one contiguous 2 KB run per plane.  Same arithmetic per element: bit-identical planes.
__global__ __launch_bounds__(256) void split2h_dual_v4_kernel(
    const float* __restrict__ X, int rows, int cols, int ld, long long sx,
    const unsigned* __restrict__ rowmax, const unsigned* __restrict__ colmax,
    const float* __restrict__ rowmul, unsigned short* __restrict__ Pn, float* __restrict__ inv_n,
    int slabs_n, unsigned short* __restrict__ Pt, float* __restrict__ inv_t, int slabs_t, int batch,
    const int* __restrict__ cond, int cond_value, int prows_n) {
  constexpr int LDT = 68;                           // floats per tile row: 16-byte rows, banks spread
  __shared__ __attribute__((aligned(16))) float tile[64 * LDT];
  if (cond && *cond != cond_value) return;
  const int bz = blockIdx.z;
  const float* Xb = X + bz * sx;
  const int c0 = blockIdx.x * 64, r0 = blockIdx.y * 64;
  const int tid = threadIdx.x;
  {
    const int tc = (tid & 15) * 4, tr = tid >> 4;
    f32x4 v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = r0 + i * 16 + tr, c = c0 + tc;
      v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (r < rows && c < cols) v[i] = *reinterpret_cast<const f32x4*>(Xb + (long long)r * ld + c);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(&tile[(i * 16 + tr) * LDT + tc]) = v[i];
  }
  __syncthreads();
  const int o = tid & 63, s = tid >> 6;
  if (Pn && r0 + o < rows && c0 / 16 + s < slabs_n) {
    float sc, isc;
    f16x2_scales(rowmax[(long long)bz * rows + r0 + o], sc, isc);
    if (c0 == 0 && s == 0) inv_n[(long long)bz * rows + r0 + o] = isc;
    const long long sp = (long long)slabs_n * 16 * prows_n, plane = (long long)batch * sp;
    const long long dst = bz * sp + ((long long)(c0 / 16 + s) * prows_n + (r0 + o)) * 16;
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(&tile[o * LDT + s * 16 + hh * 8]);
      const f32x4 b = *reinterpret_cast<const f32x4*>(&tile[o * LDT + s * 16 + hh * 8 + 4]);
      const float t[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
      u32x4 hi4, lo4;
      split2h_pack(t, sc, hi4, lo4);
      *reinterpret_cast<u32x4*>(Pn + dst + hh * 8) = hi4;
      *reinterpret_cast<u32x4*>(Pn + plane + dst + hh * 8) = lo4;
    }
  }
  if (Pt && c0 + o < cols && r0 / 16 + s < slabs_t) {
    float sc, isc;
    f16x2_scales(colmax[(long long)bz * cols + c0 + o], sc, isc);
    if (r0 == 0 && s == 0) inv_t[(long long)bz * cols + c0 + o] = isc;
    const long long sp = (long long)slabs_t * 16 * cols, plane = (long long)batch * sp;
    const long long dst = bz * sp + ((long long)(r0 / 16 + s) * cols + (c0 + o)) * 16;
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      float t[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int kk = s * 16 + hh * 8 + e;
        t[e] = tile[kk * LDT + o] * ((rowmul && r0 + kk < rows) ? rowmul[r0 + kk] : 1.f);
      }
      u32x4 hi4, lo4;
      split2h_pack(t, sc, hi4, lo4);
      *reinterpret_cast<u32x4*>(Pt + dst + hh * 8) = hi4;
      *reinterpret_cast<u32x4*>(Pt + plane + dst + hh * 8) = lo4;
    }
  }
}

// The 16-byte form for the transposed planes alone, where it wins (tools/ab_split.py, 4000 x 8192:
// 78 vs 90 us; with both forms 54 vs 50 us).  Knob "split": 1 = always scalar, 2 = 16-byte
// whenever the operand's alignment allows.
static void launch_split2h_dual(dim3 grid, hipStream_t s, const float* X, int rows, int cols, int ld,
                                long long sx, const unsigned* rowmax, const unsigned* colmax,
                                const float* rowmul, unsigned short* Pn, float* inv_n, int slabs_n,
                                unsigned short* Pt, float* inv_t, int slabs_t, int batch,
                                const int* cond, int cond_value, int prows_n = 0) {
  if (prows_n <= 0) prows_n = rows;
  const int knob = naws_knob(NAWS_KNOB_SPLIT);
  const bool aligned = (((uintptr_t)X & 15) == 0) && ld % 4 == 0 && cols % 4 == 0 && sx % 4 == 0;
  if (aligned && knob != 1 && (knob == 2 || !Pn))
    hipLaunchKernelGGL(split2h_dual_v4_kernel, grid, dim3(256), 0, s, X, rows, cols, ld, sx, rowmax,
                       colmax, rowmul, Pn, inv_n, slabs_n, Pt, inv_t, slabs_t, batch, cond, cond_value,
                       prows_n);
  else
    hipLaunchKernelGGL(split2h_dual_kernel, grid, dim3(256), 0, s, X, rows, cols, ld, sx, rowmax,
                       colmax, rowmul, Pn, inv_n, slabs_n, Pt, inv_t, slabs_t, batch, cond, cond_value,
                       prows_n);
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
  if (N <= 64 || M <= 128) return launch_x3<128, 128, 2, 2, 2>(g, batch, s);
  // short K (the Winograd batch GEMMs): prologue and epilogue are a large share of a tile's life,
  // two 4-wave workgroups per CU overlap one's epilogue with the other's K loop
  if (K <= 1024 && g_x3_variant != 4) {
    if (g_x3_variant == 5) return launch_x3<128, 128, 2, 2, 2>(g, batch, s);
    if (g_x3_variant == 6) return launch_x3<256, 128, 2, 2, 2>(g, batch, s);     // rounds 1-3
    // round 4: the form the f16 batch GEMMs settled on in round 2 - 128 x 128 tiles, 16-deep
    // K-steps on a ring of 3 stages (72 KB with three planes: two workgroups per CU), the DMA two
    // steps ahead
    return launch_x3<128, 128, 2, 2, 3>(g, batch, s);
  }
  switch (g_x3_variant) {
    case 7: return launch_x3<256, 256, 2, 4, 3>(g, batch, s);      // round 1 / 2's form (tools/ab_x3.py)
#ifdef NAWS_AB
    case 1: return launch_x3<256, 128, 2, 2, 2>(g, batch, s);
    case 2: return launch_x3<256, 256, 2, 4, 2>(g, batch, s);
    case 3: return launch_x3<256, 128, 2, 2, 3>(g, batch, s);
    case 9: if (K % 32 == 0) return launch_x3_m16<128, 256, 2, 4, 2, 3, 2, false>(g, batch, s); break;
#endif
    default: break;
  }
  // round 3: the six-term product on v_mfma_f32_16x16x32_bf16 (256 x 128 tiles on 4 x 2 waves, two
  // 72 KB LDS stages; 256 x 256 does not fit three planes of 32-deep steps).  tools/ab_x3.py,
  // interleaved: fc6 fwd 7.08 vs 7.19 ms, fc6 wgrad 7.19 vs 7.59, fc7 1.17 / 1.20 vs 1.20 / 1.24 -
  // +2...5 %, NOT the +25 % the f16 kernel got from this shape: the bf16 32x32x16 form already ran
  // at 1.84 GHz (the f16 one had been throttled to 1.5 GHz), both land on the same power ceiling
  if (K % 32 == 0) return launch_x3_m16<256, 128, 4, 2, 2, 3, 2, false>(g, batch, s);
  return launch_x3<256, 256, 2, 4, 3>(g, batch, s);
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
  launch_split2h_dual(grid, (hipStream_t)stream, X, rows, cols, ld,
                     (long long)strideX, (const unsigned*)rowmax, (const unsigned*)colmax, rowmul,
                     (unsigned short*)Pn, Pn ? scales_n + (long long)batch * rows : nullptr,
                     kpad_n / 16, (unsigned short*)Pt, Pt ? scales_t + (long long)batch * cols : nullptr,
                     kpad_t / 16, batch, (const int*)nullptr, 0);
  return naws_check_launch();
}

// The row-scaled planes from given maxima, conditionally: every workgroup leaves at once unless
// *cond == cond_value (the device-side fallback of naws_acm_sgd_update_f16x2).
extern "C" int naws_split_f16x2_rows_if(const float* X, int batch, int rows, int cols, int ld,
                                        int64_t strideX, const uint32_t* rowmax, void* P,
                                        float* inv_scale, int kpad, const int32_t* cond,
                                        int32_t cond_value, void* stream) {
  if (batch <= 0 || rows <= 0 || cols <= 0 || ld < cols) return NAWS_ERR_SHAPE;
  NAWS_REQUIRE_PTR(X); NAWS_REQUIRE_PTR(rowmax); NAWS_REQUIRE_PTR(P); NAWS_REQUIRE_PTR(inv_scale);
  if (kpad != (cols + 31) / 32 * 32 || ((uintptr_t)P & 15) != 0) return NAWS_ERR_ARG;
  const long long gy = naws_cdiv(rows, 64);
  if (batch > 65535 || gy > 65535) return NAWS_ERR_UNSUPPORTED;
  dim3 grid((unsigned)naws_cdiv(cols, 64), (unsigned)gy, batch);
  launch_split2h_dual(grid, (hipStream_t)stream, X, rows, cols, ld,
                     (long long)strideX, (const unsigned*)rowmax, (const unsigned*)nullptr,
                     (const float*)nullptr, (unsigned short*)P, inv_scale, kpad / 16,
                     (unsigned short*)nullptr, (float*)nullptr, 0, batch, (const int*)cond, cond_value);
  return naws_check_launch();
}

// The same for rows [row0, row0 + rows) of an unbatched [plane_rows, cols] matrix whose planes
// P[2][kpad/16][plane_rows][16], maxima and 1/scale vectors are indexed by the matrix row: the
// conditional re-split of ONE row block (an fc6_w piece of the pipelined N > 1 update, the rows a
// rank does not own under NAWS.SHARDED_UPDATE).  X points at row 0 of the matrix.
extern "C" int naws_split_f16x2_row_range_if(const float* X, int plane_rows, int row0, int rows,
                                             int cols, int ld, const uint32_t* rowmax, void* P,
                                             float* inv_scale, int kpad, const int32_t* cond,
                                             int32_t cond_value, void* stream) {
  if (plane_rows <= 0 || rows <= 0 || cols <= 0 || ld < cols) return NAWS_ERR_SHAPE;
  if (row0 < 0 || row0 + rows > plane_rows) return NAWS_ERR_SHAPE;
  NAWS_REQUIRE_PTR(X); NAWS_REQUIRE_PTR(rowmax); NAWS_REQUIRE_PTR(P); NAWS_REQUIRE_PTR(inv_scale);
  if (kpad != (cols + 31) / 32 * 32 || ((uintptr_t)P & 15) != 0) return NAWS_ERR_ARG;
  const long long gy = naws_cdiv(rows, 64);
  if (gy > 65535) return NAWS_ERR_UNSUPPORTED;
  dim3 grid((unsigned)naws_cdiv(cols, 64), (unsigned)gy, 1);
  launch_split2h_dual(grid, (hipStream_t)stream, X + (long long)row0 * ld, rows, cols, ld, 0LL,
                     (const unsigned*)rowmax + row0, (const unsigned*)nullptr, (const float*)nullptr,
                     (unsigned short*)P + (long long)row0 * 16, inv_scale + row0, kpad / 16,
                     (unsigned short*)nullptr, (float*)nullptr, 0, 1, (const int*)cond, cond_value,
                     plane_rows);
  return naws_check_launch();
}

extern "C" int naws_split_f16x2(const float* X, int batch, int rows, int cols, int ld,
                                int64_t strideX, int transpose, int kpad, void* P, float* scales,
                                void* stream) {
  return naws_split_f16x2_kscaled(X, batch, rows, cols, ld, strideX, transpose, kpad, P, scales,
                                  nullptr, stream);
}

#define g_h2_variant naws_knob(NAWS_KNOB_H2)

static int gemm_f32_f16x2_nt_impl(int M, int N, int K, const void* A2, int64_t slabA,
                                  int64_t planeA, const float* scaleA, const void* B2,
                                  int64_t slabB, int64_t planeB, const float* scaleB,
                                  float* C, int ldc, int batch, int64_t strideA,
                                  int64_t strideB, int64_t strideC, int64_t strideScaleA,
                                  int64_t strideScaleB, int epilogue, const float* bias,
                                  int64_t strideBias, const float* aux, int ldaux,
                                  float alpha, float drop_ratio, uint64_t seed,
                                  int accumulate, uint32_t* rowmax, int rowmax_seg_cols,
                                  uint32_t* colmax, const float* colmax_rowmul,
                                  int drop_ld, int drop_col0, void* stream) {
  if (M <= 0 || N <= 0 || K <= 0 || batch <= 0) return NAWS_ERR_SHAPE;
  if (drop_ld < 0 || drop_col0 < 0 || (drop_ld > 0 && drop_col0 + N > drop_ld)) return NAWS_ERR_ARG;
  if (drop_ld > 0 && batch != 1) return NAWS_ERR_UNSUPPORTED;
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
  g.drop_ld = drop_ld; g.drop_c0 = drop_col0;
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
  // (a column range of a wider product takes the kernel the full-width launch would take: the
  // pieces are then bit-identical to it at every size, not only where both land on one form)
  const int Nd = drop_ld > 0 ? drop_ld : N;
  if (Nd <= 64 || M <= 128) return launch_x3<128, 128, 2, 2, 2, 2, 2, true>(g, batch, s);
  // fewer 256x256 tiles than CUs (the column remainder of a wgrad split into whole waves); the
  // batched short-K products (Winograd: 16 or 36 frequencies) choose among the forms below
  if (naws_cdiv(M, 256) * naws_cdiv(Nd, 256) * batch < 256 && g_h2_variant != 5 &&
      !(batch >= 16 && K <= 1024))
    return launch_x3<128, 128, 2, 2, 2, 2, 2, true>(g, batch, s);
  // short K (the Winograd batch GEMMs): two 4-wave workgroups per CU overlap one's prologue /
  // epilogue with the other's K loop
  if (K <= 1024 && g_h2_variant != 5) {
#ifdef NAWS_AB   // the forms measured against the default (tools/ab_h2.py): A/B build only
    if (g_h2_variant == 6) return launch_x3<256, 128, 2, 2, 2, 2, 1, true>(g, batch, s);
    if (g_h2_variant == 12) return launch_x3<128, 128, 2, 2, 4, 2, 1, true>(g, batch, s);
    if (g_h2_variant == 14) return launch_x3<128, 128, 2, 2, 3, 2, 2, true>(g, batch, s);
    if (g_h2_variant == 15) return launch_x3_m16<128, 128, 2, 2, 2, 2, 2, true>(g, batch, s);
#endif
    if (g_h2_variant == 17) return launch_x3<128, 128, 2, 2, 2, 2, 2, true>(g, batch, s);
    // (round 6, the 36-frequency products of F(4x4), kernel trace of tools/ab_wino4.py: this form
    // 51.1 us, the 16x16x32 shape on 2 x 32-deep stages 52.1, 2 x 32 of this shape 62; at
    // 150 x 250 tiles 186 vs 191)
    // 16-deep K-steps on a ring of 3 stages (48 KB: still two workgroups per CU): the DMA runs two
    // steps ahead instead of one, which is what a 16..32-step tile needs (tools/ab_h2.py, the
    // Winograd batch GEMM of conv4_2: 0.101 vs 0.114 ms with 32-deep steps and 2 stages)
    return launch_x3<128, 128, 2, 2, 3, 2, 1, true>(g, batch, s);
  }
  switch (g_h2_variant) {
#ifdef NAWS_AB   // superseded tile / stage / MFMA-shape forms: A/B build only
    case 1: return launch_x3<256, 256, 2, 4, 3, 2, 1, true>(g, batch, s);
    case 2: return launch_x3<256, 256, 2, 4, 4, 2, 1, true>(g, batch, s);
    case 3: return launch_x3<256, 128, 2, 2, 2, 2, 2, true>(g, batch, s);
    case 4: return launch_x3<256, 128, 2, 2, 3, 2, 2, true>(g, batch, s);
    case 7: return launch_x3_m16<256, 256, 2, 4, 2, 2, 2, true>(g, batch, s);
    case 11: return launch_x3_m16<256, 128, 2, 2, 3, 2, 2, true>(g, batch, s);
    case 9: return launch_x3<256, 256, 2, 4, 2, 2, 2, true>(g, batch, s);
#endif
    // 16x16x32 MFMAs, 4 x 2 waves (64 x 128 per wave): the chip holds a higher clock on this shape
    // in the power-limited long-K GEMMs (tools/ab_h2.py, interleaved: fc6 fwd 3.62 vs 3.83 ms)
    default: return launch_x3_m16<256, 256, 4, 2, 2, 2, 2, true>(g, batch, s);
  }
}

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
  return gemm_f32_f16x2_nt_impl(M, N, K, A2, slabA, planeA, scaleA, B2, slabB, planeB, scaleB, C, ldc,
                                batch, strideA, strideB, strideC, strideScaleA, strideScaleB,
                                epilogue, bias, strideBias, aux, ldaux, alpha, drop_ratio, seed,
                                accumulate, rowmax, rowmax_seg_cols, colmax, colmax_rowmul, 0, 0,
                                stream);
}

// One COLUMN RANGE [drop_col0, drop_col0 + N) of a drop_ld-wide product (unbatched): B2, scaleB,
// bias, colmax and C point at the range's first weight row / column; the dropout counters are
// those of the full-width launch, so the ranges of one activation may be produced by separate
// launches (fc6 forward cut along the weight rows: each piece starts as soon as ITS rows of
// fc6_w have been exchanged and updated - the pipelined N > 1 step) with bit-identical results.
extern "C" int naws_gemm_f32_f16x2_nt_cols(int M, int N, int K, const void* A2, int64_t slabA,
                                           int64_t planeA, const float* scaleA, const void* B2,
                                           int64_t slabB, int64_t planeB, const float* scaleB,
                                           float* C, int ldc, int epilogue, const float* bias,
                                           float drop_ratio, uint64_t seed, uint32_t* rowmax,
                                           int rowmax_seg_cols, uint32_t* colmax, int drop_ld,
                                           int drop_col0, void* stream) {
  if (drop_ld <= 0) return NAWS_ERR_ARG;
  return gemm_f32_f16x2_nt_impl(M, N, K, A2, slabA, planeA, scaleA, B2, slabB, planeB, scaleB, C, ldc,
                                1, 0, 0, 0, 0, 0, epilogue, bias, 0, nullptr, 0, 1.f, drop_ratio, seed,
                                0, rowmax, rowmax_seg_cols, colmax, nullptr, drop_ld, drop_col0,
                                stream);
}

#ifdef NAWS_AB   // Winograd frequency-column batch GEMM: A/B build only (tools/ab_wino_col.py)
int naws_wino_col_gemm_impl(int P, int Cout, int Cin, const void* V2, const float* invV,
                            const void* U2, const float* scaleU, float* S, hipStream_t stream) {
  if (P <= 0 || Cout <= 0 || Cin <= 0 || Cin % 16 != 0) return NAWS_ERR_SHAPE;
  if ((long long)(P - 1) * Cout + Cout > 0x7fffffffLL) return NAWS_ERR_UNSUPPORTED;
  XArgs g{};
  g.A = (const unsigned short*)V2; g.B = (const unsigned short*)U2; g.C = S;
  g.M = P; g.N = Cout; g.K = 4 * Cin; g.ldc = Cout;
  g.slabA = (long long)P * 16; g.slabB = (long long)Cout * 16;
  g.planeA = (long long)16 * P * Cin; g.planeB = (long long)16 * Cout * Cin;
  g.sA = (long long)4 * Cin * P; g.sB = (long long)4 * Cin * Cout;
  g.sC = (long long)2 * P * Cout; g.sC2 = (long long)P * Cout;
  g.col_steps = Cin / 16;
  g.rs = invV; g.cs = scaleU; g.sRs = 0; g.sCs = Cout;
  g.epilogue = NAWS_EPI_NONE; g.alpha = 1.f; g.drop_scale = 1.f;
  // tile forms of the column GEMM (knob "wino", tools/ab_wino_col.py): 8 = 64 x 128 on 1 x 2 waves
  // (twice the workgroups: 608 per image), 9 = 128 x 64 on 2 x 1 waves
  const int wv = naws_knob(NAWS_KNOB_WINO);
  if (wv == 8) return launch_x3<64, 128, 1, 2, 3, 2, 1, true, true>(g, 4, stream);
  if (wv == 9) return launch_x3<128, 64, 2, 1, 3, 2, 1, true, true>(g, 4, stream);
  return launch_x3<128, 128, 2, 2, 3, 2, 1, true, true>(g, 4, stream);
}
#endif  // NAWS_AB

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

// naws_gemm_bf16_slab_nt with param's SGD update in place of the store (the bf16 plan's fc6_w
// weight gradient at one process: the product A B^T [M x N] is the gradient of param [M][ldp],
// never written): param / momentum_buf updated in place exactly as naws_acm_sgd_update_planes
// (format NAWS_PLANES_BF16) would from that gradient, the updated rows rounded into P, param's
// bf16 operand plane [N/16][prows][16] (P points at this block's first row).  M: any; N % 16 == 0;
// K % 64 == 0; ldp % 4 == 0.
extern "C" int naws_gemm_bf16_slab_nt_sgd(int M, int N, int K, const void* A, int64_t slabA,
                                          const void* B, int64_t slabB, float* momentum_buf,
                                          float* param, int ldp, const float* lr, float lr_mult,
                                          float weight_decay, float momentum, int nesterov,
                                          int gpu_num, int64_t iter_count, void* P, int prows,
                                          void* stream) {
  if (M <= 0 || N <= 0 || K <= 0 || gpu_num <= 0 || prows < M || ldp < N) return NAWS_ERR_SHAPE;
  NAWS_REQUIRE_PTR(A); NAWS_REQUIRE_PTR(B); NAWS_REQUIRE_PTR(momentum_buf); NAWS_REQUIRE_PTR(param);
  NAWS_REQUIRE_PTR(lr); NAWS_REQUIRE_PTR(P);
  if (slabA < (int64_t)M * 16 || slabB < (int64_t)N * 16) return NAWS_ERR_SHAPE;
  if (K % 64 != 0 || N % 16 != 0 || ldp % 4 != 0 || slabA % 8 != 0 || slabB % 8 != 0) return NAWS_ERR_ARG;
  if ((((uintptr_t)A | (uintptr_t)B | (uintptr_t)momentum_buf | (uintptr_t)param | (uintptr_t)P) & 15) != 0)
    return NAWS_ERR_ARG;
  if (iter_count < 0) return NAWS_ERR_ARG;
  XArgs g{};
  g.A = (const unsigned short*)A; g.B = (const unsigned short*)B; g.C = nullptr;
  g.M = M; g.N = N; g.K = K; g.ldc = ldp;
  g.slabA = slabA; g.slabB = slabB;
  g.mom = momentum_buf; g.param = param; g.ldp = ldp; g.lr = lr; g.lr_mult = lr_mult;
  g.wd = weight_decay; g.momentum = momentum; g.gscale = (float)(1.0 / (double)gpu_num);
  g.nesterov = nesterov; g.first = iter_count == 0 ? 1 : 0;
  g.P = (unsigned short*)P; g.prows = prows;
  return launch_x3_m16<256, 256, 4, 2, 2, 1, 4, false, true>(g, 1, (hipStream_t)stream);
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
  if (g_h2_variant == 9) return launch_x3<256, 256, 2, 4, 2, 1, 4>(g, batch, s);
  return launch_x3_m16<256, 256, 4, 2, 2, 1, 4, false>(g, batch, s);
}

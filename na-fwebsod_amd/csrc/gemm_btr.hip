// fc6 weight gradient on the fp16x2 plan, reading the pooled features in their FORWARD layout.
//
//   dW[m][n] = sum_r dZ[r][m] x[r][n]        m < 8192 (both branches), n < 25088, r = proposal
//
// A = the transposing split of dZ (planes [2][R/16][M][16], K = r contiguous), as before.  B used
// to be a second copy of x: the RoIPool kernel writes x as the fc6 FORWARD operand, planes
// [2][n/16][r][16] (16 features of one proposal contiguous), and planes_transpose_kernel rewrote
// all 0.8 GB of it K-contiguous ([2][r/16][n][16]) once per step.  Here the GEMM takes the forward
// planes as they are: for a K-step of 32 proposals and a block of 16 features the forward layout
// holds 32 x 32 B = one contiguous KB - one LDS-DMA wave instruction, as many pieces per step as
// the K-contiguous form needs - and `ds_read_b64_tr_b16` reads it column-wise: per 16-lane group
// a block of 4 proposals x 16 features arrives feature-major, which IS the k-group of the
// 16x16x32 B operand (lane = feature, 4 consecutive k per read, two reads per fragment).
// LDS image of a (plane, 16-feature block): [32 proposal slots][32 B]; slot = proposal with bits 2
// and 3 swapped, so that the two 16-lane groups of a 32-lane half (k-groups kg, kg + 1) hit
// disjoint bank halves - conflict-free.  The swap is applied in the DMA's global source address
// (the DMA writes LDS lane-linearly).  Same tile (256 x 256, 4 x 2 waves of 64 x 128), ring,
// MFMA order and epilogue scaling as gemm_x3_m16_kernel: the accumulation order over k is the
// same, results are bit-identical to the transposed-copy route.
//
// replaces: FCGradient's dW for fc6 (reference detectron/modeling/wsl_heads.py:674-679 via
// Caffe2 FCGradient), with naws_f16_planes_transpose no longer on the path.
#include "x3_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short i16x4 __attribute__((ext_vector_type(4)));

struct BArgs {
  const unsigned short* A;   // planes [2][K/16][M][16]
  const unsigned short* X;   // planes [2][N/16][xrows][16]
  float* C;
  int M, N, K, ldc, xrows;
  long long planeA, slabA, planeX, slabX;
  const float* rs;           // per-row factors undoing A's scaling
  const float* cs;           // per-column factors (null: 1)
  int tiles_m, tiles_n;
  // SGD form (naws_gemm_f32_f16x2_nt_xk_sgd): C is never written; the tile's products are the
  // gradient of param[M][ldp] and go straight into the update
  float* mom;
  float* param;
  int ldp;
  const float* lr;           // device scalar: the base learning rate
  float lr_mult, wd, momentum, gscale;
  int nesterov, first;
  unsigned short* P;         // param's fp16x2 operand planes [2][N/16][prows][16] (hi, then lo)
  long long planeP;
  int prows;
  const unsigned* bound;     // [M] max|param row| before this update (bit patterns)
  unsigned* rowmax;          // [M] max|param row| after it (atomic max; caller zeroes)
  float* inv_scale;          // [M]
  int* overflow;
  int overflow_tag;
};

// s_waitcnt lgkmcnt(0) that the fragments' consumers depend on (the asm reads are invisible to
// the compiler's own wait insertion)
template <int TJ>
__device__ __forceinline__ void frag_fence(f16x8 (&b)[TJ], bool wait) {
  static_assert(TJ == 8 || TJ == 4, "fragments per plane");
  if constexpr (TJ == 8) {
    if (wait)
      asm volatile("s_waitcnt lgkmcnt(0)"
                   : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(b[4]), "+v"(b[5]), "+v"(b[6]), "+v"(b[7]));
    else
      asm volatile("" : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(b[4]), "+v"(b[5]), "+v"(b[6]), "+v"(b[7]));
  } else {
    if (wait)
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]));
    else
      asm volatile("" : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]));
  }
}

constexpr int KS = 2, NPL = 2, STAGES = 2, NQ = NPL * KS;

// <256, 256, 4, 2>: 512 threads, waves of 64 x 128;  <128, 128, 2, 2>: 256 threads, waves of 64 x 64
// (the last column tiles of a problem whose tile count is not a multiple of the CU count)
template <int BM, int BN, int WM, int WN, bool SGD = false>
__global__ __launch_bounds__(64 * WM * WN, (WM * WN >= 8 ? 1 : 2)) void gemm_h2_btr_kernel(BArgs g) {
  constexpr int NT = 64 * WM * WN, NW = WM * WN;
  constexpr int WTM = BM / WM, WTN = BN / WN, TI = WTM / 16, TJ = WTN / 16;
  constexpr int A_PLANE = BM * 32, B_PLANE = BN * 32;
  constexpr int STAGE = NQ * (A_PLANE + B_PLANE);
  static_assert(BM == NT / 2, "one DMA round = one (plane, slab) of the A tile");
  static_assert(BN / 16 == 2 * NW && WTN / 16 == NW, "two rounds of NW feature blocks per plane; a wave's columns = one round");
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

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid / WN, wn = wid % WN;
  const int l15 = lane & 15, kg = lane >> 4;

  // A: thread -> (row tid >> 1, k-half tid & 1) of a 256-row plane-slab
  const unsigned short* srcA = g.A + (long long)min(m0 + (tid >> 1), g.M - 1) * 16 + (tid & 1) * 8;
  // B: a workgroup round = NW feature blocks x 1 KB; wave -> feature block, lane -> 16-B chunk c of
  // the block's LDS image = (slot c >> 1, feature half c & 1); slot -> proposal by the bit swap
  const int slot = lane >> 1;
  const int prop = (slot & 0x13) | ((slot & 4) << 1) | ((slot & 8) >> 1);
  const int nfb = g.N / 16;
  const unsigned short* srcX[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks)
    srcX[ks] = g.X + (long long)min(n0 / 16 + ks * NW + wid, nfb - 1) * g.slabX + (lane & 1) * 8;

  auto issue = [&](int t, int st) {
    unsigned char* base = smx + st * STAGE + wid * 1024;
    const long long xr = (long long)min(t * 32 + prop, g.xrows - 1) * 16;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int pl = q / KS, ks = q % KS;
      __builtin_amdgcn_global_load_lds(
          NAWS_GLB_PTR(srcA + pl * g.planeA + (long long)(t * KS + ks) * g.slabA),
          NAWS_LDS_PTR(base + q * A_PLANE), 16, 0, 0);
      __builtin_amdgcn_global_load_lds(NAWS_GLB_PTR(srcX[ks] + pl * g.planeX + xr),
                                       NAWS_LDS_PTR(base + NQ * A_PLANE + q * B_PLANE), 16, 0, 0);
    }
  };

  f32x4 acc[TI][TJ];
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int j = 0; j < TJ; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.f;

  const int rd_a = (wm * WTM + l15) * 32 + (kg & 1) * 16 + (kg >> 1) * A_PLANE;
  // transposed read: lane 4q + p of a 16-lane group addresses slot row q, features 4p .. 4p + 3;
  // k-group kg, first / second half of its 8 proposals -> slot (kg >> 1) * 16 + hh * 8 + (kg & 1) * 4 + q
  const int tq = (lane & 15) >> 2, tp = lane & 3;
  const int rd_b = NQ * A_PLANE + wn * B_PLANE + ((kg >> 1) * 16 + (kg & 1) * 4 + tq) * 32 + tp * 8;

  const int T = g.K / 32;
  issue(0, 0);
  int st_cur = 0;
  for (int t = 0; t < T; ++t) {
    wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    if (t + 1 < T) issue(t + 1, st_cur ^ 1);
    const unsigned char* st = smx + st_cur * STAGE;
    // (inline asm: behind the builtin form hipcc puts s_waitcnt vmcnt(0) - it cannot tell the
    // transposed read from the LDS-DMA's destination - which waits out the NEXT step's DMA in
    // every step; the fragments' own latency is retired by frag_fence below)
    f16x8 b[NPL][TJ];
    const unsigned bbase = (unsigned)(size_t)NAWS_LDS_PTR(st + rd_b);
#pragma unroll
    for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
      for (int j = 0; j < TJ; ++j) {
        const unsigned p = bbase + pl * (KS * B_PLANE) + j * 1024;
        i16x4 lo, hi;
        asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo) : "v"(p));
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:256" : "=v"(hi) : "v"(p));
        typedef short i16x8 __attribute__((ext_vector_type(8)));
        const i16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        b[pl][j] = *reinterpret_cast<const f16x8*>(&v);
      }
    frag_fence<TJ>(b[0], true);
    frag_fence<TJ>(b[1], false);
#pragma unroll
    for (int ih = 0; ih < 2; ++ih) {
      f16x8 a[NPL][TI / 2];
#pragma unroll
      for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
        for (int i = 0; i < TI / 2; ++i)
          a[pl][i] = *reinterpret_cast<const f16x8*>(st + rd_a + pl * (KS * A_PLANE) +
                                                     (ih * (TI / 2) + i) * 512);
#define NAWS_BTR_TERM(P, Q)                                                                       \
  _Pragma("unroll") for (int i = 0; i < TI / 2; ++i) _Pragma("unroll") for (int j = 0; j < TJ; ++j) \
      acc[ih * (TI / 2) + i][j] =                                                                 \
          __builtin_amdgcn_mfma_f32_16x16x32_f16(b[Q][j], a[P][i], acc[ih * (TI / 2) + i][j], 0, 0, 0);
      NAWS_BTR_TERM(0, 0)
      NAWS_BTR_TERM(0, 1)
      NAWS_BTR_TERM(1, 0)
#undef NAWS_BTR_TERM
    }
    st_cur ^= 1;
  }

  // the MFMAs ran with the operands swapped (B fragment first): the accumulator block is C^T, so a
  // lane holds FOUR CONSECUTIVE COLUMNS of one row - row l15, columns kg * 4 + e - and the
  // epilogue moves 16 bytes per lane (the products and their k order are the same: bit-identical
  // to the un-swapped form, which holds four rows of one column and stores 4 bytes at a time)
  if constexpr (SGD) {
    // ---- the update in place of the store (one process, no gradient exchange between the two:
    // reference optimizer_wsl.py adds its all-reduce ops only for NUM_GPUS > 1).  g = the value
    // the plain epilogue would have stored; then exactly acm_sgd_planes_kernel's element work:
    // sgd_elem, the updated weight scaled by the row's bound-derived power of two and split into
    // the hi / lo f16 planes (a lane's four columns = 8 bytes per plane, a fragment's 16 rows x
    // 32 bytes = one contiguous 512-byte run of the K-slab), max|w| folded over the wave's
    // columns, one guarded atomic per (row, wave) and the overflow word.
    const float LR = g.lr[0] * g.lr_mult;
#pragma unroll
    for (int i = 0; i < TI; ++i) {
      const int row = m0 + wm * WTM + i * 16 + l15;
      const bool row_on = row < g.M;
      const int rr = row_on ? row : g.M - 1;
      const float rsv = g.rs[rr];
      const unsigned bb = g.bound[rr];
      const unsigned b2 = ((bb >> 23) >= 1u && (bb >> 23) < 0xfeu) ? bb + (1u << 23) : bb;
      float sc, isc;
      naws_f16x2_scales(b2, sc, isc);
      if (row_on && n0 + wn * WTN == 0 && kg == 0) g.inv_scale[row] = isc;
      const long long prow = (long long)rr * 16;
      float mx = 0.f;
      bool bad = false;
      // all of the row group's parameter / momentum loads first (16 x 16 bytes in flight per
      // lane), then the arithmetic and the stores: fragment by fragment, every load waited
      // behind the previous fragment's stores (0.73 ms on the fc6 problem instead of 0.3)
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
        if (!row_on || col >= g.N) continue;
        f32x4 v = acc[i][j];
        if (g.cs) {
          const f32x4 c4 = *reinterpret_cast<const f32x4*>(g.cs + col);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = v[e] * rsv * c4[e];
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = v[e] * rsv;
        }
        const long long o = (long long)row * g.ldp + col;
        f32x4 p = pw[j], m = pm[j];
        unsigned short hq[4], lq[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float me = m[e], pe = p[e];
          sgd_elem(v[e], me, pe, g.gscale, g.wd, LR, g.momentum, g.nesterov);
          m[e] = me; p[e] = pe;
          mx = fmaxf(mx, fabsf(pe));
          bad = bad || (pe != pe);
          const float t = pe * sc;
          const _Float16 hi = (_Float16)t;
          float rem = t - (float)hi;
          if (!(fabsf(t) <= 65504.f)) rem = 0.f;         // NaN / overflow live in the hi plane only
          const _Float16 lo = (_Float16)rem;
          hq[e] = *reinterpret_cast<const unsigned short*>(&hi);
          lq[e] = *reinterpret_cast<const unsigned short*>(&lo);
        }
        *reinterpret_cast<f32x4*>(g.mom + o) = m;
        *reinterpret_cast<f32x4*>(g.param + o) = p;
        const long long po = (long long)(col >> 4) * g.prows * 16 + prow + (col & 15);
        *reinterpret_cast<uint2*>(g.P + po) =
            make_uint2(hq[0] | ((unsigned)hq[1] << 16), hq[2] | ((unsigned)hq[3] << 16));
        *reinterpret_cast<uint2*>(g.P + g.planeP + po) =
            make_uint2(lq[0] | ((unsigned)lq[1] << 16), lq[2] | ((unsigned)lq[3] << 16));
      }
      // (a NaN weight must reach the overflow test: fmaxf drops NaNs, so it travels as +inf)
      if (bad) mx = __uint_as_float(0x7f800000u);
      mx = fmaxf(mx, __shfl_xor(mx, 16));
      mx = fmaxf(mx, __shfl_xor(mx, 32));
      if (kg == 0 && row_on) {
        const bool is_inf = __float_as_uint(mx) == 0x7f800000u;
        if (mx > 0.f && !is_inf) naws_atomic_max_bits(g.rowmax + row, mx);
        if (!(mx <= __uint_as_float(b2)) || is_inf) atomicMax(g.overflow, g.overflow_tag);
      }
    }
    return;
  }
#pragma unroll
  for (int i = 0; i < TI; ++i) {
    const int row = m0 + wm * WTM + i * 16 + l15;
    if (row >= g.M) continue;
    const float rsv = g.rs[row];
    float* crow = g.C + (long long)row * g.ldc;
#pragma unroll
    for (int j = 0; j < TJ; ++j) {
      const int col = n0 + wn * WTN + j * 16 + kg * 4;
      if (col >= g.N) continue;                      // N % 16 == 0: the four columns are all in or out
      f32x4 v = acc[i][j];
      if (g.cs) {
        const f32x4 c4 = *reinterpret_cast<const f32x4*>(g.cs + col);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = v[e] * rsv * c4[e];        // powers of two: exact
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = v[e] * rsv;
      }
      *reinterpret_cast<f32x4*>(crow + col) = v;
    }
  }
}

template <int BM, int BN, int WM, int WN, bool SGD = false>
int launch_btr(BArgs& g, hipStream_t s) {
  g.tiles_m = (int)naws_cdiv(g.M, BM);
  g.tiles_n = (int)naws_cdiv(g.N, BN);
  auto kern = gemm_h2_btr_kernel<BM, BN, WM, WN, SGD>;
  if (naws_allow_lds(kern) != NAWS_OK) return NAWS_ERR_LAUNCH;
  hipLaunchKernelGGL(kern, dim3((unsigned)(g.tiles_m * g.tiles_n)), dim3(64 * WM * WN),
                     (size_t)STAGES * NQ * (BM + BN) * 32, s, g);
  return naws_check_launch();
}

}  // namespace

// C [M x N] (ld ldc) = A^T-planes x X-planes: A2 = planes [2][K/16][M][16] with per-row factors
// scaleA; X2 = planes [2][N/16][xrows][16] (the forward operand of an [xrows x N] matrix, e.g.
// naws_roi_pool_f_f16x2_fwd's output), k = its row index, scaleX per column (nullable: ones).
// K % 32 == 0 (A's planes zero-padded beyond the xrows valid proposals), N % 16 == 0.
extern "C" int naws_gemm_f32_f16x2_nt_xk(int M, int N, int K, const void* A2, int64_t slabA,
                                         int64_t planeA, const float* scaleA, const void* X2,
                                         int64_t slabX, int64_t planeX, int xrows,
                                         const float* scaleX, float* C, int ldc, void* stream) {
  if (M <= 0 || N <= 0 || K <= 0 || xrows <= 0 || xrows > K) return NAWS_ERR_SHAPE;
  if (K % 32 != 0 || N % 16 != 0 || ldc < N || ldc % 4 != 0) return NAWS_ERR_UNSUPPORTED;
  NAWS_REQUIRE_PTR(A2); NAWS_REQUIRE_PTR(X2); NAWS_REQUIRE_PTR(scaleA); NAWS_REQUIRE_PTR(C);
  if ((((uintptr_t)A2 | (uintptr_t)X2 | (uintptr_t)C) & 15) != 0) return NAWS_ERR_ARG;
  if (scaleX && ((uintptr_t)scaleX & 15) != 0) return NAWS_ERR_ARG;
  if (slabA < (int64_t)M * 16 || slabX < (int64_t)xrows * 16) return NAWS_ERR_ARG;
  BArgs g{};
  g.A = (const unsigned short*)A2; g.X = (const unsigned short*)X2; g.C = C;
  g.M = M; g.N = N; g.K = K; g.ldc = ldc; g.xrows = xrows;
  g.planeA = planeA; g.slabA = slabA; g.planeX = planeX; g.slabX = slabX;
  g.rs = scaleA; g.cs = scaleX;
  hipStream_t s = (hipStream_t)stream;
  // few tiles (the column remainder of fc6's dW): 128 x 128 tiles, two workgroups per CU
  if (naws_cdiv(M, 256) * naws_cdiv(N, 256) < 256) return launch_btr<128, 128, 2, 2>(g, s);
  return launch_btr<256, 256, 4, 2>(g, s);
}

// The same product with the ACM SGD update of `param` (an [M x N] block, ld ldp, of the parameter
// arena; `mom` its momentum) in the epilogue instead of the store: the gradient
// g = A^T-planes x X-planes never reaches memory.  For a run WITHOUT a gradient exchange (one
// process: the reference adds its all-reduce ops only when NUM_GPUS > 1,
// detectron/modeling/optimizer_wsl.py:52-72); with more ranks the gradient must be written,
// reduced and then applied (naws_acm_sgd_update_f16x2).  Element arithmetic, scale bound, maxima
// and overflow word exactly as naws_acm_sgd_update_f16x2 applies them to a region (head_ops.hip):
// parameters, momentum and planes come out bit-identical to the two-kernel route.
// replaces: FCGradient's dW for fc6 + ACMWeightDecayMomentumSGDUpdate on fc6_w
// (reference detectron/ops/acm_weightdecay_momentum_sgd_op.h:72-109), ITER_SIZE 1.
extern "C" int naws_gemm_f32_f16x2_nt_xk_sgd(
    int M, int N, int K, const void* A2, int64_t slabA, int64_t planeA, const float* scaleA,
    const void* X2, int64_t slabX, int64_t planeX, int xrows, const float* scaleX, float* param,
    float* momentum_buf, int ldp, const float* lr, float lr_mult, float weight_decay, float momentum,
    int nesterov, int gpu_num, int64_t iter_count, void* planes, int64_t plane_stride,
    int plane_rows, const uint32_t* bound, uint32_t* rowmax, float* inv_scale, int32_t* overflow,
    int32_t overflow_tag, void* stream) {
  if (M <= 0 || N <= 0 || K <= 0 || xrows <= 0 || xrows > K || gpu_num <= 0 || iter_count < 0 ||
      plane_rows <= 0)
    return NAWS_ERR_SHAPE;
  if (K % 32 != 0 || N % 16 != 0 || ldp < N || ldp % 4 != 0) return NAWS_ERR_UNSUPPORTED;
  NAWS_REQUIRE_PTR(A2); NAWS_REQUIRE_PTR(X2); NAWS_REQUIRE_PTR(scaleA); NAWS_REQUIRE_PTR(param);
  NAWS_REQUIRE_PTR(momentum_buf); NAWS_REQUIRE_PTR(lr); NAWS_REQUIRE_PTR(planes);
  NAWS_REQUIRE_PTR(bound); NAWS_REQUIRE_PTR(rowmax); NAWS_REQUIRE_PTR(inv_scale);
  NAWS_REQUIRE_PTR(overflow);
  if ((((uintptr_t)A2 | (uintptr_t)X2 | (uintptr_t)param | (uintptr_t)momentum_buf) & 15) != 0)
    return NAWS_ERR_ARG;
  if (((uintptr_t)planes & 7) != 0 || (const void*)bound == (const void*)rowmax) return NAWS_ERR_ARG;
  if (scaleX && ((uintptr_t)scaleX & 15) != 0) return NAWS_ERR_ARG;
  if (slabA < (int64_t)M * 16 || slabX < (int64_t)xrows * 16) return NAWS_ERR_ARG;
  if (plane_rows < M) return NAWS_ERR_ARG;      // the block lies inside one batch item of the planes
  BArgs g{};
  g.A = (const unsigned short*)A2; g.X = (const unsigned short*)X2; g.C = nullptr;
  g.M = M; g.N = N; g.K = K; g.ldc = ldp; g.xrows = xrows;
  g.planeA = planeA; g.slabA = slabA; g.planeX = planeX; g.slabX = slabX;
  g.rs = scaleA; g.cs = scaleX;
  g.mom = momentum_buf; g.param = param; g.ldp = ldp; g.lr = lr; g.lr_mult = lr_mult;
  g.wd = weight_decay; g.momentum = momentum; g.gscale = (float)(1.0 / (double)gpu_num);
  g.nesterov = nesterov; g.first = iter_count == 0 ? 1 : 0;
  g.P = (unsigned short*)planes; g.planeP = plane_stride; g.prows = plane_rows;
  g.bound = bound; g.rowmax = rowmax; g.inv_scale = inv_scale; g.overflow = overflow;
  g.overflow_tag = overflow_tag;
  hipStream_t s = (hipStream_t)stream;
  if (naws_cdiv(M, 256) * naws_cdiv(N, 256) < 256) return launch_btr<128, 128, 2, 2, true>(g, s);
  return launch_btr<256, 256, 4, 2, true>(g, s);
}

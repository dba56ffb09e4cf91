// fp32 MFMA GEMM for gfx950 (v_mfma_f32_32x32x2_f32) with fused epilogues, and
// the 3x3 implicit-GEMM convolution built on the same main loop.
//
// Serves: FC forward  Y = X W^T + b (+ReLU, +Dropout)   [A K-contig, B K-contig]
//         FC dgrad    dX = dY W  (gated by ReLU/Dropout) [A K-contig, B N-contig]
//         FC wgrad    dW = dY^T X                        [A M-contig, B N-contig]
//         conv3x3     Y[pix,Cout] = im2col(X)[pix,9Cin] Wp[Cout,9Cin]^T (NHWC)
// ref: detectron/modeling/wsl_heads.py:654-681, webly_heads.py:463-502,
//      detectron/modeling/VGG16.py:9-48 (Caffe2 FC / Conv semantics).
//
// Design (MI355X, 64-wide waves): 256-thread workgroup = 4 waves in a 2x2 grid,
// each wave owns TI x TJ tiles of 32x32 accumulators, 2 workgroups per CU.
// Operand tiles go global -> registers -> LDS (two LDS buffers, one barrier per
// K-step).  Global loads are branch-free buffer loads (out-of-range lanes use
// an out-of-bounds offset and read 0), issued right after the K-step's first
// LDS fragment reads so their issue hides in the MFMA shadow, one or two tiles
// ahead (DIST); the LDS writes of the next tile sit among the last k-group's
// MFMAs; BK = 16 keeps LDS small enough for 3-5 workgroups per CU.
// K-contiguous operands sit in LDS as [row][BK+4] (pad 4 floats => the 16-lane
// ds_read_b128 groups hit 16 distinct 16-B slots) and each lane fetches FOUR
// k-values with one ds_read_b128; M/N-contiguous operands sit as [k][row] and
// are fetched with conflict-free ds_read_b32.  Both forms use the same k
// assignment inside an 8-deep group (MFMA step s, lane half h -> k = 4h + s),
// so any A/B layout pair multiplies matching k's.
#include <stdlib.h>
#include <type_traits>
#include "naws_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

struct GemmArgs {
  const float* A;
  const float* B;
  float* C;
  int M, N, K;
  int lda, ldb, ldc;
  long long sA, sB, sC, sBias;
  unsigned bytesA, bytesB;  // extent of one batch slice (for the buffer descriptors)
  const float* bias;
  const float* aux;
  int ldaux;
  float alpha;
  unsigned drop_thr;
  float drop_scale;
  unsigned long long seed;
  int epilogue;
  int accumulate;
  int tiles_m, tiles_n;
  // implicit-GEMM conv (A gather): NHWC input [Nimg,H,W,Cin]
  int H, W, Cin, dil;
  NawsAmax am;   // |C| maxima for the consumer's fp16x2 operand split (naws_common.h)
  int ksplit;    // > 1: blockIdx.z = batch item * ksplit + K slice; C = the slice's partial product
};

constexpr int PADK = 4;                // K-contiguous LDS row pad (floats)
constexpr unsigned OOB = 0xFFFFFFF0u;  // byte offset beyond any descriptor range

template <int ROWS, int BK, bool KC, int NT>
struct TileGeom {
  static constexpr int LD = KC ? (BK + PADK) : ROWS;          // LDS leading dim
  static constexpr int FLOATS = KC ? ROWS * (BK + PADK) : BK * ROWS;
  static constexpr int VPT = ROWS * BK / 4 / NT;              // float4 per thread
  static_assert(ROWS * BK / 4 % NT == 0, "tile must divide over the workgroup");
};

__device__ __forceinline__ float4 buf_load4(__amdgpu_buffer_rsrc_t rsrc, unsigned byte_off) {
  u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)byte_off, 0, 0);
  float4 f;
  f.x = __uint_as_float(v.x); f.y = __uint_as_float(v.y);
  f.z = __uint_as_float(v.z); f.w = __uint_as_float(v.w);
  return f;
}

// Per-thread staging state of one operand tile: the K-invariant part of each
// 16-byte load's byte offset (or OOB when the row is outside the matrix).
template <int ROWS, int BK, bool KC, int NT>
struct Stage {
  using G = TileGeom<ROWS, BK, KC, NT>;
  unsigned base[G::VPT];
  unsigned kstep;  // bytes added per K-step

  __device__ __forceinline__ void init(int ld, int row0, int nrows) {
#pragma unroll
    for (int i = 0; i < G::VPT; ++i) {
      const int f = threadIdx.x + i * NT;
      if (KC) {
        const int row = f / (BK / 4), c4 = f % (BK / 4);
        const int gr = row0 + row;
        base[i] = gr < nrows ? ((unsigned)gr * (unsigned)ld + c4 * 4) * 4u : OOB;
      } else {
        const int kr = f / (ROWS / 4), m4 = f % (ROWS / 4);
        const int gr = row0 + m4 * 4;
        base[i] = gr < nrows ? ((unsigned)kr * (unsigned)ld + gr) * 4u : OOB;
      }
    }
    kstep = KC ? BK * 4u : (unsigned)BK * (unsigned)ld * 4u;
  }

  __device__ __forceinline__ void load(__amdgpu_buffer_rsrc_t rsrc, int t, int K,
                                       float4* __restrict__ r) const {
#pragma unroll
    for (int i = 0; i < G::VPT; ++i) {
      const int f = threadIdx.x + i * NT;
      const int kk = KC ? (f % (BK / 4)) * 4 : f / (ROWS / 4);
      const bool ok = (base[i] != OOB) && (t * BK + kk < K);
      r[i] = buf_load4(rsrc, ok ? base[i] + (unsigned)t * kstep : OOB);
    }
  }

  __device__ __forceinline__ static void store(float* __restrict__ S,
                                               const float4* __restrict__ r) {
#pragma unroll
    for (int i = 0; i < G::VPT; ++i) {
      const int f = threadIdx.x + i * NT;
      if (KC) {
        const int row = f / (BK / 4), c4 = f % (BK / 4);
        *reinterpret_cast<float4*>(S + row * G::LD + c4 * 4) = r[i];
      } else {
        const int kr = f / (ROWS / 4), m4 = f % (ROWS / 4);
        *reinterpret_cast<float4*>(S + kr * G::LD + m4 * 4) = r[i];
      }
    }
  }
};

// conv A tile: rows are output pixels, a K-step lies inside one 3x3 tap.
template <int ROWS, int BK, int NT>
struct ConvStage {
  using G = TileGeom<ROWS, BK, true, NT>;
  int y[G::VPT], x[G::VPT];
  unsigned base[G::VPT];  // byte offset of (pixel, channel chunk) or OOB

  __device__ __forceinline__ void init(int row0, int M, int H, int W, int Cin) {
#pragma unroll
    for (int i = 0; i < G::VPT; ++i) {
      const int f = threadIdx.x + i * NT;
      const int row = f / (BK / 4), c4 = f % (BK / 4);
      const int gm = row0 + row;
      if (gm < M) {
        x[i] = gm % W;
        y[i] = (gm / W) % H;
        base[i] = ((unsigned)gm * (unsigned)Cin + c4 * 4) * 4u;
      } else {
        x[i] = 0; y[i] = 0; base[i] = OOB;
      }
    }
  }

  __device__ __forceinline__ void load(__amdgpu_buffer_rsrc_t rsrc, int t, int H, int W, int Cin,
                                       int dil, float4* __restrict__ r) const {
    const int k0 = t * BK;
    const int tap = k0 / Cin, c0 = k0 - tap * Cin;
    const int dy = (tap / 3 - 1) * dil, dx = (tap % 3 - 1) * dil;
    const int delta = ((dy * W + dx) * Cin + c0) * 4;  // bytes, may be negative
#pragma unroll
    for (int i = 0; i < G::VPT; ++i) {
      const int yy = y[i] + dy, xx = x[i] + dx;
      const bool ok = (base[i] != OOB) && yy >= 0 && yy < H && xx >= 0 && xx < W;
      r[i] = buf_load4(rsrc, ok ? (unsigned)((int)base[i] + delta) : OOB);
    }
  }
};

// ---- the kernel --------------------------------------------------------------
// Occupancy target (waves per SIMD = workgroups per CU): what the LDS footprint admits, capped
// at 4 (128 VGPRs); the register allocator is held to it.
template <int BM, int BN, int BK>
constexpr int occupancy_target() {
  constexpr int lds = 2 * (BM + BN) * (BK + PADK) * 4;
  constexpr int by_lds = 160 * 1024 / lds;
  return (BM * BN >= 128 * 128 || BK != 16) ? 1 : (by_lds > 5 ? 5 : by_lds);
}

template <int BM, int BN, int BK, bool A_KC, bool B_KC, bool CONV, int WM = 2, int WN = 2>
__global__ __launch_bounds__(64 * WM * WN, (occupancy_target<BM, BN, BK>()))
void gemm_f32_kernel(GemmArgs g) {
  constexpr int NT = 64 * WM * WN;
  using GA = TileGeom<BM, BK, A_KC, NT>;
  using GB = TileGeom<BN, BK, B_KC, NT>;
  constexpr int WTM = BM / WM, WTN = BN / WN;  // per-wave output
  constexpr int TI = WTM / 32, TJ = WTN / 32;
  constexpr int NKG = BK / 8;
  // Prefetch distance in K-steps.  2 = two register sets: the loads of tile t+2 are issued
  // in step t and written to LDS in step t+1, so the wait in front of the LDS writes never
  // sees memory latency (+1..4% on the N-contiguous / conv forms, interleaved A/B).  The
  // K-contiguous x K-contiguous 128x128 form keeps one set: its 116 VGPRs admit 4 workgroups
  // per CU, which is worth more there than the longer distance (-2% with two sets).
  constexpr int DIST = (BM * BN >= 128 * 128 && A_KC && B_KC && !CONV) ? 1 : 2;
  constexpr int STASH_AT = (DIST == 1 && BK == 16) ? 2 : 0;  // MFMA step of the last k-group
  static_assert(TI >= 1 && TJ >= 1, "wave tile too small");

  // One LDS array, addressed by integer offsets only: a pointer table indexed
  // by the buffer parity degrades to FLAT accesses, whose waits also drain the
  // global prefetch.
  extern __shared__ __attribute__((aligned(16))) float sm[];
  constexpr int A_OFF = 0, B_OFF = 2 * GA::FLOATS;

  // XCD-aware tile order: consecutive logical tiles (sharing an A panel) land
  // on one XCD's L2; groups of 8 M-tiles sweep N together.
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

  // split K (naws_gemm_f32_splitk): slice ks of batch item bz walks K-steps t0 .. t0 + T and
  // stores its partial product, un-epilogued, as item blockIdx.z of the workspace
  const int ksp = g.ksplit > 1 ? g.ksplit : 1;
  const long long bz = blockIdx.z / ksp;
  const int kslice = (int)(blockIdx.z - bz * ksp);
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(g.A + bz * g.sA), 0, (int)g.bytesA, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(g.B + bz * g.sB), 0, (int)g.bytesB, 0x00020000);
  float* C = g.C + (ksp > 1 ? (long long)blockIdx.z : bz) * g.sC;

  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int wm = wid / WN, wn = wid % WN;
  const int l31 = lane & 31, h = lane >> 5;

  f32x16 acc[TI][TJ];
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int j = 0; j < TJ; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  float4 ra0[GA::VPT], rb0[GB::VPT], ra1[GA::VPT], rb1[GB::VPT];
  Stage<BM, BK, A_KC, NT> stA;
  ConvStage<BM, BK, NT> cvA;
  Stage<BN, BK, B_KC, NT> stB;
  if constexpr (CONV) cvA.init(m0, g.M, g.H, g.W, g.Cin);
  else stA.init(g.lda, m0, g.M);
  stB.init(g.ldb, n0, g.N);

  const int Tall = (g.K + BK - 1) / BK;
  const int Tc = (Tall + ksp - 1) / ksp, t0 = kslice * Tc;     // (the host keeps every slice >= 1 step)
  const int T = ksp > 1 ? min(Tc, Tall - t0) : Tall;
  auto fetch = [&](int t, float4* ra, float4* rb) {
    if constexpr (CONV) cvA.load(rsA, t0 + t, g.H, g.W, g.Cin, g.dil, ra);
    else stA.load(rsA, t0 + t, g.K, ra);
    stB.load(rsB, t0 + t, g.K, rb);
  };
  auto stash = [&](int buf, const float4* ra, const float4* rb) {
    Stage<BM, BK, A_KC, NT>::store(sm + A_OFF + buf * GA::FLOATS, ra);
    Stage<BN, BK, B_KC, NT>::store(sm + B_OFF + buf * GB::FLOATS, rb);
  };

  // One K-step on LDS buffer t&1: compute tile t, load tile t+2 into (lra, lrb), write tile
  // t+1 from (sra, srb) into the other buffer.  PIPE (compile-time) = steady state, no
  // bounds on t: the body is then a single basic block and the staging instructions stay
  // where they are written, between the MFMAs.
  auto kstep = [&](int t, auto pipe_tag, float4* lra, float4* lrb, const float4* sra,
                   const float4* srb) {
    constexpr bool PIPE = decltype(pipe_tag)::value;
    const int cur = t & 1;
    const bool load2 = PIPE || (t + DIST < T);
    const bool store1 = PIPE || (t + 1 < T);
    const float* as = sm + A_OFF + cur * GA::FLOATS;
    const float* bs = sm + B_OFF + cur * GB::FLOATS;
#pragma unroll
    for (int kg = 0; kg < NKG; ++kg) {
      float af[TI][4], bf[TJ][4];
#pragma unroll
      for (int i = 0; i < TI; ++i) {
        const int row = wm * WTM + i * 32 + l31;
        if (A_KC) {
          const float4 v = *reinterpret_cast<const float4*>(as + row * GA::LD + kg * 8 + h * 4);
          af[i][0] = v.x; af[i][1] = v.y; af[i][2] = v.z; af[i][3] = v.w;
        } else {
#pragma unroll
          for (int s = 0; s < 4; ++s) af[i][s] = as[(kg * 8 + h * 4 + s) * GA::LD + row];
        }
      }
#pragma unroll
      for (int j = 0; j < TJ; ++j) {
        const int col = wn * WTN + j * 32 + l31;
        if (B_KC) {
          const float4 v = *reinterpret_cast<const float4*>(bs + col * GB::LD + kg * 8 + h * 4);
          bf[j][0] = v.x; bf[j][1] = v.y; bf[j][2] = v.z; bf[j][3] = v.w;
        } else {
#pragma unroll
          for (int s = 0; s < 4; ++s) bf[j][s] = bs[(kg * 8 + h * 4 + s) * GB::LD + col];
        }
      }
      if (kg == 0) {
        if (load2) fetch(t + DIST, lra, lrb);
        if (PIPE) __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        if (kg == NKG - 1 && s == STASH_AT) {
          if (PIPE) __builtin_amdgcn_sched_barrier(0);
          if (store1) stash(cur ^ 1, sra, srb);
          if (PIPE) __builtin_amdgcn_sched_barrier(0);
        }
        // keep the barrier (and its wait for the LDS writes) behind most of this group's
        // MFMAs; the final ones may sink below it and cover the next step's first reads
        if (PIPE && kg == NKG - 1 && s == 3 && STASH_AT < 3) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
          for (int j = 0; j < TJ; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][s], bf[j][s], acc[i][j], 0, 0, 0);
      }
    }
    __syncthreads();
  };

  fetch(0, ra0, rb0);
  stash(0, ra0, rb0);
  if constexpr (DIST == 2) {
    if (T > 1) fetch(1, ra1, rb1);
    __syncthreads();
    int t = 0;
    for (; t + 3 < T; t += 2) {
      kstep(t, std::true_type{}, ra0, rb0, ra1, rb1);
      kstep(t + 1, std::true_type{}, ra1, rb1, ra0, rb0);
    }
    if (t < T) kstep(t, std::false_type{}, ra0, rb0, ra1, rb1);
    if (t + 1 < T) kstep(t + 1, std::false_type{}, ra1, rb1, ra0, rb0);
    if (t + 2 < T) kstep(t + 2, std::false_type{}, ra0, rb0, ra1, rb1);
  } else {
    __syncthreads();
    int t = 0;
    for (; t + 1 < T; ++t) kstep(t, std::true_type{}, ra0, rb0, ra0, rb0);
    kstep(T - 1, std::false_type{}, ra0, rb0, ra0, rb0);
  }

  // ---- epilogue: C/D map col = lane&31, row = (e&3) + 8*(e>>2) + 4*(lane>>5)
  const float* bias = g.bias ? g.bias + bz * g.sBias : nullptr;
  const float* aux = g.aux ? g.aux + bz * g.sC : nullptr;
  const int epi = g.epilogue;
#pragma unroll
  for (int j = 0; j < TJ; ++j) {
    const int col = n0 + wn * WTN + j * 32 + l31;
    if (col >= g.N) continue;
    const float bv = (bias && epi >= NAWS_EPI_BIAS && epi <= NAWS_EPI_BIAS_RELU_DROP) ? bias[col] : 0.f;
#pragma unroll
    for (int i = 0; i < TI; ++i) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = m0 + wm * WTM + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
        if (row >= g.M) continue;
        float v = acc[i][j][e] + bv;
        if (epi == NAWS_EPI_BIAS_RELU || epi == NAWS_EPI_BIAS_RELU_DROP) v = fmaxf(v, 0.f);
        if (epi == NAWS_EPI_BIAS_RELU_DROP) {
          const unsigned long long idx =
              (unsigned long long)bz * g.M * g.N + (unsigned long long)row * g.N + col;
          v = naws_keep(g.seed, idx, g.drop_thr) ? v * g.drop_scale : 0.f;
        } else if (epi == NAWS_EPI_GATE_POS) {
          v = (aux[(long long)row * g.ldaux + col] > 0.f) ? v * g.alpha : 0.f;
        }
        float* dst = C + (long long)row * g.ldc + col;
        if (g.accumulate) v += *dst;
        *dst = v;
        acc[i][j][e] = v;
      }
    }
  }
  if (g.am.rowmax || g.am.colmax)
    naws_tile_amax_32<TI, TJ>(acc, m0 + wm * WTM, n0 + wn * WTN, g.M, g.N, lane, g.am, bz);
}

// ---- C = epilogue(A B) for a SHORT inner dimension (K <= 64): dZ7 = gate(dL W8) -----------------
// With K = 2C = 40 the product has 80 flops per output element: the 128x128 MFMA tile form spends
// its life in the epilogue (dependent aux loads, 244 us for 131 MB out + 131 MB aux).  Here a
// workgroup owns 128 rows x 1024 columns (one guarded column atomic per 128 rows); a thread keeps its four columns of B - K float4s - in
// registers for all its rows, A's row (K floats, workgroup-uniform) comes through the scalar cache,
// and every output float4 is one coalesced aux load + one coalesced store: HBM-bound.  Sums run in
// k order like the MFMA chain.  Also reports |C| row / column maxima (NawsAmax).
constexpr int SK_MAXK = 64;
#ifdef NAWS_AB      // the register-resident FMA form: A/B build only (tools/bench_smallk.py)
constexpr int SK_ROWS = 128, SK_RG = 4;
template <int KMAX>
__global__ __launch_bounds__(256) void gemm_smallk_nn_kernel(GemmArgs g) {
  __shared__ float s_rm[SK_ROWS][4];
  __shared__ __attribute__((aligned(16))) float s_a[SK_ROWS][KMAX];   // the block's rows of A
  const long long bz = blockIdx.z;
  const float* A = g.A + bz * g.sA;
  const float* B = g.B + bz * g.sB;
  float* C = g.C + bz * g.sC;
  const float* aux = g.aux ? g.aux + bz * g.sC : nullptr;
  const int col = (blockIdx.x * 256 + threadIdx.x) * 4;
  const int r0 = blockIdx.y * SK_ROWS;
  const bool col_ok = col < g.N;                       // N % 4 == 0: a float4 is all in or all out
  float4 b[KMAX];
#pragma unroll
  for (int k = 0; k < KMAX; ++k)
    b[k] = (k < g.K && col_ok) ? *reinterpret_cast<const float4*>(B + (long long)k * g.ldb + col)
                               : make_float4(0.f, 0.f, 0.f, 0.f);
  for (int i = threadIdx.x; i < SK_ROWS * KMAX; i += 256) {
    const int r = i / KMAX, k = i - r * KMAX;
    s_a[r][k] = (r0 + r < g.M && k < g.K) ? A[(long long)(r0 + r) * g.lda + k] : 0.f;
  }
  __syncthreads();
  float4 cm = make_float4(0.f, 0.f, 0.f, 0.f);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const bool gate = g.epilogue == NAWS_EPI_GATE_POS;
  // rows in groups of SK_RG: the group's aux loads are in flight together, and the row maxima go
  // through LDS so that the guarded atomics are issued once, in parallel, at the end
  for (int rg = 0; rg < SK_ROWS; rg += SK_RG) {
    float4 x[SK_RG], acc[SK_RG];
#pragma unroll
    for (int j = 0; j < SK_RG; ++j) {
      const int r = r0 + rg + j;
      x[j] = make_float4(1.f, 1.f, 1.f, 1.f);
      if (gate && col_ok && r < g.M)
        x[j] = *reinterpret_cast<const float4*>(aux + (long long)r * g.ldaux + col);
      acc[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int k4 = 0; k4 < KMAX / 4; ++k4) {              // (zero-padded beyond K: adds exact zeros)
#pragma unroll
      for (int j = 0; j < SK_RG; ++j) {
        const float4 av = *reinterpret_cast<const float4*>(&s_a[rg + j][k4 * 4]);   // LDS broadcast
        const float a4[4] = {av.x, av.y, av.z, av.w};
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const float4 bb = b[k4 * 4 + u];
          acc[j].x = fmaf(a4[u], bb.x, acc[j].x); acc[j].y = fmaf(a4[u], bb.y, acc[j].y);
          acc[j].z = fmaf(a4[u], bb.z, acc[j].z); acc[j].w = fmaf(a4[u], bb.w, acc[j].w);
        }
      }
    }
#pragma unroll
    for (int j = 0; j < SK_RG; ++j) {
      const int r = r0 + rg + j;
      float rm = 0.f;
      if (col_ok && r < g.M) {
        float4 v = acc[j];
        if (gate) {
          v.x = x[j].x > 0.f ? v.x * g.alpha : 0.f; v.y = x[j].y > 0.f ? v.y * g.alpha : 0.f;
          v.z = x[j].z > 0.f ? v.z * g.alpha : 0.f; v.w = x[j].w > 0.f ? v.w * g.alpha : 0.f;
        }
        float4* dst = reinterpret_cast<float4*>(C + (long long)r * g.ldc + col);
        if (g.accumulate) {
          const float4 o = *dst;
          v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
        }
        *dst = v;
        const float cmul = g.am.colmul ? fabsf(g.am.colmul[r]) : 1.f;
        const float ax = fabsf(v.x), ay = fabsf(v.y), az = fabsf(v.z), aw = fabsf(v.w);
        rm = fmaxf(fmaxf(ax, ay), fmaxf(az, aw));
        cm.x = fmaxf(cm.x, ax * cmul); cm.y = fmaxf(cm.y, ay * cmul);
        cm.z = fmaxf(cm.z, az * cmul); cm.w = fmaxf(cm.w, aw * cmul);
      }
      if (g.am.rowmax) {
        rm = wave_max(rm);
        if (lane == 0) s_rm[rg + j][wave] = rm;
      }
    }
  }
  if (g.am.rowmax) {
    __syncthreads();
    if (threadIdx.x < SK_ROWS && r0 + threadIdx.x < g.M) {
      const float m = fmaxf(fmaxf(s_rm[threadIdx.x][0], s_rm[threadIdx.x][1]),
                            fmaxf(s_rm[threadIdx.x][2], s_rm[threadIdx.x][3]));
      unsigned* rowmax = g.am.rowmax + bz * g.am.sRow +
                         (long long)((blockIdx.x * 1024) / g.am.seg_cols) * g.M;
      if (m > 0.f) naws_atomic_max_bits(rowmax + r0 + threadIdx.x, m);
    }
  }
  if (g.am.colmax && col_ok) {
    unsigned* cmx = g.am.colmax + bz * g.am.sCol + col;
    if (cm.x > 0.f) naws_atomic_max_bits(cmx + 0, cm.x);
    if (cm.y > 0.f) naws_atomic_max_bits(cmx + 1, cm.y);
    if (cm.z > 0.f) naws_atomic_max_bits(cmx + 2, cm.z);
    if (cm.w > 0.f) naws_atomic_max_bits(cmx + 3, cm.w);
  }
}

#endif

// The same product on the fp32 MFMA (v_mfma_f32_16x16x4_f32, operands swapped so that a lane
// holds FOUR CONSECUTIVE COLUMNS of one row: 16-byte gate loads and stores).  The register-resident
// form above keeps K float4s of B per thread (256 VGPRs, one wave per SIMD) and spends 160 FMAs
// per output float4: 168 us without / 203 us with the maxima against a 58 us HBM floor
// (tools/bench_smallk.py).  Here K is 10..16 MFMA steps, both operands of a 64 x 64 wave tile are
// 2 x 40 VGPRs, and the kernel is the gate read + the store: a tile's 16 gate loads are issued
// BEFORE its MFMAs.  A workgroup owns 64 rows x 1024 columns, wave w its 256-column strip in four
// 64-column chunks; A's 64 rows sit in LDS (row stride K + 1: conflict-free fragment reads).
// Maxima: rows through LDS (one guarded atomic per row and workgroup); columns are parked in LDS
// too and leave at the end, four per thread in parallel - issued from the MFMA layout (16 per
// lane, each a dependent load - compare - atomic round trip) they cost 80 us.
typedef float sk_f32x4 __attribute__((ext_vector_type(4)));
constexpr int SKM_ROWS = 64;
template <int KS>
__global__ __launch_bounds__(256) void gemm_smallk_mfma_kernel(GemmArgs g) {
  constexpr int KP = KS * 4, LDA_S = KP + 1;
  constexpr bool EARLY = KS <= 10;                 // registers for the gate loads ahead of the MFMAs
  __shared__ float s_rm[SKM_ROWS][4];
  __shared__ float s_cm[1024];
  __shared__ float s_a[SKM_ROWS * LDA_S];
  const long long bz = blockIdx.z;
  const float* A = g.A + bz * g.sA;
  const float* B = g.B + bz * g.sB;
  float* C = g.C + bz * g.sC;
  const float* aux = g.aux ? g.aux + bz * g.sC : nullptr;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l15 = lane & 15, kg = lane >> 4;
  const int m0 = blockIdx.y * SKM_ROWS;
  const int c0 = blockIdx.x * 1024 + wave * 256;
  const bool gate = g.epilogue == NAWS_EPI_GATE_POS;
  for (int i = threadIdx.x; i < SKM_ROWS * KP; i += 256) {
    const int r = i / KP, k = i - r * KP;
    s_a[r * LDA_S + k] = (m0 + r < g.M && k < g.K) ? A[(long long)(m0 + r) * g.lda + k] : 0.f;
  }
  for (int i = threadIdx.x; i < 1024; i += 256) s_cm[i] = 0.f;
  __syncthreads();
  float a[4][KS];                                         // a[i][ks] = A[m0 + i*16 + l15][ks*4 + kg]
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) a[i][ks] = s_a[(i * 16 + l15) * LDA_S + ks * 4 + kg];
  float cmul[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = m0 + i * 16 + l15;
    cmul[i] = (g.am.colmul && row < g.M) ? fabsf(g.am.colmul[row]) : 1.f;
  }
  float rmx[4] = {0.f, 0.f, 0.f, 0.f};
  for (int ch = 0; ch < 4; ++ch) {
    const int n0 = c0 + ch * 64;
    if (n0 >= g.N) break;                                 // (wave-uniform)
    float b[4][KS];                                       // b[j][ks] = B[ks*4 + kg][n0 + j*16 + l15]
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const int k = ks * 4 + kg, n = n0 + j * 16 + l15;
        b[j][ks] = (k < g.K && n < g.N) ? B[(long long)k * g.ldb + n] : 0.f;
      }
    sk_f32x4 x[4][4];
    auto load_gate = [&](int i) {
      const int row = m0 + i * 16 + l15;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int col = n0 + j * 16 + kg * 4;
        x[i][j] = sk_f32x4{1.f, 1.f, 1.f, 1.f};
        if (gate && row < g.M && col < g.N)
          x[i][j] = *reinterpret_cast<const sk_f32x4*>(aux + (long long)row * g.ldaux + col);
      }
    };
    if constexpr (EARLY) {
#pragma unroll
      for (int i = 0; i < 4; ++i) load_gate(i);
    }
    sk_f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = sk_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[j][ks], a[i][ks], acc[i][j], 0, 0, 0);
    float cm[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) cm[j][e] = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = m0 + i * 16 + l15;
      const bool row_on = row < g.M;
      if constexpr (!EARLY) load_gate(i);
      float rm = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int col = n0 + j * 16 + kg * 4;
        if (!row_on || col >= g.N) continue;              // N % 4 == 0: a float4 is all in or all out
        sk_f32x4 v = acc[i][j];
        if (gate) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = x[i][j][e] > 0.f ? v[e] * g.alpha : 0.f;
        }
        sk_f32x4* dst = reinterpret_cast<sk_f32x4*>(C + (long long)row * g.ldc + col);
        if (g.accumulate) {
          const sk_f32x4 o = *dst;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] += o[e];
        }
        *dst = v;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float av = fabsf(v[e]);
          rm = fmaxf(rm, av);
          cm[j][e] = fmaxf(cm[j][e], av * cmul[i]);
        }
      }
      rmx[i] = fmaxf(rmx[i], rm);
    }
    if (g.am.colmax) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float v = cm[j][e];
#pragma unroll
          for (int d = 8; d > 0; d >>= 1) v = fmaxf(v, __shfl_xor(v, d));
          if (l15 == 0) s_cm[wave * 256 + ch * 64 + j * 16 + kg * 4 + e] = v;
        }
    }
  }
  if (g.am.rowmax) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float v = rmx[i];
      v = fmaxf(v, __shfl_xor(v, 16));
      v = fmaxf(v, __shfl_xor(v, 32));
      if (kg == 0) s_rm[i * 16 + l15][wave] = v;
    }
  }
  if (g.am.rowmax || g.am.colmax) __syncthreads();
  if (g.am.rowmax && threadIdx.x < SKM_ROWS && m0 + threadIdx.x < g.M) {
    const float m = fmaxf(fmaxf(s_rm[threadIdx.x][0], s_rm[threadIdx.x][1]),
                          fmaxf(s_rm[threadIdx.x][2], s_rm[threadIdx.x][3]));
    unsigned* rowmax = g.am.rowmax + bz * g.am.sRow +
                       (long long)((blockIdx.x * 1024) / g.am.seg_cols) * g.M;
    if (m > 0.f) naws_atomic_max_bits(rowmax + m0 + threadIdx.x, m);
  }
  if (g.am.colmax) {
    // thread t: columns t, t + 256, ... of the workgroup's 1024 - the four current maxima are
    // fetched together (independent loads), then compared
    unsigned* cmx = g.am.colmax + bz * g.am.sCol + blockIdx.x * 1024;
    unsigned cur[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int col = q * 256 + threadIdx.x;
      cur[q] = (blockIdx.x * 1024 + col < g.N)
                   ? __hip_atomic_load(cmx + col, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0xffffffffu;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int col = q * 256 + threadIdx.x;
      const unsigned v = __float_as_uint(s_cm[col]);
      if (v > cur[q] && s_cm[col] > 0.f) atomicMax(cmx + col, v);
    }
  }
}

// Tuning knob for A/B experiments (tools/kernel_bench.py): naws_set_variant("gemm", v)
//   0 default, 2: pad LDS so only 1 workgroup fits a CU, 4: BK=32 tile forms
int gemm_variant() { return naws_knob(NAWS_KNOB_GEMM); }

template <int BM, int BN, int BK, bool A_KC, bool B_KC, bool CONV, int WM = 2, int WN = 2>
int launch(GemmArgs& g, int batch, hipStream_t s) {
  constexpr int NT = 64 * WM * WN;
  using GA = TileGeom<BM, BK, A_KC, NT>;
  using GB = TileGeom<BN, BK, B_KC, NT>;
  g.tiles_m = (int)naws_cdiv(g.M, BM);
  g.tiles_n = (int)naws_cdiv(g.N, BN);
  size_t lds = (size_t)2 * (GA::FLOATS + GB::FLOATS) * sizeof(float);
  if (gemm_variant() == 2) lds = std::max<size_t>(lds, 84 * 1024);
  auto kern = gemm_f32_kernel<BM, BN, BK, A_KC, B_KC, CONV, WM, WN>;
  if (naws_allow_lds(kern) != NAWS_OK) return NAWS_ERR_LAUNCH;
  dim3 grid((unsigned)(g.tiles_m * g.tiles_n), 1, batch * (g.ksplit > 1 ? g.ksplit : 1));
  hipLaunchKernelGGL(kern, grid, dim3(NT), lds, s, g);
  return naws_check_launch();
}

// C[b] = epilogue(sum over the K slices, in slice order, of the partial products) - the second,
// deterministic pass of naws_gemm_f32_splitk.  One thread per element (the outputs are small:
// that is why K was split).
__global__ void splitk_reduce_kernel(const float* __restrict__ part, int ksplit, int batch, int M,
                                     int N, float* __restrict__ C, int ldc, long long sC,
                                     const float* __restrict__ bias, long long sBias) {
  const long long per = (long long)M * N, total = per * batch;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const long long b = i / per, r = (i - b * per) / N;
    const int c = (int)(i - b * per - r * N);
    const float* p = part + (b * ksplit) * per + r * N + c;
    float v = p[0];
    for (int k = 1; k < ksplit; ++k) v += p[(long long)k * per];
    if (bias) v += bias[b * sBias + c];
    C[b * sC + r * ldc + c] = v;
  }
}

// Tile choice.  BK = 16 keeps a workgroup's LDS at <= 40 KB so 3-5 workgroups share a CU:
// more independent barrier domains per SIMD fill each other's staging gaps (measured
// +3-5% over BK = 32 at 2 workgroups/CU), and finer granularity when a layer has only a
// few tiles per CU.  NAWS_GEMM_VARIANT=4 selects the BK = 32 forms for A/B runs.
template <bool A_KC, bool B_KC, bool CONV>
int dispatch(GemmArgs& g, int batch, hipStream_t s) {
  const long long t128 = naws_cdiv(g.M, 128) * naws_cdiv(g.N, 128) * batch;
  const bool bk32 = gemm_variant() == 4;
  if (g.N <= 64) {
    if (naws_cdiv(g.M, 128) * batch >= 512)
      return bk32 ? launch<128, 64, 32, A_KC, B_KC, CONV>(g, batch, s)
                  : launch<128, 64, 16, A_KC, B_KC, CONV>(g, batch, s);
    return launch<64, 64, 32, A_KC, B_KC, CONV>(g, batch, s);
  }
  // FC forward (both operands K-contiguous): a 256x128 tile on 8 waves moves 25% fewer bytes
  // per MFMA through L2/LDS, worth +1.6% there (interleaved A/B); the N-contiguous forms
  // lose 1-7% with it and keep 128x128.  NAWS_GEMM_VARIANT=5 disables it.
  // (long K only: with one or two workgroups per CU nothing covers a tile's prologue/epilogue,
  // which short-K problems such as the Winograd batch GEMMs, K = Cin, cannot amortise)
  if (!CONV && A_KC && B_KC && gemm_variant() != 5 && !bk32 && g.K >= 2048) {
    // 256x256 on 16 waves (one workgroup per CU): another +1.5% when it fills the chip twice
    if (naws_cdiv(g.M, 256) * naws_cdiv(g.N, 256) * batch >= 512)
      return launch<256, 256, 16, A_KC, B_KC, CONV, 4, 4>(g, batch, s);
    if (t128 >= 2048) return launch<256, 128, 16, A_KC, B_KC, CONV, 4, 2>(g, batch, s);
  }
  if (t128 >= 2048 || (!CONV && t128 >= 1024))
    return bk32 ? launch<128, 128, 32, A_KC, B_KC, CONV>(g, batch, s)
                : launch<128, 128, 16, A_KC, B_KC, CONV>(g, batch, s);
  if (naws_cdiv(g.M, 64) * naws_cdiv(g.N, 128) * batch >= 768)
    return bk32 ? launch<64, 128, 32, A_KC, B_KC, CONV>(g, batch, s)
                : launch<64, 128, 16, A_KC, B_KC, CONV>(g, batch, s);
  return launch<64, 64, 32, A_KC, B_KC, CONV>(g, batch, s);
}

bool aligned4(long long v) { return (v & 3) == 0; }
bool ptr16(const void* p) { return ((uintptr_t)p & 15) == 0; }

// bytes spanned by one [rows, cols] row-major slice with leading dimension ld
long long extent_bytes(long long rows, long long cols, long long ld) {
  return ((rows - 1) * ld + cols) * 4;
}
constexpr long long MAX_EXTENT = 0xFFFFFFF0LL - 64;  // 32-bit buffer byte offsets

}  // namespace

extern "C" int naws_gemm_f32_amax(int transA, int transB, int M, int N, int K, const float* A,
                                  int lda, const float* B, int ldb, float* C, int ldc, int batch,
                                  int64_t strideA, int64_t strideB, int64_t strideC, int epilogue,
                                  const float* bias, int64_t strideBias, const float* aux,
                                  int ldaux, float alpha, float drop_ratio, uint64_t seed,
                                  int accumulate, uint32_t* rowmax, int rowmax_seg_cols,
                                  uint32_t* colmax, const float* colmax_rowmul, void* stream) {
  if (M <= 0 || N <= 0 || K <= 0 || batch <= 0) return NAWS_ERR_SHAPE;
  NAWS_REQUIRE_PTR(A); NAWS_REQUIRE_PTR(B); NAWS_REQUIRE_PTR(C);
  if (epilogue < NAWS_EPI_NONE || epilogue > NAWS_EPI_GATE_POS) return NAWS_ERR_ARG;
  if (epilogue == NAWS_EPI_GATE_POS && aux == nullptr) return NAWS_ERR_NULL;
  if (epilogue == NAWS_EPI_BIAS_RELU_DROP && !(drop_ratio >= 0.f && drop_ratio < 1.f))
    return NAWS_ERR_ARG;
  if (lda < (transA ? M : K) || ldb < (transB ? K : N) || ldc < N) return NAWS_ERR_SHAPE;
  // 16-byte vector loads along each operand's contiguous dimension
  if (!aligned4(lda) || !aligned4(ldb) || !aligned4(strideA) || !aligned4(strideB) ||
      !ptr16(A) || !ptr16(B))
    return NAWS_ERR_ARG;
  if (!aligned4(transA ? M : K) || !aligned4(transB ? K : N)) return NAWS_ERR_ARG;
  if (batch > 65535) return NAWS_ERR_UNSUPPORTED;
  const long long exA = transA ? extent_bytes(K, M, lda) : extent_bytes(M, K, lda);
  const long long exB = transB ? extent_bytes(N, K, ldb) : extent_bytes(K, N, ldb);
  if (exA > MAX_EXTENT || exB > MAX_EXTENT) return NAWS_ERR_UNSUPPORTED;

  GemmArgs g{};
  g.A = A; g.B = B; g.C = C; g.M = M; g.N = N; g.K = K;
  g.lda = lda; g.ldb = ldb; g.ldc = ldc;
  g.sA = strideA; g.sB = strideB; g.sC = strideC; g.sBias = strideBias;
  g.bytesA = (unsigned)exA; g.bytesB = (unsigned)exB;
  g.bias = bias; g.aux = aux; g.ldaux = ldaux; g.alpha = alpha;
  g.drop_thr = naws_drop_threshold(drop_ratio);
  g.drop_scale = (float)(1.0 / (1.0 - (double)drop_ratio));
  g.seed = seed; g.epilogue = epilogue; g.accumulate = accumulate;
  if (rowmax || colmax) {
    const int seg = rowmax_seg_cols > 0 ? rowmax_seg_cols : N;
    if (rowmax && seg < N && seg % 256 != 0) return NAWS_ERR_ARG;
    const int nseg = (int)naws_cdiv(N, seg);
    g.am.rowmax = rowmax; g.am.colmax = colmax; g.am.colmul = colmax_rowmul;
    g.am.seg_cols = seg >= N ? 0x40000000 : seg;
    g.am.sRow = (long long)nseg * M; g.am.sCol = N;
  }
  hipStream_t s = (hipStream_t)stream;
  // short inner dimension, plain NN form (dZ7 = gate(dL W8), K = 2C): the register-resident kernel
  if (!transA && !transB && K <= SK_MAXK && N % 4 == 0 && ldc % 4 == 0 && ((uintptr_t)C & 15) == 0 &&
      strideC % 4 == 0 && (epilogue == NAWS_EPI_NONE || epilogue == NAWS_EPI_GATE_POS) &&
      (!aux || (ldaux % 4 == 0 && ((uintptr_t)aux & 15) == 0)) && (long long)M * N >= (1 << 20) &&
      (!rowmax || g.am.seg_cols % 1024 == 0 || g.am.seg_cols >= N) && gemm_variant() != 7) {
    dim3 grid((unsigned)naws_cdiv(N, 1024), (unsigned)naws_cdiv(M, SKM_ROWS), batch);
#ifdef NAWS_AB
    if (gemm_variant() == 8) grid.y = (unsigned)naws_cdiv(M, SK_ROWS);
#endif
    if (grid.y <= 65535) {
#ifdef NAWS_AB
      if (gemm_variant() == 8) {
        if (K <= 40)
          hipLaunchKernelGGL(gemm_smallk_nn_kernel<40>, grid, dim3(256), 0, s, g);
        else
          hipLaunchKernelGGL(gemm_smallk_nn_kernel<64>, grid, dim3(256), 0, s, g);
      } else
#endif
      if (K <= 40) {
        hipLaunchKernelGGL(gemm_smallk_mfma_kernel<10>, grid, dim3(256), 0, s, g);
      } else {
        hipLaunchKernelGGL(gemm_smallk_mfma_kernel<16>, grid, dim3(256), 0, s, g);
      }
      return naws_check_launch();
    }
  }
  const bool a_kc = !transA, b_kc = transB != 0;
  if (a_kc && b_kc) return dispatch<true, true, false>(g, batch, s);
  if (a_kc && !b_kc) return dispatch<true, false, false>(g, batch, s);
  if (!a_kc && b_kc) return dispatch<false, true, false>(g, batch, s);
  return dispatch<false, false, false>(g, batch, s);
}

extern "C" int64_t naws_gemm_f32_splitk_workspace_floats(int M, int N, int batch, int ksplit) {
  if (M <= 0 || N <= 0 || batch <= 0 || ksplit <= 0) return 0;
  return (int64_t)M * N * batch * ksplit;
}

// naws_gemm_f32 for a SMALL output with a LONG inner dimension (fc8: logits = H7 W8^T, M x 2C
// from K = 4096; dW8 = dL^T H7, 2C x 4096 from K = proposals): with one 64 x 64 tile per
// workgroup only ~126 workgroups exist and each walks K alone, one memory round trip per step.
// K is cut into `ksplit` slices (more workgroups, more bytes in flight), the partial products go
// to `workspace`, and a second pass adds them in slice order - deterministic, unlike atomics -
// and applies the epilogue (NAWS_EPI_NONE or NAWS_EPI_BIAS).
extern "C" int naws_gemm_f32_splitk(int transA, int transB, int M, int N, int K, const float* A,
                                    int lda, const float* B, int ldb, float* C, int ldc, int batch,
                                    int64_t strideA, int64_t strideB, int64_t strideC, int epilogue,
                                    const float* bias, int64_t strideBias, int ksplit,
                                    float* workspace, void* stream) {
  if (M <= 0 || N <= 0 || K <= 0 || batch <= 0 || ksplit <= 0) return NAWS_ERR_SHAPE;
  NAWS_REQUIRE_PTR(A); NAWS_REQUIRE_PTR(B); NAWS_REQUIRE_PTR(C); NAWS_REQUIRE_PTR(workspace);
  if (epilogue != NAWS_EPI_NONE && epilogue != NAWS_EPI_BIAS) return NAWS_ERR_UNSUPPORTED;
  if (epilogue == NAWS_EPI_BIAS) NAWS_REQUIRE_PTR(bias);
  if (!aligned4(lda) || !aligned4(ldb) || !aligned4(strideA) || !aligned4(strideB) || !ptr16(A) ||
      !ptr16(B))
    return NAWS_ERR_ARG;
  if (lda < (transA ? M : K) || ldb < (transB ? K : N) || ldc < N) return NAWS_ERR_ARG;
  const long long exA = transA ? extent_bytes(K, M, lda) : extent_bytes(M, K, lda);
  const long long exB = transB ? extent_bytes(N, K, ldb) : extent_bytes(K, N, ldb);
  if (exA > MAX_EXTENT || exB > MAX_EXTENT || (long long)batch * ksplit > 65535) return NAWS_ERR_UNSUPPORTED;
  // every slice at least one 32-deep K-step
  const int Tall = (int)naws_cdiv(K, 32);
  int ks = std::min(ksplit, Tall);
  while (ks > 1 && (long long)(ks - 1) * naws_cdiv(Tall, ks) >= Tall) --ks;
  GemmArgs g{};
  g.A = A; g.B = B; g.C = workspace; g.M = M; g.N = N; g.K = K;
  g.lda = lda; g.ldb = ldb; g.ldc = N;
  g.sA = strideA; g.sB = strideB; g.sC = (long long)M * N;
  g.bytesA = (unsigned)exA; g.bytesB = (unsigned)exB;
  g.epilogue = NAWS_EPI_NONE; g.ksplit = ks;
  hipStream_t s = (hipStream_t)stream;
  int rc;
  const bool a_kc = !transA, b_kc = transB != 0;
  // (ksplit == 1 also goes through the workspace: one code path)
  if (ks == 1) g.ksplit = 0;
  if (a_kc && b_kc) rc = launch<64, 64, 32, true, true, false>(g, batch, s);
  else if (a_kc && !b_kc) rc = launch<64, 64, 32, true, false, false>(g, batch, s);
  else if (!a_kc && b_kc) rc = launch<64, 64, 32, false, true, false>(g, batch, s);
  else rc = launch<64, 64, 32, false, false, false>(g, batch, s);
  if (rc != NAWS_OK) return rc;
  const long long total = (long long)M * N * batch;
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)std::min<long long>(naws_cdiv(total, 256), 2048)),
                     dim3(256), 0, s, (const float*)workspace, ks, batch, M, N, C, ldc,
                     (long long)strideC, epilogue == NAWS_EPI_BIAS ? bias : nullptr,
                     (long long)strideBias);
  return naws_check_launch();
}

extern "C" int naws_gemm_f32(int transA, int transB, int M, int N, int K, const float* A, int lda,
                             const float* B, int ldb, float* C, int ldc, int batch,
                             int64_t strideA, int64_t strideB, int64_t strideC, int epilogue,
                             const float* bias, int64_t strideBias, const float* aux, int ldaux,
                             float alpha, float drop_ratio, uint64_t seed, int accumulate,
                             void* stream) {
  return naws_gemm_f32_amax(transA, transB, M, N, K, A, lda, B, ldb, C, ldc, batch, strideA,
                            strideB, strideC, epilogue, bias, strideBias, aux, ldaux, alpha,
                            drop_ratio, seed, accumulate, nullptr, 0, nullptr, nullptr, stream);
}

extern "C" int naws_conv3x3_nhwc_fwd(const float* X, const float* Wp, const float* bias, int N,
                                     int H, int W, int Cin, int Cout, int dilation, int relu,
                                     float* Y, void* stream) {
  if (N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return NAWS_ERR_SHAPE;
  if (dilation < 1) return NAWS_ERR_ARG;
  if (Cin % 32 != 0 || Cout % 4 != 0) return NAWS_ERR_UNSUPPORTED;
  NAWS_REQUIRE_PTR(X); NAWS_REQUIRE_PTR(Wp); NAWS_REQUIRE_PTR(Y);
  if (!ptr16(X) || !ptr16(Wp)) return NAWS_ERR_ARG;
  const long long pix = (long long)N * H * W;
  if (pix > 0x7fffffffLL || pix * Cin * 4 > MAX_EXTENT) return NAWS_ERR_UNSUPPORTED;
  GemmArgs g{};
  g.A = X; g.B = Wp; g.C = Y;
  g.M = (int)pix; g.N = Cout; g.K = 9 * Cin;
  g.lda = Cin; g.ldb = 9 * Cin; g.ldc = Cout;
  g.bytesA = (unsigned)(pix * Cin * 4);
  g.bytesB = (unsigned)((long long)Cout * 9 * Cin * 4);
  g.bias = bias; g.epilogue = bias ? (relu ? NAWS_EPI_BIAS_RELU : NAWS_EPI_BIAS) : NAWS_EPI_NONE;
  if (!bias && relu) return NAWS_ERR_ARG;
  g.H = H; g.W = W; g.Cin = Cin; g.dil = dilation;
  return dispatch<true, true, true>(g, 1, (hipStream_t)stream);
}

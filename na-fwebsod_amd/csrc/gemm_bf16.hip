// bf16 MFMA GEMM (v_mfma_f32_32x32x16_bf16, fp32 accumulate) for the "bf16 MFMA conv/fc with
// fp32 loss" configuration (BASELINE.json configs[3]; SURVEY.md §8 a-1 / a-4 "bf16 option").
//
//   C[M,N] (+)= A[M,K] * B[N,K]^T     both operands K-contiguous ("NT"), fp32 output,
//   same fused epilogues as the fp32 kernel (bias / ReLU / Dropout / ReLU-gate),
//   plus the 3x3 implicit-GEMM convolution form (A gathered from an NHWC fp32 image).
//
// Each operand is read from HBM either as fp32 (converted to bf16 on its way into LDS, so
// parameters and activations stay fp32 in memory) or as bf16 (the transposed copies that
// naws_transpose_to_bf16 makes for the backward GEMMs, whose natural operands are not
// K-contiguous).  At 16x the fp32 MFMA rate these GEMMs are bandwidth-bound, so the kernel is
// built like the fp32 one for latency hiding (buffer loads two tiles ahead, 3 workgroups per
// CU) but with 16-byte LDS reads feeding one MFMA each: LDS rows are BK+8 bf16 (80 B), which
// puts the 16 lanes of a ds_read_b128 group on 16 distinct 16-B slots.
#include <stdlib.h>
#include <type_traits>
#include "naws_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

struct BArgs {
  const void* A;
  const void* B;
  float* C;
  int M, N, K;
  int lda, ldb, ldc;            // in elements of the operand's own type
  long long sA, sB, sC, sBias;  // batch strides, elements
  unsigned bytesA, bytesB;
  const float* bias;
  const float* aux;
  int ldaux;
  float alpha;
  unsigned drop_thr;
  float drop_scale;
  unsigned long long seed;
  int epilogue, accumulate;
  int tiles_m, tiles_n;
  int H, W, Cin, dil;           // conv form
};

constexpr int BK = 32;          // bf16 elements per K-step
constexpr int LDS_LD = BK + 8;  // bf16 elements per LDS row (80 B)
constexpr unsigned OOB = 0xFFFFFFF0u;

__device__ __forceinline__ u32x4 buf_load16(__amdgpu_buffer_rsrc_t r, unsigned off) {
  return __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0);
}

__device__ __forceinline__ unsigned pack2(float a, float b) {
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  bf16x2 v;
  v[0] = (__bf16)a;
  v[1] = (__bf16)b;
  return *reinterpret_cast<unsigned*>(&v);
}

// One operand tile: ROWS x 32 k.  A "unit" = 8 consecutive k of one row = 16 B of bf16 in LDS.
template <int ROWS, bool SRC_BF16, int NT>
struct BStage {
  static constexpr int UNITS = ROWS * 4 / NT;
  static_assert(ROWS * 4 % NT == 0, "tile must divide over the workgroup");
  static constexpr int ESZ = SRC_BF16 ? 2 : 4;
  unsigned base[UNITS];   // byte offset of (row, kq) at k0 = 0, or OOB

  __device__ __forceinline__ void init(int ld, int row0, int nrows) {
#pragma unroll
    for (int i = 0; i < UNITS; ++i) {
      const int u = threadIdx.x + i * NT;
      const int row = u >> 2, kq = u & 3;
      const int gr = row0 + row;
      base[i] = gr < nrows ? ((unsigned)gr * (unsigned)ld + kq * 8) * ESZ : OOB;
    }
  }
  // regs: 2 x u32x4 per unit for fp32 sources, 1 for bf16 sources
  __device__ __forceinline__ void load(__amdgpu_buffer_rsrc_t rs, int t, int K, u32x4* r) const {
#pragma unroll
    for (int i = 0; i < UNITS; ++i) {
      const int u = threadIdx.x + i * NT;
      const int kq = u & 3;
      const bool ok = (base[i] != OOB) && (t * BK + kq * 8 < K);
      const unsigned off = ok ? base[i] + (unsigned)t * (BK * ESZ) : OOB;
      if (SRC_BF16) {
        r[i] = buf_load16(rs, off);
      } else {
        r[2 * i] = buf_load16(rs, off);
        r[2 * i + 1] = buf_load16(rs, ok ? off + 16 : OOB);
      }
    }
  }
  __device__ __forceinline__ static void store(unsigned short* S, const u32x4* r) {
#pragma unroll
    for (int i = 0; i < UNITS; ++i) {
      const int u = threadIdx.x + i * NT;
      const int row = u >> 2, kq = u & 3;
      u32x4 v;
      if (SRC_BF16) {
        v = r[i];
      } else {
        const u32x4 a = r[2 * i], b = r[2 * i + 1];
        v.x = pack2(__uint_as_float(a.x), __uint_as_float(a.y));
        v.y = pack2(__uint_as_float(a.z), __uint_as_float(a.w));
        v.z = pack2(__uint_as_float(b.x), __uint_as_float(b.y));
        v.w = pack2(__uint_as_float(b.z), __uint_as_float(b.w));
      }
      *reinterpret_cast<u32x4*>(S + row * LDS_LD + kq * 8) = v;
    }
  }
};

// conv A tile (fp32 NHWC source): rows = output pixels, a K-step lies inside one 3x3 tap
template <int ROWS, int NT>
struct BConvStage {
  static constexpr int UNITS = ROWS * 4 / NT;
  int y[UNITS], x[UNITS];
  unsigned base[UNITS];
  __device__ __forceinline__ void init(int row0, int M, int H, int W, int Cin) {
#pragma unroll
    for (int i = 0; i < UNITS; ++i) {
      const int u = threadIdx.x + i * NT;
      const int row = u >> 2, kq = u & 3;
      const int gm = row0 + row;
      if (gm < M) {
        x[i] = gm % W; y[i] = (gm / W) % H;
        base[i] = ((unsigned)gm * (unsigned)Cin + kq * 8) * 4u;
      } else { x[i] = 0; y[i] = 0; base[i] = OOB; }
    }
  }
  __device__ __forceinline__ void load(__amdgpu_buffer_rsrc_t rs, int t, int H, int W, int Cin,
                                       int dil, u32x4* r) const {
    const int k0 = t * BK;
    const int tap = k0 / Cin, c0 = k0 - tap * Cin;
    const int dy = (tap / 3 - 1) * dil, dx = (tap % 3 - 1) * dil;
    const int delta = ((dy * W + dx) * Cin + c0) * 4;
#pragma unroll
    for (int i = 0; i < UNITS; ++i) {
      const int yy = y[i] + dy, xx = x[i] + dx;
      const bool ok = (base[i] != OOB) && yy >= 0 && yy < H && xx >= 0 && xx < W;
      const unsigned off = ok ? (unsigned)((int)base[i] + delta) : OOB;
      r[2 * i] = buf_load16(rs, off);
      r[2 * i + 1] = buf_load16(rs, ok ? off + 16 : OOB);
    }
  }
};

template <int BM, int BN, int WM, int WN, bool A_BF16, bool B_BF16, bool CONV>
__global__ __launch_bounds__(64 * WM * WN) void gemm_bf16_kernel(BArgs g) {
  constexpr int NT = 64 * WM * WN;
  constexpr int WTM = BM / WM, WTN = BN / WN;
  constexpr int TI = WTM / 32, TJ = WTN / 32;
  using SA = BStage<BM, A_BF16, NT>;
  using SB = BStage<BN, B_BF16, NT>;
  using SC = BConvStage<BM, NT>;
  constexpr int A_ELEMS = BM * LDS_LD, B_ELEMS = BN * LDS_LD;
  constexpr int RA = SA::UNITS * (A_BF16 ? 1 : 2), RB = SB::UNITS * (B_BF16 ? 1 : 2);

  extern __shared__ __attribute__((aligned(16))) unsigned short smb[];
  constexpr int A_OFF = 0, B_OFF = 2 * A_ELEMS;

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
  const char* Ap = reinterpret_cast<const char*>(g.A) + bz * g.sA * (A_BF16 ? 2 : 4);
  const char* Bp = reinterpret_cast<const char*>(g.B) + bz * g.sB * (B_BF16 ? 2 : 4);
  const __amdgpu_buffer_rsrc_t rsA =
      __builtin_amdgcn_make_buffer_rsrc((void*)Ap, 0, (int)g.bytesA, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB =
      __builtin_amdgcn_make_buffer_rsrc((void*)Bp, 0, (int)g.bytesB, 0x00020000);
  float* C = g.C + bz * g.sC;

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

  u32x4 ra0[RA], rb0[RB], ra1[RA], rb1[RB];
  SA stA; SC cvA; SB stB;
  if constexpr (CONV) cvA.init(m0, g.M, g.H, g.W, g.Cin);
  else stA.init(g.lda, m0, g.M);
  stB.init(g.ldb, n0, g.N);

  const int T = (g.K + BK - 1) / BK;
  auto fetch = [&](int t, u32x4* ra, u32x4* rb) {
    if constexpr (CONV) cvA.load(rsA, t, g.H, g.W, g.Cin, g.dil, ra);
    else stA.load(rsA, t, g.K, ra);
    stB.load(rsB, t, g.K, rb);
  };
  auto stash = [&](int buf, const u32x4* ra, const u32x4* rb) {
    if constexpr (CONV) BStage<BM, false, NT>::store(smb + A_OFF + buf * A_ELEMS, ra);
    else SA::store(smb + A_OFF + buf * A_ELEMS, ra);
    SB::store(smb + B_OFF + buf * B_ELEMS, rb);
  };

  auto kstep = [&](int t, auto pipe_tag, u32x4* lra, u32x4* lrb, const u32x4* sra,
                   const u32x4* srb) {
    constexpr bool PIPE = decltype(pipe_tag)::value;
    const int cur = t & 1;
    const bool load2 = PIPE || (t + 2 < T);
    const bool store1 = PIPE || (t + 1 < T);
    const unsigned short* as = smb + A_OFF + cur * A_ELEMS;
    const unsigned short* bs = smb + B_OFF + cur * B_ELEMS;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 af[TI], bf[TJ];
#pragma unroll
      for (int i = 0; i < TI; ++i)
        af[i] = *reinterpret_cast<const bf16x8*>(as + (wm * WTM + i * 32 + l31) * LDS_LD + ks * 16 + h * 8);
#pragma unroll
      for (int j = 0; j < TJ; ++j)
        bf[j] = *reinterpret_cast<const bf16x8*>(bs + (wn * WTN + j * 32 + l31) * LDS_LD + ks * 16 + h * 8);
      if (ks == 0) {
        if (load2) fetch(t + 2, lra, lrb);
        if (PIPE) __builtin_amdgcn_sched_barrier(0);
      }
      if (ks == 1) {
        if (PIPE) __builtin_amdgcn_sched_barrier(0);
        if (store1) stash(cur ^ 1, sra, srb);
        if (PIPE) __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
  };

  fetch(0, ra0, rb0);
  stash(0, ra0, rb0);
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
      }
    }
  }
}

template <int BM, int BN, int WM, int WN, bool A_BF16, bool B_BF16, bool CONV>
int launch(BArgs& g, int batch, hipStream_t s) {
  g.tiles_m = (int)naws_cdiv(g.M, BM);
  g.tiles_n = (int)naws_cdiv(g.N, BN);
  const size_t lds = (size_t)2 * (BM + BN) * LDS_LD * sizeof(unsigned short);
  auto kern = gemm_bf16_kernel<BM, BN, WM, WN, A_BF16, B_BF16, CONV>;
  if (naws_allow_lds(kern) != NAWS_OK) return NAWS_ERR_LAUNCH;
  dim3 grid((unsigned)(g.tiles_m * g.tiles_n), 1, batch);
  hipLaunchKernelGGL(kern, grid, dim3(64 * WM * WN), lds, s, g);
  return naws_check_launch();
}

template <bool A_BF16, bool B_BF16, bool CONV>
int dispatch(BArgs& g, int batch, hipStream_t s) {
  if (g.N <= 64) return launch<128, 64, 2, 2, A_BF16, B_BF16, CONV>(g, batch, s);
  if (naws_cdiv(g.M, 256) * naws_cdiv(g.N, 128) * batch >= 512)
    return launch<256, 128, 4, 2, A_BF16, B_BF16, CONV>(g, batch, s);
  return launch<128, 128, 2, 2, A_BF16, B_BF16, CONV>(g, batch, s);
}

// fp32 [rows, cols] (ld) -> bf16 [cols, rows_pad] (zero-filled pad): the K-contiguous copies
// the backward GEMMs need.  32x32 LDS tiles.
__global__ void transpose_to_bf16_kernel(const float* __restrict__ X, int rows, int cols, int ld,
                                         int rows_pad, unsigned short* __restrict__ Y) {
  __shared__ float tile[32][33];
  const long long bz = blockIdx.z;
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  for (int i = threadIdx.y; i < 32; i += blockDim.y) {
    const int r = r0 + i, c = c0 + threadIdx.x;
    tile[i][threadIdx.x] = (r < rows && c < cols) ? X[bz * (long long)rows * ld + (long long)r * ld + c] : 0.f;
  }
  __syncthreads();
  for (int i = threadIdx.y; i < 32; i += blockDim.y) {
    const int c = c0 + i, r = r0 + threadIdx.x;
    if (c < cols && r < rows_pad) {
      const __bf16 v = (__bf16)tile[threadIdx.x][i];
      Y[bz * (long long)cols * rows_pad + (long long)c * rows_pad + r] =
          *reinterpret_cast<const unsigned short*>(&v);
    }
  }
}

bool al(long long v, int a) { return (v % a) == 0; }

}  // namespace

extern "C" int naws_gemm_bf16_nt(int M, int N, int K, const void* A, int a_is_bf16, int lda,
                                 const void* B, int b_is_bf16, int ldb, float* C, int ldc, int batch,
                                 int64_t strideA, int64_t strideB, int64_t strideC, int epilogue,
                                 const float* bias, int64_t strideBias, const float* aux, int ldaux,
                                 float alpha, float drop_ratio, uint64_t seed, int accumulate,
                                 void* stream) {
  if (M <= 0 || N <= 0 || K <= 0 || batch <= 0) return NAWS_ERR_SHAPE;
  NAWS_REQUIRE_PTR(A); NAWS_REQUIRE_PTR(B); NAWS_REQUIRE_PTR(C);
  if (epilogue < NAWS_EPI_NONE || epilogue > NAWS_EPI_GATE_POS) return NAWS_ERR_ARG;
  if (epilogue == NAWS_EPI_GATE_POS && aux == nullptr) return NAWS_ERR_NULL;
  if (epilogue == NAWS_EPI_BIAS_RELU_DROP && !(drop_ratio >= 0.f && drop_ratio < 1.f)) return NAWS_ERR_ARG;
  if (lda < K || ldb < K || ldc < N) return NAWS_ERR_SHAPE;
  // 8 consecutive k per 16/32-byte unit
  if (!al(K, 8) || !al(lda, 8) || !al(ldb, 8) || !al(strideA, 8) || !al(strideB, 8)) return NAWS_ERR_ARG;
  if ((((uintptr_t)A | (uintptr_t)B) & 15) != 0) return NAWS_ERR_ARG;
  if (batch > 65535) return NAWS_ERR_UNSUPPORTED;
  const long long exA = ((long long)(M - 1) * lda + K) * (a_is_bf16 ? 2 : 4);
  const long long exB = ((long long)(N - 1) * ldb + K) * (b_is_bf16 ? 2 : 4);
  if (exA > 0xFFFFFF00LL || exB > 0xFFFFFF00LL) return NAWS_ERR_UNSUPPORTED;
  BArgs g{};
  g.A = A; g.B = B; g.C = C; g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
  g.sA = strideA; g.sB = strideB; g.sC = strideC; g.sBias = strideBias;
  g.bytesA = (unsigned)exA; g.bytesB = (unsigned)exB;
  g.bias = bias; g.aux = aux; g.ldaux = ldaux; g.alpha = alpha;
  g.drop_thr = naws_drop_threshold(drop_ratio);
  g.drop_scale = (float)(1.0 / (1.0 - (double)drop_ratio));
  g.seed = seed; g.epilogue = epilogue; g.accumulate = accumulate;
  hipStream_t s = (hipStream_t)stream;
  if (a_is_bf16 && b_is_bf16) return dispatch<true, true, false>(g, batch, s);
  if (a_is_bf16) return dispatch<true, false, false>(g, batch, s);
  if (b_is_bf16) return dispatch<false, true, false>(g, batch, s);
  return dispatch<false, false, false>(g, batch, s);
}

extern "C" int naws_conv3x3_nhwc_bf16_fwd(const float* X, const float* Wp, const float* bias, int N,
                                          int H, int W, int Cin, int Cout, int dilation, int relu,
                                          float* Y, void* stream) {
  if (N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return NAWS_ERR_SHAPE;
  if (dilation < 1) return NAWS_ERR_ARG;
  if (Cin % 32 != 0 || Cout % 4 != 0) return NAWS_ERR_UNSUPPORTED;
  NAWS_REQUIRE_PTR(X); NAWS_REQUIRE_PTR(Wp); NAWS_REQUIRE_PTR(Y);
  if (!bias && relu) return NAWS_ERR_ARG;
  const long long pix = (long long)N * H * W;
  if (pix > 0x7fffffffLL || pix * Cin * 4 > 0xFFFFFF00LL) return NAWS_ERR_UNSUPPORTED;
  BArgs g{};
  g.A = X; g.B = Wp; g.C = Y; g.M = (int)pix; g.N = Cout; g.K = 9 * Cin;
  g.lda = Cin; g.ldb = 9 * Cin; g.ldc = Cout;
  g.bytesA = (unsigned)(pix * Cin * 4);
  g.bytesB = (unsigned)((long long)Cout * 9 * Cin * 4);
  g.bias = bias; g.epilogue = bias ? (relu ? NAWS_EPI_BIAS_RELU : NAWS_EPI_BIAS) : NAWS_EPI_NONE;
  g.H = H; g.W = W; g.Cin = Cin; g.dil = dilation;
  return dispatch<false, false, true>(g, 1, (hipStream_t)stream);
}

extern "C" int naws_transpose_to_bf16(const float* X, int batch, int rows, int cols, int ld,
                                      int rows_pad, void* Y, void* stream) {
  if (batch <= 0 || rows <= 0 || cols <= 0 || ld < cols || rows_pad < rows) return NAWS_ERR_SHAPE;
  NAWS_REQUIRE_PTR(X); NAWS_REQUIRE_PTR(Y);
  if (batch > 65535 || naws_cdiv(rows_pad, 32) > 65535) return NAWS_ERR_UNSUPPORTED;
  dim3 grid((unsigned)naws_cdiv(cols, 32), (unsigned)naws_cdiv(rows_pad, 32), batch);
  hipLaunchKernelGGL(transpose_to_bf16_kernel, grid, dim3(32, 8), 0, (hipStream_t)stream, X, rows,
                     cols, ld, rows_pad, (unsigned short*)Y);
  return naws_check_launch();
}

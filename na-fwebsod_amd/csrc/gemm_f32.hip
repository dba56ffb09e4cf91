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
// each wave owns TI x TJ tiles of 32x32 accumulators.  Operand tiles are staged
// global -> registers -> LDS (two LDS buffers, one barrier per K-step; the next
// tile's global loads are in flight during the current tile's MFMAs).
// K-contiguous operands sit in LDS as [row][BK+4] (pad 4 floats => the 16-lane
// ds_read_b128 groups hit 16 distinct 16-B slots) and each lane fetches FOUR
// k-values with one ds_read_b128; M/N-contiguous operands sit as [k][row] and
// are fetched with conflict-free ds_read_b32.  Both forms use the same k
// assignment inside an 8-deep group (MFMA step s, lane half h -> k = 4h + s),
// so any A/B layout pair multiplies matching k's.
#include "naws_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct GemmArgs {
  const float* A;
  const float* B;
  float* C;
  int M, N, K;
  int lda, ldb, ldc;
  long long sA, sB, sC, sBias;
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
};

constexpr int NT = 256;  // threads per workgroup
constexpr int PADK = 4;  // K-contiguous LDS row pad (floats)

template <int ROWS, int BK, bool KC>
struct TileGeom {
  static constexpr int LD = KC ? (BK + PADK) : ROWS;          // LDS leading dim
  static constexpr int FLOATS = KC ? ROWS * (BK + PADK) : BK * ROWS;
  static constexpr int VEC_PER_THREAD = ROWS * BK / 4 / NT;   // float4 per thread
  static_assert(ROWS * BK / 4 % NT == 0, "tile must divide over the workgroup");
};

// ---- global -> registers ---------------------------------------------------
template <int ROWS, int BK, bool KC>
__device__ __forceinline__ void load_tile(const float* __restrict__ P, int ld, int row0,
                                          int nrows, int k0, int K, float4* __restrict__ r) {
  using G = TileGeom<ROWS, BK, KC>;
#pragma unroll
  for (int i = 0; i < G::VEC_PER_THREAD; ++i) {
    const int f = threadIdx.x + i * NT;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (KC) {
      const int row = f / (BK / 4), c4 = f % (BK / 4);
      const int gr = row0 + row, gk = k0 + c4 * 4;
      if (gr < nrows && gk < K) v = *reinterpret_cast<const float4*>(P + (long long)gr * ld + gk);
    } else {
      const int kr = f / (ROWS / 4), m4 = f % (ROWS / 4);
      const int gk = k0 + kr, gr = row0 + m4 * 4;
      if (gk < K && gr < nrows) v = *reinterpret_cast<const float4*>(P + (long long)gk * ld + gr);
    }
    r[i] = v;
  }
}

template <int ROWS, int BK, bool KC>
__device__ __forceinline__ void store_tile(float* __restrict__ S, const float4* __restrict__ r) {
  using G = TileGeom<ROWS, BK, KC>;
#pragma unroll
  for (int i = 0; i < G::VEC_PER_THREAD; ++i) {
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

// conv A tile: rows are output pixels, the K-step lies inside one 3x3 tap.
template <int ROWS, int BK>
struct ConvRows {
  static constexpr int VPT = TileGeom<ROWS, BK, true>::VEC_PER_THREAD;
  int y[VPT], x[VPT];
  long long base[VPT];  // pixel offset (n*H + y)*W + x, or -1 when the row is out of range
};

template <int ROWS, int BK>
__device__ __forceinline__ void conv_rows_init(ConvRows<ROWS, BK>& cr, int row0, int M, int H,
                                               int W) {
#pragma unroll
  for (int i = 0; i < ConvRows<ROWS, BK>::VPT; ++i) {
    const int f = threadIdx.x + i * NT;
    const int row = f / (BK / 4);
    const int gm = row0 + row;
    if (gm < M) {
      const int xx = gm % W;
      const int t = gm / W;
      cr.x[i] = xx;
      cr.y[i] = t % H;
      cr.base[i] = gm;
    } else {
      cr.x[i] = 0; cr.y[i] = 0; cr.base[i] = -1;
    }
  }
}

template <int ROWS, int BK>
__device__ __forceinline__ void load_tile_conv(const float* __restrict__ X,
                                               const ConvRows<ROWS, BK>& cr, int k0, int H, int W,
                                               int Cin, int dil, float4* __restrict__ r) {
  const int tap = k0 / Cin, c0 = k0 - tap * Cin;
  const int dy = (tap / 3 - 1) * dil, dx = (tap % 3 - 1) * dil;
#pragma unroll
  for (int i = 0; i < ConvRows<ROWS, BK>::VPT; ++i) {
    const int f = threadIdx.x + i * NT;
    const int c4 = f % (BK / 4);
    const int yy = cr.y[i] + dy, xx = cr.x[i] + dx;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (cr.base[i] >= 0 && yy >= 0 && yy < H && xx >= 0 && xx < W)
      v = *reinterpret_cast<const float4*>(X + (cr.base[i] + (long long)dy * W + dx) * Cin + c0 +
                                           c4 * 4);
    r[i] = v;
  }
}

// ---- the kernel --------------------------------------------------------------
template <int BM, int BN, int BK, bool A_KC, bool B_KC, bool CONV>
__global__ __launch_bounds__(NT) void gemm_f32_kernel(GemmArgs g) {
  using GA = TileGeom<BM, BK, A_KC>;
  using GB = TileGeom<BN, BK, B_KC>;
  constexpr int WTM = BM / 2, WTN = BN / 2;  // per-wave output
  constexpr int TI = WTM / 32, TJ = WTN / 32;
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

  const long long bz = blockIdx.z;
  const float* A = g.A + bz * g.sA;
  const float* B = g.B + bz * g.sB;
  float* C = g.C + bz * g.sC;

  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int wm = wid >> 1, wn = wid & 1;
  const int l31 = lane & 31, h = lane >> 5;

  f32x16 acc[TI][TJ];
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int j = 0; j < TJ; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  float4 ra[GA::VEC_PER_THREAD], rb[GB::VEC_PER_THREAD];
  ConvRows<BM, BK> cr;
  if (CONV) conv_rows_init<BM, BK>(cr, m0, g.M, g.H, g.W);

  const int T = (g.K + BK - 1) / BK;
  auto fetch = [&](int t) {
    if (CONV) load_tile_conv<BM, BK>(A, cr, t * BK, g.H, g.W, g.Cin, g.dil, ra);
    else load_tile<BM, BK, A_KC>(A, g.lda, m0, g.M, t * BK, g.K, ra);
    load_tile<BN, BK, B_KC>(B, g.ldb, n0, g.N, t * BK, g.K, rb);
  };

  fetch(0);
  store_tile<BM, BK, A_KC>(sm + A_OFF, ra);
  store_tile<BN, BK, B_KC>(sm + B_OFF, rb);
  __syncthreads();

  for (int t = 0; t < T; ++t) {
    const int cur = t & 1;
    if (t + 1 < T) fetch(t + 1);
    const float* as = sm + A_OFF + cur * GA::FLOATS;
    const float* bs = sm + B_OFF + cur * GB::FLOATS;
#pragma unroll
    for (int kg = 0; kg < BK / 8; ++kg) {
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
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
          for (int j = 0; j < TJ; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][s], bf[j][s], acc[i][j], 0, 0, 0);
    }
    if (t + 1 < T) {
      store_tile<BM, BK, A_KC>(sm + A_OFF + (cur ^ 1) * GA::FLOATS, ra);
      store_tile<BN, BK, B_KC>(sm + B_OFF + (cur ^ 1) * GB::FLOATS, rb);
    }
    __syncthreads();
  }

  // ---- epilogue: C/D map col = lane&31, row = (e&3) + 8*(e>>2) + 4*(lane>>5)
  const float* bias = g.bias ? g.bias + bz * g.sBias : nullptr;
  const float* aux = g.aux ? g.aux + bz * g.sC : nullptr;
#pragma unroll
  for (int j = 0; j < TJ; ++j) {
    const int col = n0 + wn * WTN + j * 32 + l31;
    if (col >= g.N) continue;
    const float bv = (bias && g.epilogue >= NAWS_EPI_BIAS && g.epilogue <= NAWS_EPI_BIAS_RELU_DROP)
                         ? bias[col] : 0.f;
#pragma unroll
    for (int i = 0; i < TI; ++i) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = m0 + wm * WTM + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
        if (row >= g.M) continue;
        float v = acc[i][j][e];
        switch (g.epilogue) {
          case NAWS_EPI_BIAS: v += bv; break;
          case NAWS_EPI_BIAS_RELU: v = fmaxf(v + bv, 0.f); break;
          case NAWS_EPI_BIAS_RELU_DROP: {
            v = fmaxf(v + bv, 0.f);
            const unsigned long long idx =
                (unsigned long long)bz * g.M * g.N + (unsigned long long)row * g.N + col;
            v = naws_keep(g.seed, idx, g.drop_thr) ? v * g.drop_scale : 0.f;
          } break;
          case NAWS_EPI_GATE_POS:
            v = (aux[(long long)row * g.ldaux + col] > 0.f) ? v * g.alpha : 0.f;
            break;
          default: break;
        }
        float* dst = C + (long long)row * g.ldc + col;
        if (g.accumulate) v += *dst;
        *dst = v;
      }
    }
  }
}

template <int BM, int BN, int BK, bool A_KC, bool B_KC, bool CONV>
int launch(GemmArgs& g, int batch, hipStream_t s) {
  using GA = TileGeom<BM, BK, A_KC>;
  using GB = TileGeom<BN, BK, B_KC>;
  g.tiles_m = (int)naws_cdiv(g.M, BM);
  g.tiles_n = (int)naws_cdiv(g.N, BN);
  const size_t lds = (size_t)2 * (GA::FLOATS + GB::FLOATS) * sizeof(float);
  auto kern = gemm_f32_kernel<BM, BN, BK, A_KC, B_KC, CONV>;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  dim3 grid((unsigned)(g.tiles_m * g.tiles_n), 1, batch);
  hipLaunchKernelGGL(kern, grid, dim3(NT), lds, s, g);
  return naws_check_launch();
}

// Tile choice: 128x128 by default; narrower N tile for N <= 64; when the grid
// would leave CUs idle (fewer than ~2 tiles per CU) fall back to 64-row tiles.
template <bool A_KC, bool B_KC, bool CONV>
int dispatch(GemmArgs& g, int batch, hipStream_t s) {
  const long long t128 = naws_cdiv(g.M, 128) * naws_cdiv(g.N, 128) * batch;
  if (g.N <= 64) {
    if (naws_cdiv(g.M, 128) * batch >= 512) return launch<128, 64, 32, A_KC, B_KC, CONV>(g, batch, s);
    return launch<64, 64, 32, A_KC, B_KC, CONV>(g, batch, s);
  }
  if (t128 >= 1024) return launch<128, 128, 32, A_KC, B_KC, CONV>(g, batch, s);
  if (naws_cdiv(g.M, 64) * naws_cdiv(g.N, 128) * batch >= 768)
    return launch<64, 128, 32, A_KC, B_KC, CONV>(g, batch, s);
  return launch<64, 64, 32, A_KC, B_KC, CONV>(g, batch, s);
}

bool aligned4(long long v) { return (v & 3) == 0; }
bool ptr16(const void* p) { return ((uintptr_t)p & 15) == 0; }

}  // namespace

extern "C" int naws_gemm_f32(int transA, int transB, int M, int N, int K, const float* A, int lda,
                             const float* B, int ldb, float* C, int ldc, int batch,
                             int64_t strideA, int64_t strideB, int64_t strideC, int epilogue,
                             const float* bias, int64_t strideBias, const float* aux, int ldaux,
                             float alpha, float drop_ratio, uint64_t seed, int accumulate,
                             void* stream) {
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

  GemmArgs g{};
  g.A = A; g.B = B; g.C = C; g.M = M; g.N = N; g.K = K;
  g.lda = lda; g.ldb = ldb; g.ldc = ldc;
  g.sA = strideA; g.sB = strideB; g.sC = strideC; g.sBias = strideBias;
  g.bias = bias; g.aux = aux; g.ldaux = ldaux; g.alpha = alpha;
  g.drop_thr = naws_drop_threshold(drop_ratio);
  g.drop_scale = (float)(1.0 / (1.0 - (double)drop_ratio));
  g.seed = seed; g.epilogue = epilogue; g.accumulate = accumulate;
  hipStream_t s = (hipStream_t)stream;
  const bool a_kc = !transA, b_kc = transB != 0;
  if (a_kc && b_kc) return dispatch<true, true, false>(g, batch, s);
  if (a_kc && !b_kc) return dispatch<true, false, false>(g, batch, s);
  if (!a_kc && b_kc) return dispatch<false, true, false>(g, batch, s);
  return dispatch<false, false, false>(g, batch, s);
}

extern "C" int naws_conv3x3_nhwc_fwd(const float* X, const float* Wp, const float* bias, int N,
                                     int H, int W, int Cin, int Cout, int dilation, int relu,
                                     float* Y, void* stream) {
  if (N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return NAWS_ERR_SHAPE;
  if (dilation < 1) return NAWS_ERR_ARG;
  if (Cin % 32 != 0 || Cout % 4 != 0) return NAWS_ERR_UNSUPPORTED;
  NAWS_REQUIRE_PTR(X); NAWS_REQUIRE_PTR(Wp); NAWS_REQUIRE_PTR(Y);
  if (!ptr16(X) || !ptr16(Wp)) return NAWS_ERR_ARG;
  if ((long long)N * H * W > 0x7fffffffLL) return NAWS_ERR_UNSUPPORTED;
  GemmArgs g{};
  g.A = X; g.B = Wp; g.C = Y;
  g.M = N * H * W; g.N = Cout; g.K = 9 * Cin;
  g.lda = Cin; g.ldb = 9 * Cin; g.ldc = Cout;
  g.bias = bias; g.epilogue = bias ? (relu ? NAWS_EPI_BIAS_RELU : NAWS_EPI_BIAS) : NAWS_EPI_NONE;
  if (!bias && relu) return NAWS_ERR_ARG;
  g.H = H; g.W = W; g.Cin = Cin; g.dil = dilation;
  return dispatch<true, true, true>(g, 1, (hipStream_t)stream);
}

// Memory-bound helpers of the conv body (conv1_1 direct, 2x2 max-pool, layout
// transposes, weight repack) and the small Caffe2 built-ins the graph executor
// runs op-by-op (unary / binary-broadcast / row softmax / transpose / column
// sums / dropout mask).
//
// ref: detectron/modeling/VGG16.py:9-48 (Conv/Relu/MaxPool emission),
//      SURVEY.md §2 #13 (Caffe2 built-ins on the path, pytorch v1.3.0).
#include <float.h>
#include <math.h>
#include "naws_common.h"

thread_local int g_naws_last_hip_error = 0;

extern "C" const char* naws_version(void) { return "naws-hip 0.1 (gfx950)"; }
extern "C" int naws_last_hip_error(void) { return g_naws_last_hip_error; }

// ---- the library's only process-wide state ------------------------------------------------------
// (1) which (kernel, device) pairs have had their dynamic-LDS limit raised: a cache of an
// idempotent driver call, correct for any number of devices and host threads; (2) the A/B knobs.
#include <atomic>
#include <mutex>
#include <string.h>
namespace {
constexpr int LDS_SLOTS = 512;           // > the number of big-LDS kernel instantiations
constexpr int MAX_DEV = 256;
struct LdsSlot {
  std::atomic<const void*> kernel;
  std::atomic<unsigned long long> dev[MAX_DEV / 64];
};
LdsSlot g_lds_slots[LDS_SLOTS];
std::mutex g_lds_mutex;
std::atomic<int> g_knobs[NAWS_KNOB_COUNT] = {{0}, {0}, {0}, {11}, {0}, {42}, {0}, {0}, {0}};
const char* const g_knob_names[NAWS_KNOB_COUNT] = {"gemm", "x3", "h2", "conv_ring", "conv_bn",
                                                    "roi_nw", "wino", "split", "sgd_wgs"};
}  // namespace

// Every launcher of a > 64 KB-LDS kernel passes through here, from any host thread (per-image conv
// chains, the update stream): the common case - this (kernel, device) pair was raised before - is
// decided by two atomic loads, no lock.  A slot's kernel pointer is claimed once under the mutex and
// stays until naws_launch_state_reset(); a device bit is published (release) only after the driver
// call returned, so a reader that sees it (acquire) may launch.  A reader racing a reset sees either
// state; the worst case is one repeated, idempotent hipFuncSetAttribute.
int naws_allow_lds_impl(const void* kernel, int bytes) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return naws_check_launch();
  const bool in_range = dev >= 0 && dev < MAX_DEV;
  const size_t h0 = ((uintptr_t)kernel >> 4) % LDS_SLOTS;
  if (in_range) {
    size_t h = h0;
    for (int probe = 0; probe < LDS_SLOTS; ++probe, h = (h + 1) % LDS_SLOTS) {
      const void* k = g_lds_slots[h].kernel.load(std::memory_order_acquire);
      if (k == kernel) {
        if ((g_lds_slots[h].dev[dev >> 6].load(std::memory_order_acquire) >> (dev & 63)) & 1ULL)
          return NAWS_OK;
        break;
      }
      if (k == nullptr) break;
    }
  }
  std::lock_guard<std::mutex> lock(g_lds_mutex);
  LdsSlot* slot = nullptr;
  if (in_range) {
    size_t h = h0;
    for (int probe = 0; probe < LDS_SLOTS; ++probe, h = (h + 1) % LDS_SLOTS) {
      const void* k = g_lds_slots[h].kernel.load(std::memory_order_relaxed);
      if (k == kernel) { slot = &g_lds_slots[h]; break; }
      if (k == nullptr) {
        slot = &g_lds_slots[h];
        slot->kernel.store(kernel, std::memory_order_release);
        break;
      }
    }
  }
  // (a full table or an out-of-range ordinal just means: set it on every launch)
  if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess)
    return naws_check_launch();
  if (slot) slot->dev[dev >> 6].fetch_or(1ULL << (dev & 63), std::memory_order_release);
  return NAWS_OK;
}

extern "C" int naws_launch_state_reset(void) {
  std::lock_guard<std::mutex> lock(g_lds_mutex);
  for (int i = 0; i < LDS_SLOTS; ++i) {
    for (int w = 0; w < MAX_DEV / 64; ++w) g_lds_slots[i].dev[w].store(0, std::memory_order_relaxed);
    g_lds_slots[i].kernel.store(nullptr, std::memory_order_release);
  }
  return NAWS_OK;
}

int naws_knob(int knob) { return g_knobs[knob].load(std::memory_order_relaxed); }

extern "C" int naws_set_variant(const char* knob, int value) {
  NAWS_REQUIRE_PTR(knob);
  for (int i = 0; i < NAWS_KNOB_COUNT; ++i)
    if (strcmp(knob, g_knob_names[i]) == 0) {
      g_knobs[i].store(value, std::memory_order_relaxed);
      return NAWS_OK;
    }
  return NAWS_ERR_ARG;
}

extern "C" int naws_stream_create(int priority, const uint32_t* cu_mask, int mask_words,
                                  void** stream) {
  NAWS_REQUIRE_PTR(stream);
  if (mask_words < 0 || (mask_words > 0 && !cu_mask)) return NAWS_ERR_ARG;
  hipStream_t s = nullptr;
  if (mask_words > 0) {
    // (a masked stream takes the default priority: the runtime has no call that sets both)
    if (hipExtStreamCreateWithCUMask(&s, (uint32_t)mask_words, cu_mask) != hipSuccess) {
      (void)hipGetLastError();
      return NAWS_ERR_LAUNCH;
    }
  } else {
    int least = 0, greatest = 0;   // numerically: least >= greatest
    if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) {
      (void)hipGetLastError();
      return NAWS_ERR_LAUNCH;
    }
    int pr = priority > 0 ? least : (priority < 0 ? greatest : 0);
    if (pr > least) pr = least;
    if (pr < greatest) pr = greatest;
    if (hipStreamCreateWithPriority(&s, hipStreamNonBlocking, pr) != hipSuccess) {
      (void)hipGetLastError();
      return NAWS_ERR_LAUNCH;
    }
  }
  *stream = s;
  return NAWS_OK;
}

// ---- measurement aid: a stand-in for an RCCL collective on a one-GPU box --------------------------
// `cus` workgroups copy `bytes` from src to dst at a paced aggregate rate (the link rate a ring
// all-reduce would see), i.e. they hold `cus` compute units and move the exchange's bytes through
// HBM for as long as the collective would last.  bench.py --emulate-exchange queues it where the
// reducer queues its all-reduce; nothing on the product path calls it.
namespace {
__global__ void __launch_bounds__(256) exchange_proxy_kernel(const float4* __restrict__ src,
                                                             float4* __restrict__ dst, int64_t n16,
                                                             double ticks_per_chunk) {
  // 64 KB per workgroup and round (16 float4 per thread in flight): at ~2 us per load -> store round
  // trip that sustains ~30 GB/s per workgroup, above any pace the callers ask of 32 workgroups
  constexpr int U = 16, CHUNK = 256 * U;
  const int64_t per = ((n16 + gridDim.x - 1) / gridDim.x + CHUNK - 1) / CHUNK * CHUNK;
  const int64_t i0 = (int64_t)blockIdx.x * per;
  const int64_t i1 = i0 + per < n16 ? i0 + per : n16;
  const uint64_t t0 = wall_clock64();
  int64_t k = 0;
  for (int64_t i = i0; i < i1; i += CHUNK, ++k) {
    float4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t j = i + u * 256 + threadIdx.x;
      v[u] = j < i1 ? src[j] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t j = i + u * 256 + threadIdx.x;
      if (j < i1) dst[j] = v[u];
    }
    const uint64_t due = (uint64_t)((double)(k + 1) * ticks_per_chunk);
    while (wall_clock64() - t0 < due) __builtin_amdgcn_s_sleep(4);
  }
}
}  // namespace

extern "C" int naws_emulate_exchange(const void* src, void* dst, int64_t bytes, int cus,
                                     float gbytes_per_sec, void* stream) {
  NAWS_REQUIRE_PTR(src);
  NAWS_REQUIRE_PTR(dst);
  if (bytes < 0 || bytes % 16 != 0 || cus < 1 || !(gbytes_per_sec > 0.f)) return NAWS_ERR_ARG;
  if (bytes == 0) return NAWS_OK;
  int dev = 0, khz = 0;
  if (hipGetDevice(&dev) != hipSuccess ||
      hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) != hipSuccess || khz <= 0)
    return naws_check_launch();
  // one workgroup moves 64 KB per round at gbytes_per_sec / cus
  const double ticks = 65536.0 * cus / (gbytes_per_sec * 1e9) * (khz * 1e3);
  exchange_proxy_kernel<<<cus, 256, 0, (hipStream_t)stream>>>((const float4*)src, (float4*)dst,
                                                              bytes / 16, ticks);
  return naws_check_launch();
}

extern "C" int naws_stream_destroy(void* stream) {
  if (!stream) return NAWS_OK;
  if (hipStreamDestroy((hipStream_t)stream) != hipSuccess) {
    (void)hipGetLastError();
    return NAWS_ERR_LAUNCH;
  }
  return NAWS_OK;
}

namespace {

constexpr int TB = 256;

// ---- conv1_1: NCHW (3 ch) -> NHWC (Cout), 3x3 pad 1 -------------------------
// One lane = one pixel x 16 output channels; the 27 input taps are loaded once
// into registers, weights come from LDS ([27][Cout], broadcast reads).
template <int OC_PER>
__global__ __launch_bounds__(TB) void conv_c3_kernel(const float* __restrict__ X,
                                                     const float* __restrict__ Wt,
                                                     const float* __restrict__ bias, int N, int H,
                                                     int W, int Cout, int relu,
                                                     float* __restrict__ Y) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* wl = reinterpret_cast<float*>(smem_raw);  // [27][Cout]
  float* bl = wl + 27 * Cout;
  for (int i = threadIdx.x; i < 27 * Cout; i += TB) {
    const int k = i / Cout, o = i % Cout;  // k = (c*3+kh)*3+kw : OIHW -> [k][o]
    wl[i] = Wt[o * 27 + k];
  }
  for (int i = threadIdx.x; i < Cout; i += TB) bl[i] = bias ? bias[i] : 0.f;
  __syncthreads();
  const int groups = Cout / OC_PER;
  const int64_t total = (int64_t)N * H * W * groups;
  for (int64_t t = (int64_t)blockIdx.x * TB + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * TB) {
    const int g = (int)(t % groups);
    const int64_t pix = t / groups;
    const int x = (int)(pix % W);
    const int y = (int)((pix / W) % H);
    const int n = (int)(pix / ((int64_t)W * H));
    float in[27];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          const int yy = y + kh - 1, xx = x + kw - 1;
          const bool ok = (yy >= 0) && (yy < H) && (xx >= 0) && (xx < W);
          in[(c * 3 + kh) * 3 + kw] =
              ok ? X[(((int64_t)n * 3 + c) * H + yy) * W + xx] : 0.f;
        }
    float acc[OC_PER];
#pragma unroll
    for (int o = 0; o < OC_PER; ++o) acc[o] = 0.f;
#pragma unroll
    for (int k = 0; k < 27; ++k) {
      const float v = in[k];
      const float* wr = wl + k * Cout + g * OC_PER;
#pragma unroll
      for (int o = 0; o < OC_PER; ++o) acc[o] = fmaf(v, wr[o], acc[o]);
    }
    float* out = Y + pix * Cout + g * OC_PER;
#pragma unroll
    for (int o = 0; o < OC_PER; o += 4) {
      float4 r;
      r.x = acc[o + 0] + bl[g * OC_PER + o + 0];
      r.y = acc[o + 1] + bl[g * OC_PER + o + 1];
      r.z = acc[o + 2] + bl[g * OC_PER + o + 2];
      r.w = acc[o + 3] + bl[g * OC_PER + o + 3];
      if (relu) { r.x = fmaxf(r.x, 0.f); r.y = fmaxf(r.y, 0.f); r.z = fmaxf(r.z, 0.f); r.w = fmaxf(r.w, 0.f); }
      *reinterpret_cast<float4*>(out + o) = r;
    }
  }
}


// ---- conv1_1, Cout = 64: LDS-tiled sliding-window kernel ------------------------------------
// 3 input channels give the MFMA nothing to contract over (K = 27), so this layer runs on the
// vector ALU, whose floor (1.04 G lane-FMA per 600x1000 image) is ~26 us and equals the time to
// write the 154 MB NHWC result.  A workgroup owns an 8-row x 64-column pixel tile: the 3x10x66
// input halo is staged in LDS once (zero-filled outside the image, so the inner loop is
// branch-free).  Sixteen adjacent lanes own one pixel: lane q keeps the 27x4 weights of output
// channels 4q..4q+3 in registers for the whole tile and the group stores 256 contiguous bytes
// per pixel.  Each 16-lane group walks a 32-pixel strip of one row, keeping the 3x3x3 input
// window in registers and reading only the 9 new values per step (LDS broadcast reads).
constexpr int C3_TW = 64, C3_TH = 8, C3_STRIP = 32, C3_LD = C3_TW + 4;  // row stride 68 floats
__global__ __launch_bounds__(256) void conv_c3_tile_kernel(const float* __restrict__ X,
                                                           const float* __restrict__ Wt,
                                                           const float* __restrict__ bias, int H,
                                                           int W, int relu, float* __restrict__ Y) {
  __shared__ float tile[3][C3_TH + 2][C3_LD];
  __shared__ __attribute__((aligned(16))) float wl[27][64];
  const int x0 = blockIdx.x * C3_TW, y0 = blockIdx.y * C3_TH, n = blockIdx.z;
  const float* Xn = X + (int64_t)n * 3 * H * W;
  for (int i = threadIdx.x; i < 3 * (C3_TH + 2) * (C3_TW + 2); i += 256) {
    const int col = i % (C3_TW + 2);
    const int r = (i / (C3_TW + 2)) % (C3_TH + 2);
    const int c = i / ((C3_TW + 2) * (C3_TH + 2));
    const int yy = y0 + r - 1, xx = x0 + col - 1;
    const bool ok = (yy >= 0) && (yy < H) && (xx >= 0) && (xx < W);
    const int yc = min(max(yy, 0), H - 1), xc = min(max(xx, 0), W - 1);
    const float v = Xn[((int64_t)c * H + yc) * W + xc];
    tile[c][r][col] = ok ? v : 0.f;
  }
  for (int i = threadIdx.x; i < 27 * 64; i += 256) wl[i % 27][i / 27] = Wt[i];  // OIHW -> [k][o]
  __syncthreads();
  const int q = threadIdx.x & 15, grp = threadIdx.x >> 4;
  const int row = grp & 7, strip = grp >> 3;
  float w[27][4];
#pragma unroll
  for (int k = 0; k < 27; ++k) {
    const float4 t = *reinterpret_cast<const float4*>(&wl[k][4 * q]);
    w[k][0] = t.x; w[k][1] = t.y; w[k][2] = t.z; w[k][3] = t.w;
  }
  float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
  if (bias) bv = *reinterpret_cast<const float4*>(bias + 4 * q);
  const int y = y0 + row;
  const int cx = strip * C3_STRIP;                 // tile column of this strip's first window
  float win[3][3][3];                              // [c][kh][slot], slot = tile column mod 3
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      win[c][kh][0] = tile[c][row + kh][cx];
      win[c][kh][1] = tile[c][row + kh][cx + 1];
    }
  float* yrow = Y + (((int64_t)n * H + y) * W + x0 + cx) * 64 + 4 * q;
#pragma unroll
  for (int i = 0; i < C3_STRIP; ++i) {
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int kh = 0; kh < 3; ++kh) win[c][kh][(i + 2) % 3] = tile[c][row + kh][cx + i + 2];
    float a0 = bv.x, a1 = bv.y, a2 = bv.z, a3 = bv.w;
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          const float v = win[c][kh][(i + kw) % 3];
          const int k = (c * 3 + kh) * 3 + kw;
          a0 = fmaf(v, w[k][0], a0); a1 = fmaf(v, w[k][1], a1);
          a2 = fmaf(v, w[k][2], a2); a3 = fmaf(v, w[k][3], a3);
        }
    if (relu) { a0 = fmaxf(a0, 0.f); a1 = fmaxf(a1, 0.f); a2 = fmaxf(a2, 0.f); a3 = fmaxf(a3, 0.f); }
    if (y < H && x0 + cx + i < W)
      *reinterpret_cast<float4*>(yrow + (int64_t)i * 64) = make_float4(a0, a1, a2, a3);
  }
}

// ---- weight repack OIHW -> [O][kh][kw][I] ----------------------------------
__global__ void pack_weight_kernel(const float* __restrict__ Wi, int Cout, int Cin,
                                   float* __restrict__ Wo) {
  const int64_t total = (int64_t)Cout * Cin * 9;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % Cin);
    const int tap = (int)((i / Cin) % 9);
    const int o = (int)(i / ((int64_t)Cin * 9));
    Wo[i] = Wi[((int64_t)o * Cin + c) * 9 + tap];
  }
}

// ---- 2x2 max pool, NHWC, float4 lanes along channels -----------------------
__global__ __launch_bounds__(TB) void maxpool2_kernel(const float4* __restrict__ X, int N, int H,
                                                      int W, int C4, int stride, int Ho, int Wo,
                                                      float4* __restrict__ Y) {
  const int64_t total = (int64_t)N * Ho * Wo * C4;
  for (int64_t t = (int64_t)blockIdx.x * TB + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * TB) {
    const int c = (int)(t % C4);
    const int xo = (int)((t / C4) % Wo);
    const int yo = (int)((t / ((int64_t)C4 * Wo)) % Ho);
    const int n = (int)(t / ((int64_t)C4 * Wo * Ho));
    const int y = yo * stride, x = xo * stride;
    const float4* p = X + (((int64_t)n * H + y) * W + x) * C4 + c;
    const float4 a = p[0], b = p[C4], d = p[(int64_t)W * C4], e = p[(int64_t)W * C4 + C4];
    float4 r;
    r.x = fmaxf(fmaxf(a.x, b.x), fmaxf(d.x, e.x));
    r.y = fmaxf(fmaxf(a.y, b.y), fmaxf(d.y, e.y));
    r.z = fmaxf(fmaxf(a.z, b.z), fmaxf(d.z, e.z));
    r.w = fmaxf(fmaxf(a.w, b.w), fmaxf(d.w, e.w));
    Y[t] = r;
  }
}

// ---- NCHW <-> NHWC via 32x32 LDS tiles over (C, HW) ------------------------
// in: [batch][rows][cols] -> out: [batch][cols][rows]
__global__ void transpose_batched_kernel(const float* __restrict__ X, int rows, int cols,
                                         float* __restrict__ Y) {
  __shared__ float tile[32][33];
  const int64_t boff = (int64_t)blockIdx.z * rows * cols;
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  for (int i = threadIdx.y; i < 32; i += blockDim.y) {
    const int r = r0 + i, c = c0 + threadIdx.x;
    if (r < rows && c < cols) tile[i][threadIdx.x] = X[boff + (int64_t)r * cols + c];
  }
  __syncthreads();
  for (int i = threadIdx.y; i < 32; i += blockDim.y) {
    const int c = c0 + i, r = r0 + threadIdx.x;
    if (r < rows && c < cols) Y[boff + (int64_t)c * rows + r] = tile[threadIdx.x][i];
  }
}

// ---- small built-ins --------------------------------------------------------
__global__ void unary_kernel(int op, const float* __restrict__ X, int64_t n, float a, float b,
                             float* __restrict__ Y) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const float x = X[i];
    float y;
    switch (op) {
      case NAWS_UN_LOG: y = logf(x); break;
      case NAWS_UN_SCALE: y = x * a; break;
      case NAWS_UN_REPLACE_NAN: y = isnan(x) ? a : x; break;
      case NAWS_UN_LEAKY_RELU: y = x >= 0.f ? x : a * x; break;
      case NAWS_UN_CLIP: y = (x < a) ? a : x; y = (y > b) ? b : y; break;
      default: y = fmaxf(x, 0.f); break;
    }
    Y[i] = y;
  }
}

__global__ void binary_kernel(int op, const float* __restrict__ A, int rowsA, int colsA,
                              const float* __restrict__ B, int rowsB, int colsB,
                              float* __restrict__ Y, int rows, int cols) {
  const int64_t n = (int64_t)rows * cols;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int r = (int)(i / cols), c = (int)(i % cols);
    const float a = A[(int64_t)(rowsA == 1 ? 0 : r) * colsA + (colsA == 1 ? 0 : c)];
    const float b = B[(int64_t)(rowsB == 1 ? 0 : r) * colsB + (colsB == 1 ? 0 : c)];
    float y;
    switch (op) {
      case NAWS_BIN_ADD: y = a + b; break;
      case NAWS_BIN_SUB: y = a - b; break;
      case NAWS_BIN_MUL: y = a * b; break;
      case NAWS_BIN_DIV: y = a / b; break;
      default: y = b > 0.f ? a : 0.f; break;
    }
    Y[i] = y;
  }
}

// one wave per row
__global__ __launch_bounds__(TB) void softmax_rows_kernel(const float* __restrict__ X, int rows,
                                                          int cols, float* __restrict__ Y) {
  const int row = blockIdx.x * (TB / 64) + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float* x = X + (int64_t)row * cols;
  float m = -INFINITY;
  for (int c = lane; c < cols; c += 64) m = fmaxf(m, x[c]);
  m = wave_max(m);
  float s = 0.f;
  for (int c = lane; c < cols; c += 64) s += expf(x[c] - m);
  s = wave_sum(s);
  float* y = Y + (int64_t)row * cols;
  for (int c = lane; c < cols; c += 64) y[c] = expf(x[c] - m) / s;
}

__global__ __launch_bounds__(TB) void softmax_rows_bwd_kernel(const float* __restrict__ Y,
                                                              const float* __restrict__ dY,
                                                              int rows, int cols,
                                                              float* __restrict__ dX) {
  const int row = blockIdx.x * (TB / 64) + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float* y = Y + (int64_t)row * cols;
  const float* dy = dY + (int64_t)row * cols;
  float d = 0.f;
  for (int c = lane; c < cols; c += 64) d += y[c] * dy[c];
  d = wave_sum(d);
  float* dx = dX + (int64_t)row * cols;
  for (int c = lane; c < cols; c += 64) dx[c] = y[c] * (dy[c] - d);
}

// Y[c] (+)= sum_r X[r*ld + c].  One workgroup per 32 columns: 8 float4 column
// groups x 32 row lanes, fixed-order LDS tree over the row lanes (deterministic).
__global__ __launch_bounds__(TB) void colsum4_kernel(const float* __restrict__ X, int M, int N,
                                                     int ld, float* __restrict__ Y,
                                                     int accumulate) {
  __shared__ float4 part[32][8];
  const int cg = threadIdx.x & 7, rl = threadIdx.x >> 3;
  const int c = blockIdx.x * 32 + cg * 4;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (c < N) {
    const float* p = X + c;
    int r = rl;
    for (; r + 96 < M; r += 128) {  // 4 independent loads in flight
      const float4 a = *reinterpret_cast<const float4*>(p + (int64_t)r * ld);
      const float4 b = *reinterpret_cast<const float4*>(p + (int64_t)(r + 32) * ld);
      const float4 d = *reinterpret_cast<const float4*>(p + (int64_t)(r + 64) * ld);
      const float4 e = *reinterpret_cast<const float4*>(p + (int64_t)(r + 96) * ld);
      acc.x += (a.x + b.x) + (d.x + e.x); acc.y += (a.y + b.y) + (d.y + e.y);
      acc.z += (a.z + b.z) + (d.z + e.z); acc.w += (a.w + b.w) + (d.w + e.w);
    }
    for (; r < M; r += 32) {
      const float4 a = *reinterpret_cast<const float4*>(p + (int64_t)r * ld);
      acc.x += a.x; acc.y += a.y; acc.z += a.z; acc.w += a.w;
    }
  }
  part[rl][cg] = acc;
  __syncthreads();
  for (int s = 16; s > 0; s >>= 1) {
    if (rl < s) {
      float4 a = part[rl][cg];
      const float4 b = part[rl + s][cg];
      a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
      part[rl][cg] = a;
    }
    __syncthreads();
  }
  if (rl == 0 && c < N) {
    float4 t = part[0][cg];
    float* y = Y + c;
    if (accumulate) { t.x += y[0]; t.y += y[1]; t.z += y[2]; t.w += y[3]; }
    y[0] = t.x; y[1] = t.y; y[2] = t.z; y[3] = t.w;
  }
}

// scalar fallback (N or ld not a multiple of 4): one workgroup per 64 columns
__global__ __launch_bounds__(TB) void colsum_kernel(const float* __restrict__ X, int M, int N,
                                                    int ld, float* __restrict__ Y,
                                                    int accumulate) {
  __shared__ float part[TB / 64][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  const int rl = threadIdx.x >> 6;
  float acc = 0.f;
  if (c < N)
    for (int r = rl; r < M; r += TB / 64) acc += X[(int64_t)r * ld + c];
  part[rl][threadIdx.x & 63] = acc;
  __syncthreads();
  if (rl == 0 && c < N) {
    float t = part[0][threadIdx.x];
#pragma unroll
    for (int i = 1; i < TB / 64; ++i) t += part[i][threadIdx.x];
    Y[c] = accumulate ? Y[c] + t : t;
  }
}

__global__ void dropout_mask_kernel(uint64_t seed, uint32_t thr, int64_t n,
                                    float* __restrict__ mask) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x)
    mask[i] = naws_keep(seed, (uint64_t)i, thr) ? 1.0f : 0.0f;
}

inline int grid_for(int64_t n, int tb = TB, int cap = 256 * 8) {
  return (int)std::min<int64_t>(std::max<int64_t>(naws_cdiv(n, tb), 1), cap);
}

}  // namespace

extern "C" int naws_conv3x3_c3_nchw_to_nhwc_fwd(const float* X, const float* Wt,
                                                const float* bias, int N, int H, int W, int Cout,
                                                int relu, float* Y, void* stream) {
  if (N <= 0 || H <= 0 || W <= 0 || Cout <= 0) return NAWS_ERR_SHAPE;
  if (Cout % 16 != 0 || ((uintptr_t)Y % 16) != 0) return NAWS_ERR_ARG;
  NAWS_REQUIRE_PTR(X); NAWS_REQUIRE_PTR(Wt); NAWS_REQUIRE_PTR(Y);
  if (Cout == 64 && N <= 65535 && naws_cdiv(H, C3_TH) <= 65535) {
    dim3 grid((unsigned)naws_cdiv(W, C3_TW), (unsigned)naws_cdiv(H, C3_TH), (unsigned)N);
    hipLaunchKernelGGL(conv_c3_tile_kernel, grid, dim3(256), 0, (hipStream_t)stream, X, Wt, bias,
                       H, W, relu, Y);
    return naws_check_launch();
  }
  const size_t lds = (size_t)(27 + 1) * Cout * sizeof(float);
  const int64_t total = (int64_t)N * H * W * (Cout / 16);
  hipLaunchKernelGGL(conv_c3_kernel<16>, dim3(grid_for(total, TB, 256 * 16)), dim3(TB), lds,
                     (hipStream_t)stream, X, Wt, bias, N, H, W, Cout, relu, Y);
  return naws_check_launch();
}

extern "C" int naws_conv3x3_pack_weight(const float* W_oihw, int Cout, int Cin, float* W_packed,
                                        void* stream) {
  if (Cout <= 0 || Cin <= 0) return NAWS_ERR_SHAPE;
  NAWS_REQUIRE_PTR(W_oihw); NAWS_REQUIRE_PTR(W_packed);
  hipLaunchKernelGGL(pack_weight_kernel, dim3(grid_for((int64_t)Cout * Cin * 9)), dim3(TB), 0,
                     (hipStream_t)stream, W_oihw, Cout, Cin, W_packed);
  return naws_check_launch();
}

extern "C" int naws_maxpool2x2_nhwc_fwd(const float* X, int N, int H, int W, int C, int stride,
                                        float* Y, void* stream) {
  if (N <= 0 || H < 2 || W < 2 || C <= 0) return NAWS_ERR_SHAPE;
  if (stride != 1 && stride != 2) return NAWS_ERR_ARG;
  if (C % 4 != 0 || (((uintptr_t)X | (uintptr_t)Y) % 16) != 0) return NAWS_ERR_ARG;
  NAWS_REQUIRE_PTR(X); NAWS_REQUIRE_PTR(Y);
  const int Ho = (H - 2) / stride + 1, Wo = (W - 2) / stride + 1;
  const int64_t total = (int64_t)N * Ho * Wo * (C / 4);
  hipLaunchKernelGGL(maxpool2_kernel, dim3(grid_for(total, TB, 256 * 16)), dim3(TB), 0,
                     (hipStream_t)stream, (const float4*)X, N, H, W, C / 4, stride, Ho, Wo,
                     (float4*)Y);
  return naws_check_launch();
}

static int transpose_batched(const float* X, int batch, int rows, int cols, float* Y,
                             void* stream) {
  if (batch <= 0 || rows <= 0 || cols <= 0) return NAWS_ERR_SHAPE;
  NAWS_REQUIRE_PTR(X); NAWS_REQUIRE_PTR(Y);
  if (batch > 65535 || naws_cdiv(rows, 32) > 65535) return NAWS_ERR_UNSUPPORTED;
  dim3 grid((unsigned)naws_cdiv(cols, 32), (unsigned)naws_cdiv(rows, 32), batch);
  hipLaunchKernelGGL(transpose_batched_kernel, grid, dim3(32, 8), 0, (hipStream_t)stream, X,
                     rows, cols, Y);
  return naws_check_launch();
}

extern "C" int naws_nchw_to_nhwc(const float* X, int N, int C, int H, int W, float* Y,
                                 void* stream) {
  if (H <= 0 || W <= 0) return NAWS_ERR_SHAPE;
  return transpose_batched(X, N, C, H * W, Y, stream);
}
extern "C" int naws_nhwc_to_nchw(const float* X, int N, int H, int W, int C, float* Y,
                                 void* stream) {
  if (H <= 0 || W <= 0) return NAWS_ERR_SHAPE;
  return transpose_batched(X, N, H * W, C, Y, stream);
}
extern "C" int naws_transpose2d_f32(const float* X, int rows, int cols, float* Y, void* stream) {
  return transpose_batched(X, 1, rows, cols, Y, stream);
}

extern "C" int naws_unary_f32(int op, const float* X, int64_t n, float a, float b, float* Y,
                              void* stream) {
  if (n < 0) return NAWS_ERR_SHAPE;
  if (op < NAWS_UN_LOG || op > NAWS_UN_RELU) return NAWS_ERR_ARG;
  if (n == 0) return NAWS_OK;
  NAWS_REQUIRE_PTR(X); NAWS_REQUIRE_PTR(Y);
  hipLaunchKernelGGL(unary_kernel, dim3(grid_for(n)), dim3(TB), 0, (hipStream_t)stream, op, X, n,
                     a, b, Y);
  return naws_check_launch();
}

extern "C" int naws_binary_f32(int op, const float* A, int rowsA, int colsA, const float* B,
                               int rowsB, int colsB, float* Y, int rows, int cols, void* stream) {
  if (rows <= 0 || cols <= 0) return NAWS_ERR_SHAPE;
  if (op < NAWS_BIN_ADD || op > NAWS_BIN_GATE_POS) return NAWS_ERR_ARG;
  if ((rowsA != 1 && rowsA != rows) || (rowsB != 1 && rowsB != rows) ||
      (colsA != 1 && colsA != cols) || (colsB != 1 && colsB != cols))
    return NAWS_ERR_SHAPE;
  NAWS_REQUIRE_PTR(A); NAWS_REQUIRE_PTR(B); NAWS_REQUIRE_PTR(Y);
  hipLaunchKernelGGL(binary_kernel, dim3(grid_for((int64_t)rows * cols)), dim3(TB), 0,
                     (hipStream_t)stream, op, A, rowsA, colsA, B, rowsB, colsB, Y, rows, cols);
  return naws_check_launch();
}

extern "C" int naws_softmax_rows_fwd(const float* X, int rows, int cols, float* Y, void* stream) {
  if (rows <= 0 || cols <= 0) return NAWS_ERR_SHAPE;
  NAWS_REQUIRE_PTR(X); NAWS_REQUIRE_PTR(Y);
  hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)naws_cdiv(rows, TB / 64)), dim3(TB), 0,
                     (hipStream_t)stream, X, rows, cols, Y);
  return naws_check_launch();
}
extern "C" int naws_softmax_rows_bwd(const float* Y, const float* dY, int rows, int cols,
                                     float* dX, void* stream) {
  if (rows <= 0 || cols <= 0) return NAWS_ERR_SHAPE;
  NAWS_REQUIRE_PTR(Y); NAWS_REQUIRE_PTR(dY); NAWS_REQUIRE_PTR(dX);
  hipLaunchKernelGGL(softmax_rows_bwd_kernel, dim3((unsigned)naws_cdiv(rows, TB / 64)), dim3(TB),
                     0, (hipStream_t)stream, Y, dY, rows, cols, dX);
  return naws_check_launch();
}

extern "C" int naws_colsum_f32(const float* dY, int M, int N, int ld, float* db, int accumulate,
                               void* stream) {
  if (M <= 0 || N <= 0 || ld < N) return NAWS_ERR_SHAPE;
  NAWS_REQUIRE_PTR(dY); NAWS_REQUIRE_PTR(db);
  if (N % 4 == 0 && ld % 4 == 0 && ((uintptr_t)dY % 16) == 0)
    hipLaunchKernelGGL(colsum4_kernel, dim3((unsigned)naws_cdiv(N, 32)), dim3(TB), 0,
                       (hipStream_t)stream, dY, M, N, ld, db, accumulate);
  else
    hipLaunchKernelGGL(colsum_kernel, dim3((unsigned)naws_cdiv(N, 64)), dim3(TB), 0,
                       (hipStream_t)stream, dY, M, N, ld, db, accumulate);
  return naws_check_launch();
}
extern "C" int naws_reduce_sum_axis0(const float* X, int rows, int cols, float* Y, void* stream) {
  return naws_colsum_f32(X, rows, cols, cols, Y, 0, stream);
}

extern "C" int naws_dropout_mask(uint64_t seed, float drop_ratio, int64_t n, float* mask,
                                 void* stream) {
  if (n < 0) return NAWS_ERR_SHAPE;
  if (!(drop_ratio >= 0.f && drop_ratio < 1.f)) return NAWS_ERR_ARG;
  if (n == 0) return NAWS_OK;
  NAWS_REQUIRE_PTR(mask);
  hipLaunchKernelGGL(dropout_mask_kernel, dim3(grid_for(n)), dim3(TB), 0, (hipStream_t)stream,
                     seed, naws_drop_threshold(drop_ratio), n, mask);
  return naws_check_launch();
}

// ---- image preparation of the minibatch loader on the GPU ------------------
// replaces, per image: detectron/roi_data/minibatch_wsl.py:121-157 (flip, crop) and
// detectron/utils/blob.py:100-131 prep_im_for_blob (float32, - PIXEL_MEANS, / PIXEL_STDS,
// cv2.resize(fx = fy = im_scale, INTER_LINEAR)) + blob.py:67-97 im_list_to_blob (HWC -> CHW into
// the zero-padded batch blob).  cv2.resize's float INTER_LINEAR path is restated from OpenCV's
// published algorithm (imgproc/resize.cpp: pixel-centre mapping fx = (dx+0.5)/im_scale - 0.5 in
// double -> float, floor, edge taps clamped with weight 1, horizontal pass then vertical pass).
namespace {
// cv2 8-bit BGR->HSV->(S*=sat, V*=expo)->BGR round trip of one pixel (WSL.USE_DISTORTION,
// minibatch_wsl.py:127-138), restated from OpenCV's RGB2HSV_b (fixed point, hsv_shift 12) and
// HSV2RGB_b (float sector formula, hscale 6/180, rounded store).
__device__ __forceinline__ void hsv_jitter(int& b, int& g, int& r, float sat, float expo) {
#pragma clang fp contract(off)
  const int v = max(max(b, g), r), vmin = min(min(b, g), r);
  const int diff = v - vmin;
  const int sdiv = v ? (int)rint((double)(255 << 12) / (double)v) : 0;
  const int hdiv = diff ? (int)rint((double)(180 << 12) / (6.0 * (double)diff)) : 0;
  int s = (diff * sdiv + 2048) >> 12;
  int h = (v == r) ? (g - b) : ((v == g) ? (b - r + 2 * diff) : (r - g + 4 * diff));
  h = (h * hdiv + 2048) >> 12;
  if (h < 0) h += 180;
  // float32 scaling, cap at 255, truncate to uint8 (np.array(hsv, dtype=np.uint8))
  const float sf0 = sat * (float)s, vf0 = expo * (float)v;
  const int s8 = (int)fminf(sf0, 255.f), v8 = (int)fminf(vf0, 255.f);
  const float sf = (float)s8 * (float)(1.0 / 255.0), vf = (float)v8 * (float)(1.0 / 255.0);
  float hf = (float)h * (float)(6.0 / 180.0);
  if (hf >= 6.f) hf = hf - 6.f;
  int sec = (int)floorf(hf);
  float fr = hf - (float)sec;
  if (sec < 0 || sec >= 6) { sec = 0; fr = 0.f; }
  float tab[4];
  tab[0] = vf;
  const float t1 = 1.f - sf;
  tab[1] = vf * t1;
  const float sh = sf * fr, t2 = 1.f - sh;
  tab[2] = vf * t2;
  const float omf = 1.f - fr, so = sf * omf, t3 = 1.f - so;
  tab[3] = vf * t3;
  const int sb[6] = {1, 1, 3, 0, 0, 2}, sg[6] = {3, 0, 0, 2, 1, 1}, sr[6] = {0, 2, 1, 1, 3, 0};
  float fb = tab[sb[sec]], fg = tab[sg[sec]], frr = tab[sr[sec]];
  if (s8 == 0) { fb = vf; fg = vf; frr = vf; }
  const float xb = fb * 255.f, xg = fg * 255.f, xr = frr * 255.f;
  b = min(max((int)rintf(xb), 0), 255);
  g = min(max((int)rintf(xg), 0), 255);
  r = min(max((int)rintf(xr), 0), 255);
}

__global__ __launch_bounds__(256) void prep_image_kernel(
    const unsigned char* __restrict__ im, int W, int flip, int cy0, int cx0, int ch, int cw,
    float m0, float m1, float m2, float s0, float s1, float s2, double inv_scale, int oh, int ow,
    long long plane_stride, int row_stride, int distort, float sat, float expo,
    float* __restrict__ out) {
  // Every product and sum is rounded separately, like the numpy / cv2 passes.  (hipcc's
  // __fmul_rn & co. are inline `x * y` compiled under the header's contract(fast), so plain
  // operators under this pragma - built with -ffp-contract=fast-honor-pragmas - are used.)
#pragma clang fp contract(off)
  const int dx = blockIdx.x * 256 + threadIdx.x, dy = blockIdx.y;
  if (dx >= ow) return;
  const double px = (dx + 0.5) * inv_scale;
  float fx = (float)(px - 0.5);
  int sx = (int)floorf(fx);
  fx = fx - (float)sx;
  if (sx < 0) { fx = 0.f; sx = 0; }
  if (sx >= cw - 1) { fx = 0.f; sx = cw - 1; }
  const double py = (dy + 0.5) * inv_scale;
  float fy = (float)(py - 0.5);
  int sy = (int)floorf(fy);
  fy = fy - (float)sy;
  if (sy < 0) { fy = 0.f; sy = 0; }
  if (sy >= ch - 1) { fy = 0.f; sy = ch - 1; }
  const int sx1 = min(sx + 1, cw - 1), sy1 = min(sy + 1, ch - 1);
  const float a0 = 1.f - fx, a1 = fx, b0 = 1.f - fy, b1 = fy;
  const float mean[3] = {m0, m1, m2}, sd[3] = {s0, s1, s2};
  const int ys[2] = {cy0 + sy, cy0 + sy1};
  const int xs[2] = {flip ? (W - 1 - (cx0 + sx)) : (cx0 + sx),       // crop is taken on the
                     flip ? (W - 1 - (cx0 + sx1)) : (cx0 + sx1)};    // flipped image
  int pix[2][2][3];
#pragma unroll
  for (int yy = 0; yy < 2; ++yy)
#pragma unroll
    for (int xx = 0; xx < 2; ++xx) {
      const unsigned char* p = im + ((long long)ys[yy] * W + xs[xx]) * 3;
      int b = p[0], g = p[1], r = p[2];
      if (distort) hsv_jitter(b, g, r, sat, expo);
      pix[yy][xx][0] = b; pix[yy][xx][1] = g; pix[yy][xx][2] = r;
    }
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    float v[2][2];
#pragma unroll
    for (int yy = 0; yy < 2; ++yy)
#pragma unroll
      for (int xx = 0; xx < 2; ++xx) {
        const float u = (float)pix[yy][xx][c];
        const float d = u - mean[c];
        v[yy][xx] = d / sd[c];
      }
    const float t00 = v[0][0] * a0, t01 = v[0][1] * a1, t10 = v[1][0] * a0, t11 = v[1][1] * a1;
    const float r0 = t00 + t01, r1 = t10 + t11;
    const float q0 = r0 * b0, q1 = r1 * b1;
    out[c * plane_stride + (long long)dy * row_stride + dx] = q0 + q1;
  }
}
}  // namespace

extern "C" int naws_prep_image_fwd(const uint8_t* im_bgr_hwc, int H, int W, int flip, int crop_y0,
                                   int crop_x0, int crop_h, int crop_w, const float* means3,
                                   const float* stds3, double im_scale, int distort,
                                   float saturation, float exposure, int out_h, int out_w,
                                   int64_t plane_stride, int row_stride, float* out, void* stream) {
  if (H <= 0 || W <= 0 || crop_h <= 0 || crop_w <= 0 || out_h <= 0 || out_w <= 0) return NAWS_ERR_SHAPE;
  if (crop_y0 < 0 || crop_x0 < 0 || crop_y0 + crop_h > H || crop_x0 + crop_w > W) return NAWS_ERR_SHAPE;
  if (!(im_scale > 0.0) || row_stride < out_w || out_h > 65535) return NAWS_ERR_ARG;
  NAWS_REQUIRE_PTR(im_bgr_hwc); NAWS_REQUIRE_PTR(means3); NAWS_REQUIRE_PTR(stds3); NAWS_REQUIRE_PTR(out);
  dim3 grid((unsigned)naws_cdiv(out_w, 256), (unsigned)out_h);
  hipLaunchKernelGGL(prep_image_kernel, grid, dim3(256), 0, (hipStream_t)stream, im_bgr_hwc, W,
                     flip, crop_y0, crop_x0, crop_h, crop_w, means3[0], means3[1], means3[2],
                     stds3[0], stds3[1], stds3[2], 1.0 / im_scale, out_h, out_w,
                     (long long)plane_stride, row_stride, distort, saturation, exposure, out);
  return naws_check_launch();
}

// GraphNorm over a single graph (K6).
// torch_geometric.nn.GraphNorm.forward(x, batch=None) [3P, PyG 2.3.1]; reference call sites
// TwoWL/model/model.py:38,54 (inside Seq, which passes only `out` to it => batch is None):
//   mean = column mean;  o = x - mean * mean_scale;  y = weight * o / sqrt(mean(o^2) + eps) + bias
//
// Three launches: (1) per-block column sums of x and x^2 in fp64 over a FIXED row partition,
// (2) one block folds the partials in block order and emits per-column shift/scale/bias,
// (3) elementwise apply with optional fused ReLU and residual add (the conv2s + conv2s_r sum,
// model.py:77).  fp64 sums make the one-pass variance safe: var = E[x^2] - mean^2*ms*(2-ms).
#include <stdlib.h>

#include "lpg_common.hpp"

namespace {
using namespace lpg;

constexpr int kMaxPartBlocks = 2048;

// Per-block column sums of x and x^2 in fp64.  A thread owns VEC consecutive columns (one 16-byte
// load per row when VEC = 4) and walks its block's rows with a stride of R = 256 / (threads per row),
// four rows per step so four independent loads are in flight; the R partial sums per column are
// folded through LDS in a fixed order.
template <int VEC>
__global__ __launch_bounds__(kBlock) void gn_partial_kernel(const float* __restrict__ x, int64_t n, int64_t d,
                                                            int64_t rows_per_block, int tpr,
                                                            double* __restrict__ part,
                                                            const int64_t* __restrict__ rw = nullptr) {
  __shared__ double sh[2][kBlock][VEC];
  const int R = kBlock / tpr;        // rows handled per sweep
  const int cl = threadIdx.x % tpr, rl = threadIdx.x / tpr;
  const int64_t rb = static_cast<int64_t>(blockIdx.x) * rows_per_block;
  int64_t re = rb + rows_per_block;
  if (re > n) re = n;
  for (int64_t cb = 0; cb < d; cb += static_cast<int64_t>(tpr) * VEC) {
    const int64_t c = cb + static_cast<int64_t>(cl) * VEC;
    double s[VEC], s2[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) s[i] = s2[i] = 0.0;
    if (c < d) {
      int64_t r = rb + rl;
      for (; r + 3 * R < re; r += 4 * R) {
        float v[4][VEC];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if constexpr (VEC == 4) {
            const float4 q = *reinterpret_cast<const float4*>(x + (r + u * R) * d + c);
            v[u][0] = q.x; v[u][1] = q.y; v[u][2] = q.z; v[u][3] = q.w;
          } else {
            v[u][0] = x[(r + u * R) * d + c];
          }
        }
        if (rw) {                              // row r counts rw[r] times (lpg_graphnorm_weighted_f32)
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const double wt = static_cast<double>(rw[r + u * R]);
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
              const double t = static_cast<double>(v[u][i]);
              s[i] += wt * t;
              s2[i] += wt * t * t;
            }
          }
        } else {
#pragma unroll
          for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
              const double t = static_cast<double>(v[u][i]);
              s[i] += t;
              s2[i] += t * t;
            }
        }
      }
      for (; r < re; r += R) {
        const double wt = rw ? static_cast<double>(rw[r]) : 1.0;
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
          const double t = static_cast<double>(x[r * d + c + i]);
          s[i] += wt * t;
          s2[i] += wt * t * t;
        }
      }
    }
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      sh[0][threadIdx.x][i] = s[i];
      sh[1][threadIdx.x][i] = s2[i];
    }
    __syncthreads();
    if (rl == 0 && c < d) {
      for (int q = 1; q < R; ++q)    // fixed order
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
          s[i] += sh[0][q * tpr + cl][i];
          s2[i] += sh[1][q * tpr + cl][i];
        }
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        part[(static_cast<int64_t>(blockIdx.x) * 2 + 0) * d + c + i] = s[i];
        part[(static_cast<int64_t>(blockIdx.x) * 2 + 1) * d + c + i] = s2[i];
      }
    }
    __syncthreads();
  }
}

// a 256-thread workgroup per column: thread t adds the partials of blocks t, t + 256, ... (up to eight independent
// loads), then a fixed LDS tree.  (First version: one wave per column striding up to 2048 partials -- 32 dependent
// round trips, 17 us per call on a [236 K, 32] tensor whose two data passes take 22.)
__global__ __launch_bounds__(kBlock) void gn_finalize_kernel(const double* __restrict__ part, int nblocks, int64_t n,
                                                             int64_t d, const float* __restrict__ weight,
                                                             const float* __restrict__ bias,
                                                             const float* __restrict__ mean_scale, float eps,
                                                             float* __restrict__ stats) {
  __shared__ double sh[2][kBlock];
  const int t = threadIdx.x;
  for (int64_t c = blockIdx.x; c < d; c += gridDim.x) {
    double s = 0.0, s2 = 0.0;
#pragma unroll 4
    for (int b = t; b < nblocks; b += kBlock) {
      s += part[(static_cast<int64_t>(b) * 2 + 0) * d + c];
      s2 += part[(static_cast<int64_t>(b) * 2 + 1) * d + c];
    }
    sh[0][t] = s;
    sh[1][t] = s2;
    __syncthreads();
    for (int o = kBlock / 2; o > 0; o >>= 1) {
      if (t < o) {
        sh[0][t] += sh[0][t + o];
        sh[1][t] += sh[1][t + o];
      }
      __syncthreads();
    }
    if (t == 0) {
      s = sh[0][0];
      s2 = sh[1][0];
      const double mean = s / static_cast<double>(n);
      const double ex2 = s2 / static_cast<double>(n);
      const double ms = mean_scale ? static_cast<double>(mean_scale[c]) : 1.0;
      double var = ex2 - mean * mean * ms * (2.0 - ms);
      if (var < 0.0) var = 0.0;
      const double inv = 1.0 / sqrt(var + static_cast<double>(eps));
      stats[c] = static_cast<float>(mean * ms);
      stats[d + c] = static_cast<float>((weight ? static_cast<double>(weight[c]) : 1.0) * inv);
      stats[2 * d + c] = bias ? bias[c] : 0.f;
    }
    __syncthreads();
  }
}

template <int VEC>
__global__ __launch_bounds__(kBlock) void gn_apply_kernel(const float* __restrict__ x, int64_t n, int64_t d,
                                                          const float* __restrict__ stats, int relu,
                                                          const float* __restrict__ add_to, float* __restrict__ y) {
  const int64_t total = n * d / VEC;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;
  for (int64_t t = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; t < total; t += stride) {
    const int64_t i = t * VEC;
    const int64_t c = i % d;
    float v[VEC], o[VEC];
    if constexpr (VEC == 4) {
      const float4 q = *reinterpret_cast<const float4*>(x + i);
      v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
    } else {
      v[0] = x[i];
    }
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
      float r = (v[k] - stats[c + k]) * stats[d + c + k] + stats[2 * d + c + k];
      if (relu) r = r > 0.f ? r : 0.f;
      o[k] = r;
    }
    if (add_to) {
      if constexpr (VEC == 4) {
        const float4 q = *reinterpret_cast<const float4*>(add_to + i);
        o[0] += q.x; o[1] += q.y; o[2] += q.z; o[3] += q.w;
      } else {
        o[0] += add_to[i];
      }
    }
    if constexpr (VEC == 4)
      store_stream(y + i, make_float4(o[0], o[1], o[2], o[3]));
    else
      y[i] = o[0];
  }
}

// Row-tiled apply for d = 4 * 2^k <= 1024: a thread keeps ONE column chunk -- its 12 shift / scale / bias values live in
// registers for the whole launch -- and walks rows, four in flight: per float4 of x one load and one store, instead of
// a 64-bit modulo and twelve 4-byte stats loads.
constexpr int kApplyU = 4;
__global__ __launch_bounds__(kBlock) void gn_apply_rows4_kernel(const float* __restrict__ x, int64_t n, int64_t d, int dv,
                                                                const float* __restrict__ stats, int relu,
                                                                const float* __restrict__ add_to,
                                                                float* __restrict__ y) {
  const int cv = threadIdx.x % dv, rin = threadIdx.x / dv, rpb = kBlock / dv;
  const float4 sh = *reinterpret_cast<const float4*>(stats + 4 * cv);
  const float4 sc = *reinterpret_cast<const float4*>(stats + d + 4 * cv);
  const float4 bi = *reinterpret_cast<const float4*>(stats + 2 * d + 4 * cv);
  const int64_t step = static_cast<int64_t>(gridDim.x) * rpb * kApplyU;
  for (int64_t base = static_cast<int64_t>(blockIdx.x) * rpb * kApplyU; base < n; base += step) {
    float4 v[kApplyU], a[kApplyU];
#pragma unroll
    for (int u = 0; u < kApplyU; ++u) {
      const int64_t i = base + u * rpb + rin, ic = i < n ? i : n - 1;
      v[u] = *reinterpret_cast<const float4*>(x + ic * d + 4 * cv);
      if (add_to) a[u] = *reinterpret_cast<const float4*>(add_to + ic * d + 4 * cv);
    }
#pragma unroll
    for (int u = 0; u < kApplyU; ++u) {
      const int64_t i = base + u * rpb + rin;
      float4 o = make_float4((v[u].x - sh.x) * sc.x + bi.x, (v[u].y - sh.y) * sc.y + bi.y, (v[u].z - sh.z) * sc.z + bi.z,
                             (v[u].w - sh.w) * sc.w + bi.w);
      if (relu) o = make_float4(o.x > 0.f ? o.x : 0.f, o.y > 0.f ? o.y : 0.f, o.z > 0.f ? o.z : 0.f, o.w > 0.f ? o.w : 0.f);
      if (add_to) o = make_float4(o.x + a[u].x, o.y + a[u].y, o.z + a[u].z, o.w + a[u].w);
      if (i < n) store_stream(y + i * d + 4 * cv, o);
    }
  }
}

// ---- small n (<= 8192 rows): the whole op in ONE launch ------------------------------------------------
// GraphNorm statistics are per column, so column blocks are independent: a 1024-thread workgroup owns four
// columns (one float4 per row) of ALL rows, keeps its rows in registers (thread t: rows t, t+1024, ...), sums
// them in fp64 (per thread in row order, then a fixed shuffle / LDS tree), derives shift / scale / bias, and
// applies them to the registers.  The fb-pages-food forward calls GraphNorm four times on [620..6576, 32..256]
// tensors and is launch-bound: one launch instead of three saves ~9 us per call.
constexpr int kSmallRowsMax = 8;     // rows per thread, at most
constexpr int kSmallBlock = 1024;
// kSmallRows = rows per thread actually needed (1 for n <= 1024, ...): the loads of absent rows are clamped duplicates,
// and at fb-pages-food's 620 rows seven of eight were
template <int kSmallRows>
__global__ __launch_bounds__(kSmallBlock) void gn_small_kernel(const float* __restrict__ x, int n, int d,
                                                              const float* __restrict__ weight,
                                                              const float* __restrict__ bias,
                                                              const float* __restrict__ mean_scale, float eps, int relu,
                                                              const float* __restrict__ add_to, float* __restrict__ y,
                                                              const int64_t* __restrict__ idx, int n_table,
                                                              int32_t* err_flag) {
  __shared__ double red[2][kSmallBlock / 64][4];
  __shared__ float st[3][4];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int c = blockIdx.x * 4;
  float4 v[kSmallRows], a[kSmallRows];
  // parameter loads up front (after the reductions they would be a round trip nothing hides)
  const float p_w = (t < 4 && weight) ? weight[c + t] : 1.f, p_b = (t < 4 && bias) ? bias[c + t] : 0.f,
              p_ms = (t < 4 && mean_scale) ? mean_scale[c + t] : 1.f;
  int64_t src[kSmallRows];
#pragma unroll
  for (int k = 0; k < kSmallRows; ++k) {
    const int r = t + k * kSmallBlock;
    src[k] = r < n ? r : n - 1;
  }
  if (idx) {                                         // x is a table, row r of the input is x[idx[r]] (Embedding)
#pragma unroll
    for (int k = 0; k < kSmallRows; ++k) src[k] = idx[src[k]];     // every index load first
    bool bad = false;
#pragma unroll
    for (int k = 0; k < kSmallRows; ++k) {
      const bool b = src[k] < 0 || src[k] >= n_table;
      bad |= b;
      src[k] = b ? 0 : src[k];
    }
    if (bad) flag_error(err_flag, LPG_F_INDEX_RANGE);
  }
#pragma unroll
  for (int k = 0; k < kSmallRows; ++k)               // every load first (clamped rows), then the arithmetic
    v[k] = *reinterpret_cast<const float4*>(x + src[k] * d + c);
  asm volatile("" ::: "memory");
  double s[4] = {0.0, 0.0, 0.0, 0.0}, s2[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int k = 0; k < kSmallRows; ++k) {
    if (t + k * kSmallBlock < n) {
      const double q[4] = {v[k].x, v[k].y, v[k].z, v[k].w};
#pragma unroll
      for (int i = 0; i < 4; ++i) { s[i] += q[i]; s2[i] += q[i] * q[i]; }
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      s[i] += __shfl_xor(s[i], o);
      s2[i] += __shfl_xor(s2[i], o);
    }
  if (lane == 0)
#pragma unroll
    for (int i = 0; i < 4; ++i) { red[0][wave][i] = s[i]; red[1][wave][i] = s2[i]; }
#pragma unroll
  for (int k = 0; k < kSmallRows; ++k) {             // the residual rows fly during the barrier and the finalize
    const int r = t + k * kSmallBlock;
    const int64_t off = static_cast<int64_t>(r < n ? r : n - 1) * d + c;
    a[k] = add_to ? *reinterpret_cast<const float4*>(add_to + off) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  __syncthreads();
  if (t < 4) {
    double ts = 0.0, ts2 = 0.0;
    for (int w = 0; w < kSmallBlock / 64; ++w) { ts += red[0][w][t]; ts2 += red[1][w][t]; }   // fixed order
    const double mean = ts / static_cast<double>(n), ex2 = ts2 / static_cast<double>(n);
    const double ms = static_cast<double>(p_ms);
    double var = ex2 - mean * mean * ms * (2.0 - ms);
    if (var < 0.0) var = 0.0;
    const double inv = 1.0 / sqrt(var + static_cast<double>(eps));
    st[0][t] = static_cast<float>(mean * ms);
    st[1][t] = static_cast<float>(static_cast<double>(p_w) * inv);
    st[2][t] = p_b;
  }
  __syncthreads();
  const float sh[4] = {st[0][0], st[0][1], st[0][2], st[0][3]}, sc[4] = {st[1][0], st[1][1], st[1][2], st[1][3]},
              bi[4] = {st[2][0], st[2][1], st[2][2], st[2][3]};
#pragma unroll
  for (int k = 0; k < kSmallRows; ++k) {
    const int r = t + k * kSmallBlock;
    if (r < n) {
      const float q[4] = {v[k].x, v[k].y, v[k].z, v[k].w}, ad[4] = {a[k].x, a[k].y, a[k].z, a[k].w};
      float o[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float w = (q[i] - sh[i]) * sc[i] + bi[i];
        if (relu) w = w > 0.f ? w : 0.f;
        o[i] = w + ad[i];
      }
      *reinterpret_cast<float4*>(y + static_cast<int64_t>(r) * d + c) = make_float4(o[0], o[1], o[2], o[3]);
    }
  }
}

template <typename... A>
inline void launch_gn_small(dim3 grid, hipStream_t st, const float* x, int n, A... rest) {
  const int rows = (n + kSmallBlock - 1) / kSmallBlock;          // rows per thread actually present
  if (rows <= 1) hipLaunchKernelGGL(gn_small_kernel<1>, grid, dim3(kSmallBlock), 0, st, x, n, rest...);
  else if (rows <= 2) hipLaunchKernelGGL(gn_small_kernel<2>, grid, dim3(kSmallBlock), 0, st, x, n, rest...);
  else if (rows <= 4) hipLaunchKernelGGL(gn_small_kernel<4>, grid, dim3(kSmallBlock), 0, st, x, n, rest...);
  else hipLaunchKernelGGL(gn_small_kernel<8>, grid, dim3(kSmallBlock), 0, st, x, n, rest...);
}

inline int part_blocks(int64_t n, int64_t* rows_per_block) {
  int64_t rpb = ceil_div(n, kMaxPartBlocks);
  if (rpb < 8) rpb = 8;           // small n: many short blocks (each thread's row loop is a latency chain)
  *rows_per_block = rpb;
  return static_cast<int>(ceil_div(n, rpb));
}
}  // namespace

// ---- row-wise LayerNorm (the use_node_feat branch of LocalWLNet, TwoWL/model/model.py:28-33,47-51) ---------------
// torch.nn.LayerNorm(d, elementwise_affine=False): y = (x - mean) / sqrt(var + eps), biased variance over the d columns
// of each row.  One wave per row; the row stays in registers between the mean and the variance pass (d <= 64 * kLnMax).
namespace {
constexpr int kLnMax = 16;                           // columns per lane held in registers: d <= 1024
__global__ __launch_bounds__(kBlock) void layernorm_rows_kernel(const float* __restrict__ x, int64_t ldx, int64_t n, int d,
                                                                float eps, const float* __restrict__ weight,
                                                                const float* __restrict__ bias, int relu,
                                                                float* __restrict__ y, int64_t ldy) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x) >> 6;
  if (row >= n) return;                              // wave-uniform
  float v[kLnMax];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < kLnMax; ++i) {
    const int c = lane + 64 * i;
    v[i] = c < d ? x[row * ldx + c] : 0.f;
    s += v[i];
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o, 64);
  const float mean = s / static_cast<float>(d);
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < kLnMax; ++i) {
    const float dv = lane + 64 * i < d ? v[i] - mean : 0.f;
    q += dv * dv;
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) q += __shfl_xor(q, o, 64);
  const float rstd = 1.0f / sqrtf(q / static_cast<float>(d) + eps);
#pragma unroll
  for (int i = 0; i < kLnMax; ++i) {
    const int c = lane + 64 * i;
    if (c < d) {
      float o = (v[i] - mean) * rstd;
      if (weight) o *= weight[c];
      if (bias) o += bias[c];
      if (relu) o = o > 0.f ? o : 0.f;
      y[row * ldy + c] = o;
    }
  }
}
}  // namespace

LPG_EXPORT int lpg_layernorm_rows_f32(const float* x, int64_t ldx, int64_t n, int64_t d, float eps, const float* weight,
                                      const float* bias, int relu, float* y, int64_t ldy, void* stream) {
  LPG_REQUIRE(n >= 0 && d >= 1 && d <= 64 * kLnMax && ldx >= d && ldy >= d, "layernorm_rows: needs 1 <= d <= %d", 64 * kLnMax);
  if (n == 0) return LPG_OK;
  LPG_REQUIRE(x && y, "layernorm_rows: null pointer");
  hipLaunchKernelGGL(layernorm_rows_kernel, dim3(static_cast<unsigned>(ceil_div(n, kBlock / 64))), dim3(kBlock), 0,
                     as_stream(stream), x, ldx, n, static_cast<int>(d), eps, weight, bias, relu, y, ldy);
  return check_launch("layernorm_rows");
}

LPG_EXPORT int64_t lpg_graphnorm_bytes(int64_t n, int64_t d) {
  if (n < 0 || d < 1) return LPG_E_INVALID;
  return static_cast<int64_t>(kMaxPartBlocks) * 2 * d * 8 + lpg::align_up(3 * d * 4, 256) + 256;
}

LPG_EXPORT int lpg_graphnorm_f32(const float* x, int64_t n, int64_t d, const float* weight, const float* bias,
                                 const float* mean_scale, float eps, int relu, const float* add_to, float* y,
                                 void* workspace, int64_t workspace_bytes, void* stream) {
  LPG_REQUIRE(n >= 0 && d >= 1, "graphnorm: bad sizes");
  if (n == 0) return LPG_OK;
  LPG_REQUIRE(x && y && workspace, "graphnorm: null pointer");
  if (workspace_bytes < lpg_graphnorm_bytes(n, d)) {
    set_error("graphnorm: workspace too small");
    return LPG_E_WORKSPACE;
  }
  hipStream_t st = as_stream(stream);
  {
    auto al16 = [](const void* p) { return reinterpret_cast<uintptr_t>(p) % 16 == 0; };
    if (n <= kSmallRowsMax * kSmallBlock && d % 4 == 0 && d / 4 < 65536 && al16(x) && al16(y) && (!add_to || al16(add_to)) &&
        options().gn_small != 0) {   // option gn_small = 0 keeps the three-launch path
      launch_gn_small(dim3(static_cast<unsigned>(d / 4)), st, x, static_cast<int>(n), static_cast<int>(d), weight, bias,
                      mean_scale, eps, relu, add_to, y, static_cast<const int64_t*>(nullptr), 0,
                      static_cast<int32_t*>(nullptr));
      return check_launch("graphnorm (one launch)");
    }
  }
  double* part = static_cast<double*>(workspace);
  float* stats = reinterpret_cast<float*>(static_cast<char*>(workspace) + static_cast<int64_t>(kMaxPartBlocks) * 2 * d * 8);
  int64_t rpb;
  const int nb = part_blocks(n, &rpb);
  const bool wide_in = d % 4 == 0 && reinterpret_cast<uintptr_t>(x) % 16 == 0;
  const int64_t cols = wide_in ? d / 4 : d;          // threads needed per row
  int tpr = 1;
  while (tpr < cols && tpr < kBlock) tpr <<= 1;
  if (wide_in)
    hipLaunchKernelGGL(gn_partial_kernel<4>, dim3(nb), dim3(kBlock), 0, st, x, n, d, rpb, tpr, part);
  else
    hipLaunchKernelGGL(gn_partial_kernel<1>, dim3(nb), dim3(kBlock), 0, st, x, n, d, rpb, tpr, part);
  hipLaunchKernelGGL(gn_finalize_kernel, dim3(static_cast<unsigned>(d < 2048 ? d : 2048)), dim3(kBlock), 0, st, part, nb, n, d, weight, bias,
                     mean_scale, eps, stats);
  const bool wide = d % 4 == 0 && reinterpret_cast<uintptr_t>(x) % 16 == 0 && reinterpret_cast<uintptr_t>(y) % 16 == 0 &&
                    (!add_to || reinterpret_cast<uintptr_t>(add_to) % 16 == 0);
  if (wide && d / 4 <= kBlock && kBlock % (d / 4) == 0)
    hipLaunchKernelGGL(gn_apply_rows4_kernel, dim3(grid_for(ceil_div(n * (d / 4), kApplyU))), dim3(kBlock), 0, st, x, n, d,
                       static_cast<int>(d / 4), stats, relu, add_to, y);
  else if (wide)
    hipLaunchKernelGGL(gn_apply_kernel<4>, dim3(grid_for(n * d / 4)), dim3(kBlock), 0, st, x, n, d, stats, relu, add_to, y);
  else
    hipLaunchKernelGGL(gn_apply_kernel<1>, dim3(grid_for(n * d)), dim3(kBlock), 0, st, x, n, d, stats, relu, add_to, y);
  return check_launch("graphnorm");
}

// GraphNorm of a table whose row v stands for row_weight[v] identical rows of the real input: the statistics are the
// weighted column sums, n_total = sum of the weights; only the V distinct rows are normalised.  GraphNorm(Embedding(deg))
// (TwoWL/model/model.py:53-55,71) has as many distinct rows as there are distinct degrees, row_weight = their histogram.
LPG_EXPORT int lpg_graphnorm_weighted_f32(const float* x, int64_t n, int64_t d, const int64_t* row_weight, int64_t n_total,
                                          const float* weight, const float* bias, const float* mean_scale, float eps,
                                          int relu, float* y, void* workspace, int64_t workspace_bytes, void* stream) {
  LPG_REQUIRE(n >= 1 && d >= 1 && n_total >= 1, "graphnorm_weighted: bad sizes");
  LPG_REQUIRE(x && y && row_weight && workspace, "graphnorm_weighted: null pointer");
  if (workspace_bytes < lpg_graphnorm_bytes(n, d)) {
    set_error("graphnorm_weighted: workspace too small");
    return LPG_E_WORKSPACE;
  }
  hipStream_t st = as_stream(stream);
  double* part = static_cast<double*>(workspace);
  float* stats = reinterpret_cast<float*>(static_cast<char*>(workspace) + static_cast<int64_t>(kMaxPartBlocks) * 2 * d * 8);
  int64_t rpb;
  const int nb = part_blocks(n, &rpb);
  const bool wide = d % 4 == 0 && reinterpret_cast<uintptr_t>(x) % 16 == 0 && reinterpret_cast<uintptr_t>(y) % 16 == 0;
  const int64_t cols = wide ? d / 4 : d;
  int tpr = 1;
  while (tpr < cols && tpr < kBlock) tpr <<= 1;
  if (wide)
    hipLaunchKernelGGL(gn_partial_kernel<4>, dim3(nb), dim3(kBlock), 0, st, x, n, d, rpb, tpr, part, row_weight);
  else
    hipLaunchKernelGGL(gn_partial_kernel<1>, dim3(nb), dim3(kBlock), 0, st, x, n, d, rpb, tpr, part, row_weight);
  hipLaunchKernelGGL(gn_finalize_kernel, dim3(static_cast<unsigned>(d < 2048 ? d : 2048)), dim3(kBlock), 0, st, part, nb, n_total, d, weight, bias,
                     mean_scale, eps, stats);
  const float* none = nullptr;
  if (wide)
    hipLaunchKernelGGL(gn_apply_kernel<4>, dim3(grid_for(n * d / 4)), dim3(kBlock), 0, st, x, n, d, stats, relu, none, y);
  else
    hipLaunchKernelGGL(gn_apply_kernel<1>, dim3(grid_for(n * d)), dim3(kBlock), 0, st, x, n, d, stats, relu, none, y);
  return check_launch("graphnorm_weighted");
}

// GraphNorm(Embedding(idx)) in one launch for small n: the emb stage of LocalWLNet (TwoWL/model/model.py:53-55, :71).
LPG_EXPORT int lpg_embedding_graphnorm_f32(const float* table, int64_t n_table, int64_t d, const int64_t* idx, int64_t n,
                                           const float* weight, const float* bias, const float* mean_scale, float eps,
                                           int relu, float* y, int32_t* err_flag, void* stream) {
  LPG_REQUIRE(n >= 1 && n <= kSmallRowsMax * kSmallBlock && d >= 4 && d % 4 == 0 && d / 4 < 65536 && n_table >= 1,
              "embedding_graphnorm: needs 1 <= n <= %d rows and d %% 4 == 0", kSmallRowsMax * kSmallBlock);
  LPG_REQUIRE(table && idx && y && reinterpret_cast<uintptr_t>(table) % 16 == 0 && reinterpret_cast<uintptr_t>(y) % 16 == 0,
              "embedding_graphnorm: null or misaligned pointer");
  launch_gn_small(dim3(static_cast<unsigned>(d / 4)), as_stream(stream), table, static_cast<int>(n), static_cast<int>(d),
                  weight, bias, mean_scale, eps, relu, static_cast<const float*>(nullptr), y, idx,
                  static_cast<int>(n_table), err_flag);
  return check_launch("embedding_graphnorm");
}

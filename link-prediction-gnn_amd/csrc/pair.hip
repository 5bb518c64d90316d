// TwoWL gather-side ops: embedding lookup, pair Hadamard (K7), readout (K9), tuple-id flip (K8).
// Reference: TwoWL/model/model.py:71 (emb(x)), :75 (x[pos[:,0]] * x[pos[:,1]]),
// :78-83 (x[idx]; even/odd product; Linear(C2,1)), TwoWL/utils.py:71-78 (reverse).
#include "lpg_common.hpp"

namespace {
using namespace lpg;

template <int VEC>
__global__ __launch_bounds__(kBlock) void embedding_kernel(const float* __restrict__ table, int64_t n_rows, int64_t d,
                                                           const int64_t* __restrict__ idx, int64_t n,
                                                           float* __restrict__ out, int32_t* err_flag) {
  const int64_t dv = d / VEC, total = n * dv;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;
  for (int64_t t = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; t < total; t += stride) {
    const int64_t i = t / dv, c = (t - i * dv) * VEC;
    const int64_t r = idx[i];
    if (r < 0 || r >= n_rows) {
      flag_error(err_flag, LPG_F_INDEX_RANGE);
      continue;
    }
    if constexpr (VEC == 4)
      *reinterpret_cast<float4*>(out + i * d + c) = *reinterpret_cast<const float4*>(table + r * d + c);
    else
      out[i * d + c] = table[r * d + c];
  }
}

template <int VEC>
__global__ __launch_bounds__(kBlock) void pair_mul_kernel(const float* __restrict__ x, int64_t n, int64_t d,
                                                          const int64_t* __restrict__ pos, int64_t P,
                                                          float* __restrict__ h, int32_t* err_flag) {
  const int64_t dv = d / VEC, total = P * dv;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;
  for (int64_t t = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; t < total; t += stride) {
    const int64_t i = t / dv, c = (t - i * dv) * VEC;
    const int64_t a = pos[2 * i], b = pos[2 * i + 1];
    if (a < 0 || a >= n || b < 0 || b >= n) {
      flag_error(err_flag, LPG_F_INDEX_RANGE);
      continue;
    }
    if constexpr (VEC == 4) {
      const float4 u = *reinterpret_cast<const float4*>(x + a * d + c);
      const float4 v = *reinterpret_cast<const float4*>(x + b * d + c);
      *reinterpret_cast<float4*>(h + i * d + c) = make_float4(u.x * v.x, u.y * v.y, u.z * v.z, u.w * v.w);
    } else {
      h[i * d + c] = x[a * d + c] * x[b * d + c];
    }
  }
}

// Row-tiled forms for d = 4 * 2^k <= 1024 (every width the models use): a thread keeps ONE column chunk and walks rows,
// so the (row, column) split is computed once instead of a 64-bit division per element, and four rows are in flight
// per thread -- all index loads, then all row loads from clamped addresses, then the stores.
constexpr int kRowsU = 8;
__global__ __launch_bounds__(kBlock) void embedding_rows4_kernel(const float* __restrict__ table, int64_t n_rows,
                                                                 int64_t d, int dv, const int64_t* __restrict__ idx,
                                                                 int64_t n, const float* __restrict__ scale,
                                                                 float* __restrict__ out, int32_t* err_flag) {
  const int cv = threadIdx.x % dv, rin = threadIdx.x / dv, rpb = kBlock / dv;
  const int64_t step = static_cast<int64_t>(gridDim.x) * rpb * kRowsU;
  for (int64_t base = static_cast<int64_t>(blockIdx.x) * rpb * kRowsU; base < n; base += step) {
    int64_t r[kRowsU];
    float sc[kRowsU];
    bool ok[kRowsU];
#pragma unroll
    for (int u = 0; u < kRowsU; ++u) {
      const int64_t i = base + u * rpb + rin;
      r[u] = idx[i < n ? i : n - 1];
      sc[u] = scale ? scale[i < n ? i : n - 1] : 1.0f;     // scale is wave-uniformly given or not
    }
    float4 v[kRowsU];
#pragma unroll
    for (int u = 0; u < kRowsU; ++u) {
      const int64_t i = base + u * rpb + rin;
      const bool in = r[u] >= 0 && r[u] < n_rows;
      if (i < n && !in) flag_error(err_flag, LPG_F_INDEX_RANGE);
      ok[u] = i < n && in;
      v[u] = *reinterpret_cast<const float4*>(table + (in ? r[u] : 0) * d + 4 * cv);
    }
#pragma unroll
    for (int u = 0; u < kRowsU; ++u)
      if (ok[u])
        store_stream(out + (base + u * rpb + rin) * d + 4 * cv,
                     scale ? make_float4(v[u].x * sc[u], v[u].y * sc[u], v[u].z * sc[u], v[u].w * sc[u]) : v[u]);
  }
}

__global__ __launch_bounds__(kBlock) void pair_mul_rows4_kernel(const float* __restrict__ x, int64_t n, int64_t d, int dv,
                                                                const int64_t* __restrict__ pos, int64_t P,
                                                                float* __restrict__ h, int32_t* err_flag) {
  const int cv = threadIdx.x % dv, rin = threadIdx.x / dv, rpb = kBlock / dv;
  const int64_t step = static_cast<int64_t>(gridDim.x) * rpb * kRowsU;
  for (int64_t base = static_cast<int64_t>(blockIdx.x) * rpb * kRowsU; base < P; base += step) {
    int64_t a[kRowsU], b[kRowsU];
    bool ok[kRowsU];
#pragma unroll
    for (int u = 0; u < kRowsU; ++u) {
      const int64_t i = base + u * rpb + rin, ic = i < P ? i : P - 1;
      a[u] = pos[2 * ic];
      b[u] = pos[2 * ic + 1];
    }
    float4 va[kRowsU], vb[kRowsU];
#pragma unroll
    for (int u = 0; u < kRowsU; ++u) {
      const int64_t i = base + u * rpb + rin;
      const bool in = a[u] >= 0 && a[u] < n && b[u] >= 0 && b[u] < n;
      if (i < P && !in) flag_error(err_flag, LPG_F_INDEX_RANGE);
      ok[u] = i < P && in;
      va[u] = *reinterpret_cast<const float4*>(x + (in ? a[u] : 0) * d + 4 * cv);
      vb[u] = *reinterpret_cast<const float4*>(x + (in ? b[u] : 0) * d + 4 * cv);
    }
#pragma unroll
    for (int u = 0; u < kRowsU; ++u)
      if (ok[u])
        store_stream(h + (base + u * rpb + rin) * d + 4 * cv,
                     make_float4(va[u].x * vb[u].x, va[u].y * vb[u].y, va[u].z * vb[u].z, va[u].w * vb[u].w));
  }
}

inline bool rows4_ok(int64_t d) { return d % 4 == 0 && d / 4 <= kBlock && kBlock % (d / 4) == 0; }

// one wave per scored edge: lanes stride the channels, xor-shuffle tree (fixed order)
__global__ __launch_bounds__(kBlock) void pair_readout_kernel(const float* __restrict__ h, int64_t P, int64_t d,
                                                              const int64_t* __restrict__ idx, int64_t B,
                                                              const float* __restrict__ w,
                                                              const float* __restrict__ b0, float* __restrict__ out,
                                                              int32_t* err_flag) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x) >> 6;
  const int64_t nwaves = (static_cast<int64_t>(gridDim.x) * kBlock) >> 6;
  for (int64_t b = wave; b < B; b += nwaves) {
    const int64_t i0 = idx[2 * b], i1 = idx[2 * b + 1];
    if (i0 < 0 || i0 >= P || i1 < 0 || i1 >= P) {
      if (lane == 0) flag_error(err_flag, LPG_F_INDEX_RANGE);
      continue;
    }
    float s = 0.f;
    for (int64_t c = lane; c < d; c += 64) s = fmaf(h[i0 * d + c] * h[i1 * d + c], w[c], s);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) out[b] = s + (b0 ? b0[0] : 0.f);
  }
}

__global__ __launch_bounds__(kBlock) void flip_kernel(const int64_t* __restrict__ ei2, int64_t E2,
                                                      int64_t* __restrict__ edge, int64_t* __restrict__ edge_r) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;
  for (int64_t e = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; e < E2; e += stride) {
    const int64_t a = ei2[e], b = ei2[E2 + e];
    // flip(k) = k+1 for even k, k-1 for odd k  (1 - 2*(k > k//2*2) in the reference)
    edge[e] = a + 1 - 2 * (a & 1);
    edge[E2 + e] = b;
    edge_r[e] = a;
    edge_r[E2 + e] = b + 1 - 2 * (b & 1);
  }
}

inline bool al16(const void* p) { return reinterpret_cast<uintptr_t>(p) % 16 == 0; }
}  // namespace

LPG_EXPORT int lpg_embedding_scaled_f32(const float* table, int64_t n_rows, int64_t d, const int64_t* idx, int64_t n,
                                        const float* row_scale, float* out, int32_t* err_flag, void* stream) {
  LPG_REQUIRE(n >= 0 && d >= 1 && n_rows >= 1, "embedding_scaled: bad sizes");
  if (n == 0) return LPG_OK;
  LPG_REQUIRE(table && idx && out && row_scale, "embedding_scaled: null pointer");
  LPG_REQUIRE(rows4_ok(d) && al16(table) && al16(out), "embedding_scaled: needs 16-byte aligned rows of d %% 4 == 0, d / 4 a divisor of 256");
  hipLaunchKernelGGL(embedding_rows4_kernel, dim3(grid_for(ceil_div(n * (d / 4), kRowsU))), dim3(kBlock), 0, as_stream(stream),
                     table, n_rows, d, static_cast<int>(d / 4), idx, n, row_scale, out, err_flag);
  return check_launch("embedding_scaled");
}

LPG_EXPORT int lpg_embedding_f32(const float* table, int64_t n_rows, int64_t d, const int64_t* idx, int64_t n,
                                 float* out, int32_t* err_flag, void* stream) {
  LPG_REQUIRE(n >= 0 && d >= 1 && n_rows >= 0, "embedding: bad sizes");
  if (n == 0) return LPG_OK;
  LPG_REQUIRE(table && idx && out, "embedding: null pointer");
  if (rows4_ok(d) && al16(table) && al16(out) && n_rows > 0)
    hipLaunchKernelGGL(embedding_rows4_kernel, dim3(grid_for(ceil_div(n * (d / 4), kRowsU))), dim3(kBlock), 0,
                       as_stream(stream), table, n_rows, d, static_cast<int>(d / 4), idx, n, static_cast<const float*>(nullptr), out,
                       err_flag);
  else if (d % 4 == 0 && al16(table) && al16(out))
    hipLaunchKernelGGL(embedding_kernel<4>, dim3(grid_for(n * d / 4)), dim3(kBlock), 0, as_stream(stream), table,
                       n_rows, d, idx, n, out, err_flag);
  else
    hipLaunchKernelGGL(embedding_kernel<1>, dim3(grid_for(n * d)), dim3(kBlock), 0, as_stream(stream), table, n_rows,
                       d, idx, n, out, err_flag);
  return check_launch("embedding");
}

LPG_EXPORT int lpg_pair_gather_mul_f32(const float* x, int64_t n, int64_t d, const int64_t* pos, int64_t P, float* h,
                                       int32_t* err_flag, void* stream) {
  LPG_REQUIRE(n >= 0 && d >= 1 && P >= 0, "pair_gather_mul: bad sizes");
  if (P == 0) return LPG_OK;
  LPG_REQUIRE(x && pos && h, "pair_gather_mul: null pointer");
  if (rows4_ok(d) && al16(x) && al16(h) && n > 0)
    hipLaunchKernelGGL(pair_mul_rows4_kernel, dim3(grid_for(ceil_div(P * (d / 4), kRowsU))), dim3(kBlock), 0,
                       as_stream(stream), x, n, d, static_cast<int>(d / 4), pos, P, h, err_flag);
  else if (d % 4 == 0 && al16(x) && al16(h))
    hipLaunchKernelGGL(pair_mul_kernel<4>, dim3(grid_for(P * d / 4)), dim3(kBlock), 0, as_stream(stream), x, n, d, pos,
                       P, h, err_flag);
  else
    hipLaunchKernelGGL(pair_mul_kernel<1>, dim3(grid_for(P * d)), dim3(kBlock), 0, as_stream(stream), x, n, d, pos, P,
                       h, err_flag);
  return check_launch("pair_gather_mul");
}

LPG_EXPORT int lpg_pair_readout_f32(const float* h, int64_t P, int64_t d, const int64_t* idx, int64_t B,
                                    const float* w, const float* b0, float* out, int32_t* err_flag, void* stream) {
  LPG_REQUIRE(P >= 0 && d >= 1 && B >= 0, "pair_readout: bad sizes");
  if (B == 0) return LPG_OK;
  LPG_REQUIRE(h && idx && w && out, "pair_readout: null pointer");
  hipLaunchKernelGGL(pair_readout_kernel, dim3(grid_for(B * 64)), dim3(kBlock), 0, as_stream(stream), h, P, d, idx, B,
                     w, b0, out, err_flag);
  return check_launch("pair_readout");
}

LPG_EXPORT int lpg_flip_index_i64(const int64_t* ei2, int64_t E2, int64_t* edge, int64_t* edge_r, void* stream) {
  LPG_REQUIRE(E2 >= 0, "flip_index: bad size");
  if (E2 == 0) return LPG_OK;
  LPG_REQUIRE(ei2 && edge && edge_r, "flip_index: null pointer");
  hipLaunchKernelGGL(flip_kernel, dim3(grid_for(E2)), dim3(kBlock), 0, as_stream(stream), ei2, E2, edge, edge_r);
  return check_launch("flip_index");
}

// scatter_add kernels (K1/K2) + the library's error plumbing.
// Reference call site: TwoWL/utils.py:10 (degree = scatter_add(ones_like(ei[1]), ei[1], dim_size=N));
// torch_scatter.scatter_add semantics: out[index[e], :] += src[e, :].
#include <stdarg.h>

#include "lpg_common.hpp"

namespace lpg {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace lpg

LPG_EXPORT const char* lpg_last_error(void) { return lpg::g_err; }
LPG_EXPORT int lpg_abi_version(void) { return LPG_ABI_VERSION; }

namespace {
using namespace lpg;

// In-degree histogram for small N: one LDS histogram per workgroup (ds_add_u32), flushed with
// one 64-bit global atomic per non-zero bin.  Integer adds commute => bit-exact in any order.
constexpr int kLdsBins = 16384;  // 64 KiB of the CU's 160 KiB LDS

__global__ __launch_bounds__(kBlock) void degree_lds_kernel(const int64_t* __restrict__ index, int64_t E,
                                                            unsigned long long* __restrict__ out, int64_t N,
                                                            int32_t* err_flag) {
  extern __shared__ uint32_t bins[];
  for (int i = threadIdx.x; i < N; i += kBlock) bins[i] = 0;
  __syncthreads();
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;
  for (int64_t e = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; e < E; e += stride) {
    const int64_t j = index[e];
    if (j >= 0 && j < N)
      atomicAdd(&bins[j], 1u);
    else
      flag_error(err_flag, LPG_F_INDEX_RANGE);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < N; i += kBlock) {
    const uint32_t c = bins[i];
    if (c) atomicAdd(&out[i], static_cast<unsigned long long>(c));
  }
}

// In-degree histogram for large N: hot destinations (RMAT hubs collect ~0.2 % of all edges EACH)
// serialise global atomics on one address -- 13 ms for |E| = 100 M.  Each workgroup therefore keeps
// a small open-addressing table in LDS (key = destination, value = count): a key that wins a slot
// is counted in LDS for the rest of the workgroup's slice and flushed with ONE global atomic; a key
// that loses its slot to another key goes straight to a global atomic.  Hubs show up early and often,
// so they get slots; integer adds commute => bit-exact regardless of who wins.
constexpr int kHashSlots = 4096;   // 16 KiB keys + 32 KiB 64-bit sums

__global__ __launch_bounds__(kBlock) void degree_hash_kernel(const int64_t* __restrict__ src,
                                                             const int64_t* __restrict__ index, int64_t E,
                                                             unsigned long long* __restrict__ out, int64_t N,
                                                             int64_t per_block, int32_t* err_flag) {
  __shared__ int32_t keys[kHashSlots];
  __shared__ unsigned long long cnts[kHashSlots];
  for (int i = threadIdx.x; i < kHashSlots; i += kBlock) { keys[i] = -1; cnts[i] = 0ull; }
  __syncthreads();
  const int64_t b = static_cast<int64_t>(blockIdx.x) * per_block;
  int64_t e_end = b + per_block;
  if (e_end > E) e_end = E;
  for (int64_t e = b + threadIdx.x; e < e_end; e += kBlock) {
    const int64_t j64 = index[e];
    if (j64 < 0 || j64 >= N) { flag_error(err_flag, LPG_F_INDEX_RANGE); continue; }
    const int32_t j = static_cast<int32_t>(j64);
    const uint32_t slot = (static_cast<uint32_t>(j) * 2654435761u) >> 20;      // top 12 bits
    const unsigned long long v = src ? static_cast<unsigned long long>(src[e]) : 1ull;
    int32_t cur = keys[slot];
    if (cur == -1) cur = atomicCAS(&keys[slot], -1, j), cur = (cur == -1) ? j : cur;
    if (cur == j)
      atomicAdd(&cnts[slot], v);
    else
      atomicAdd(&out[j], v);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < kHashSlots; i += kBlock) {
    const unsigned long long c = cnts[i];
    if (c) atomicAdd(&out[keys[i]], c);
  }
}

// General integer scatter-add: lanes walk the flattened [E, d] source so one wave touches
// contiguous bytes of a destination row.
__global__ __launch_bounds__(kBlock) void scatter_add_i64_kernel(const int64_t* __restrict__ src,
                                                                 const int64_t* __restrict__ index, int64_t E,
                                                                 int64_t d, unsigned long long* __restrict__ out,
                                                                 int64_t N, int32_t* err_flag) {
  const int64_t total = E * d;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;
  for (int64_t t = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; t < total; t += stride) {
    const int64_t e = t / d, c = t - e * d;
    const int64_t j = index[e];
    if (j >= 0 && j < N) {
      const int64_t v = src ? src[t] : 1;
      atomicAdd(&out[j * d + c], static_cast<unsigned long long>(v));
    } else {
      flag_error(err_flag, LPG_F_INDEX_RANGE);
    }
  }
}

// Float scatter-add: global_atomic_add_f32, one dword per lane, a wave covers 256 contiguous
// bytes of a destination row whenever d >= 64 (the shape the memory-side atomic unit runs at
// full rate for; MI355X guide, "Global float atomics").
__global__ __launch_bounds__(kBlock) void scatter_add_f32_kernel(const float* __restrict__ src,
                                                                 const int64_t* __restrict__ index, int64_t E,
                                                                 int64_t d, float* __restrict__ out, int64_t N,
                                                                 int32_t* err_flag) {
  const int64_t total = E * d;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;
  for (int64_t t = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; t < total; t += stride) {
    const int64_t e = t / d, c = t - e * d;
    const int64_t j = index[e];
    if (j >= 0 && j < N)
      unsafeAtomicAdd(&out[j * d + c], src[t]);
    else
      flag_error(err_flag, LPG_F_INDEX_RANGE);
  }
}

// out[index[e], :] += src[e, :] for DISTINCT index entries (the rows one peer contributes to a shard in the sparse
// multi-GPU exchange): no two threads touch the same element, so plain 16-byte read-modify-writes replace the
// memory-side float atomics (which run at ~1.3 TB/s).
template <int VEC>
__global__ __launch_bounds__(kBlock) void rows_add_f32_kernel(const float* __restrict__ src,
                                                              const int64_t* __restrict__ index, int64_t E, int64_t d,
                                                              float* __restrict__ out, int64_t N, int32_t* err_flag) {
  const int64_t dv = d / VEC, total = E * dv;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;
  for (int64_t t = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; t < total; t += stride) {
    const int64_t e = t / dv, c = (t - e * dv) * VEC;
    const int64_t j = index[e];
    if (j < 0 || j >= N) {
      flag_error(err_flag, LPG_F_INDEX_RANGE);
      continue;
    }
    if constexpr (VEC == 4) {
      const float4 a = *reinterpret_cast<const float4*>(src + e * d + c);
      float4* o = reinterpret_cast<float4*>(out + j * d + c);
      const float4 b = *o;
      *o = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
    } else {
      out[j * d + c] += src[e * d + c];
    }
  }
}
// the same with bf16 rows on the source side (the partials of a bf16 exchange): summed into the fp32 shard, so the
// sum over the peers is rounded once at the end, not once per peer as a bf16 ring reduction does
template <int VEC>
__global__ __launch_bounds__(kBlock) void rows_add_bf16_kernel(const uint16_t* __restrict__ src,
                                                               const int64_t* __restrict__ index, int64_t E, int64_t d,
                                                               float* __restrict__ out, int64_t N, int32_t* err_flag) {
  const int64_t dv = d / VEC, total = E * dv;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;
  for (int64_t t = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; t < total; t += stride) {
    const int64_t e = t / dv, c = (t - e * dv) * VEC;
    const int64_t j = index[e];
    if (j < 0 || j >= N) {
      flag_error(err_flag, LPG_F_INDEX_RANGE);
      continue;
    }
    if constexpr (VEC == 8) {
      const uint4 a = *reinterpret_cast<const uint4*>(src + e * d + c);
      float4* o = reinterpret_cast<float4*>(out + j * d + c);
      const float4 b0 = o[0], b1 = o[1];
      o[0] = make_float4(bf16_to_f32(a.x & 0xFFFFu) + b0.x, bf16_to_f32(a.x >> 16) + b0.y, bf16_to_f32(a.y & 0xFFFFu) + b0.z,
                         bf16_to_f32(a.y >> 16) + b0.w);
      o[1] = make_float4(bf16_to_f32(a.z & 0xFFFFu) + b1.x, bf16_to_f32(a.z >> 16) + b1.y, bf16_to_f32(a.w & 0xFFFFu) + b1.z,
                         bf16_to_f32(a.w >> 16) + b1.w);
    } else {
      out[j * d + c] += bf16_to_f32(src[e * d + c]);
    }
  }
}
}  // namespace

LPG_EXPORT int lpg_rows_add_bf16(const uint16_t* src, const int64_t* index, int64_t E, int64_t d, float* out, int64_t N,
                                 int32_t* err_flag, void* stream) {
  LPG_REQUIRE(E >= 0 && N >= 0 && d >= 1, "rows_add_bf16: bad sizes");
  if (E == 0) return LPG_OK;
  LPG_REQUIRE(src && index && out, "rows_add_bf16: null pointer");
  const bool wide = d % 8 == 0 && reinterpret_cast<uintptr_t>(src) % 16 == 0 && reinterpret_cast<uintptr_t>(out) % 16 == 0;
  if (wide)
    hipLaunchKernelGGL(rows_add_bf16_kernel<8>, dim3(grid_for(E * d / 8)), dim3(kBlock), 0, as_stream(stream), src, index, E,
                       d, out, N, err_flag);
  else
    hipLaunchKernelGGL(rows_add_bf16_kernel<1>, dim3(grid_for(E * d)), dim3(kBlock), 0, as_stream(stream), src, index, E, d,
                       out, N, err_flag);
  return check_launch("rows_add_bf16");
}

LPG_EXPORT int lpg_rows_add_f32(const float* src, const int64_t* index, int64_t E, int64_t d, float* out, int64_t N,
                                int32_t* err_flag, void* stream) {
  LPG_REQUIRE(E >= 0 && N >= 0 && d >= 1, "rows_add_f32: bad sizes");
  if (E == 0) return LPG_OK;
  LPG_REQUIRE(src && index && out, "rows_add_f32: null pointer");
  const bool wide = d % 4 == 0 && reinterpret_cast<uintptr_t>(src) % 16 == 0 && reinterpret_cast<uintptr_t>(out) % 16 == 0;
  if (wide)
    hipLaunchKernelGGL(rows_add_f32_kernel<4>, dim3(grid_for(E * d / 4)), dim3(kBlock), 0, as_stream(stream), src, index, E,
                       d, out, N, err_flag);
  else
    hipLaunchKernelGGL(rows_add_f32_kernel<1>, dim3(grid_for(E * d)), dim3(kBlock), 0, as_stream(stream), src, index, E, d,
                       out, N, err_flag);
  return check_launch("rows_add_f32");
}

LPG_EXPORT int lpg_scatter_add_i64(const int64_t* src, const int64_t* index, int64_t E, int64_t d, int64_t* out,
                                   int64_t N, int32_t* err_flag, void* stream) {
  LPG_REQUIRE(E >= 0 && N >= 0 && d >= 1, "scatter_add_i64: bad sizes E=%lld N=%lld d=%lld", (long long)E,
              (long long)N, (long long)d);
  if (E == 0) return LPG_OK;
  LPG_REQUIRE(index && out, "scatter_add_i64: null pointer");
  hipStream_t st = as_stream(stream);
  auto* o = reinterpret_cast<unsigned long long*>(out);
  if (!src && d == 1 && N <= kLdsBins) {
    const int grid = grid_for(E, kBlock, 256 * 2);
    hipLaunchKernelGGL(degree_lds_kernel, dim3(grid), dim3(kBlock), static_cast<size_t>(N) * 4, st, index, E, o, N,
                       err_flag);
  } else if (d == 1 && N < (int64_t(1) << 31) && E >= (int64_t(1) << 16)) {
    const int grid = grid_for(E, kBlock, 256 * 8);
    const int64_t per_block = ceil_div(E, grid);
    hipLaunchKernelGGL(degree_hash_kernel, dim3(grid), dim3(kBlock), 0, st, src, index, E, o, N, per_block, err_flag);
  } else {
    hipLaunchKernelGGL(scatter_add_i64_kernel, dim3(grid_for(E * d)), dim3(kBlock), 0, st, src, index, E, d, o, N,
                       err_flag);
  }
  return check_launch("scatter_add_i64");
}

LPG_EXPORT int lpg_scatter_add_f32(const float* src, const int64_t* index, int64_t E, int64_t d, float* out,
                                   int64_t N, int32_t* err_flag, void* stream) {
  LPG_REQUIRE(E >= 0 && N >= 0 && d >= 1, "scatter_add_f32: bad sizes");
  if (E == 0) return LPG_OK;
  LPG_REQUIRE(src && index && out, "scatter_add_f32: null pointer");
  hipLaunchKernelGGL(scatter_add_f32_kernel, dim3(grid_for(E * d)), dim3(kBlock), 0, as_stream(stream), src, index,
                     E, d, out, N, err_flag);
  return check_launch("scatter_add_f32");
}

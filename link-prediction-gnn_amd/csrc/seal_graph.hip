// The train-positive graph of SEAL (`A` / `G` of SEAL/config/subgraph.py:27-31, :70-72) as a CSR with ascending, unique
// neighbours per row, self loops kept -- for SMALL graphs (<= 8192 nodes: the reference's fb-pages-food has 620), where
// the general route (doubled key list -> sort -> unique -> bincount -> cumsum: a dozen launches and a host sync for the
// unique's size) costs 0.2 ms of a 0.65 ms scoring call.  Here: an n x n bit matrix (50 KB at n = 620), three launches:
//   set bits (u, v) and (v, u)  ->  row popcounts + exclusive scan in one workgroup  ->  rows emit their set bits.
// Integer work, bit-exact against the sort-based route by construction (a set has no order to get wrong).
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_select.hpp>

#include "lpg_common.hpp"

namespace {
using namespace lpg;

constexpr int kMaxNodes = 8192;
constexpr int kScanBlock = 1024;

__global__ __launch_bounds__(kBlock) void tg_set_bits_kernel(const int64_t* __restrict__ pairs, int64_t n_pairs, int n,
                                                             int words, uint32_t* __restrict__ bits, int32_t* err_flag,
                                                             int directed) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < n_pairs; i += stride) {
    const int64_t u = pairs[2 * i], v = pairs[2 * i + 1];
    if (u < 0 || u >= n || v < 0 || v >= n) {
      flag_error(err_flag, LPG_F_INDEX_RANGE);
      continue;
    }
    atomicOr(&bits[u * words + (v >> 5)], 1u << (v & 31));
    if (!directed) atomicOr(&bits[v * words + (u >> 5)], 1u << (u & 31));     // (SEAL/config/subgraph.py:30-31)
  }
}

// one workgroup: thread t owns rows t, t + 1024, ... (<= 8); popcounts -> exclusive scan -> rowptr
__global__ __launch_bounds__(kScanBlock) void tg_rowptr_kernel(const uint32_t* __restrict__ bits, int n, int words,
                                                               int64_t* __restrict__ rowptr) {
  __shared__ int wave_tot[kScanBlock / 64];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  constexpr int RPT = kMaxNodes / kScanBlock;
  // rows are dealt in CONTIGUOUS runs (thread t: rows [t * rpt, (t + 1) * rpt)) so that the scan order is the row order
  const int rpt = (n + kScanBlock - 1) / kScanBlock;
  int cnt[RPT];
  int mine = 0;
#pragma unroll
  for (int k = 0; k < RPT; ++k) {
    const int r = t * rpt + k;
    int c = 0;
    if (k < rpt && r < n)
      for (int w = 0; w < words; ++w) c += __popc(bits[static_cast<int64_t>(r) * words + w]);
    cnt[k] = c;
    mine += c;
  }
  int inc = mine;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int up = __shfl_up(inc, o, 64);
    if (lane >= o) inc += up;
  }
  if (lane == 63) wave_tot[wave] = inc;
  __syncthreads();
  int base = 0;
  for (int w = 0; w < wave; ++w) base += wave_tot[w];
  int run = base + inc - mine;                       // exclusive prefix of this thread's first row
#pragma unroll
  for (int k = 0; k < RPT; ++k) {
    const int r = t * rpt + k;
    if (k < rpt && r < n) rowptr[r] = run;
    run += cnt[k];
  }
  if (t == kScanBlock - 1) rowptr[n] = run;          // the last thread's running total is the grand total
}

// a wave per row: lanes take words, positions by a prefix of the popcounts
__global__ __launch_bounds__(kBlock) void tg_fill_kernel(const uint32_t* __restrict__ bits, int n, int words,
                                                         const int64_t* __restrict__ rowptr, int32_t* __restrict__ col,
                                                         int64_t col_cap) {
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * kBlock + threadIdx.x) >> 6, nwaves = (gridDim.x * kBlock) >> 6;
  for (int r = wave; r < n; r += nwaves) {
    int64_t out = rowptr[r];
    for (int w0 = 0; w0 < words; w0 += 64) {
      const int w = w0 + lane;
      uint32_t m = w < words ? bits[static_cast<int64_t>(r) * words + w] : 0u;
      const int c = __popc(m);
      int inc = c;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const int up = __shfl_up(inc, o, 64);
        if (lane >= o) inc += up;
      }
      int64_t p = out + inc - c;
      while (m) {
        const int b = __ffs(m) - 1;
        m &= m - 1;
        if (p < col_cap) col[p] = w * 32 + b;
        ++p;
      }
      out += __shfl(inc, 63, 64);
    }
  }
}

// ---- any size: both directions of every pair as a 64-bit key row * n + col, sorted, deduplicated ---------------------
inline unsigned key_bits(int64_t n) {                       // bits of the largest key, n * n (the dropped sentinel)
  unsigned b = 1;
  const unsigned __int128 top = static_cast<unsigned __int128>(n) * static_cast<unsigned __int128>(n);
  while ((static_cast<unsigned __int128>(1) << b) <= top && b < 64) ++b;
  return b;
}

__global__ __launch_bounds__(kBlock) void tg_keys_kernel(const int64_t* __restrict__ pairs, int64_t P, int64_t n,
                                                         uint64_t* __restrict__ keys, int32_t* err_flag, int directed) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < P; i += stride) {
    const int64_t u = pairs[2 * i], v = pairs[2 * i + 1];
    const bool ok = u >= 0 && u < n && v >= 0 && v < n;
    if (!ok) flag_error(err_flag, LPG_F_INDEX_RANGE);
    const uint64_t drop = static_cast<uint64_t>(n) * static_cast<uint64_t>(n);
    keys[2 * i] = ok ? static_cast<uint64_t>(u) * n + v : drop;
    keys[2 * i + 1] = ok && !directed ? static_cast<uint64_t>(v) * n + u : drop;
  }
}

// unique sorted keys -> col and rowptr; thread k owns the rows in (row[k-1], row[k]] (as rowptr_from_sorted in
// csr_build.hip); the count comes from the device (rocprim::unique), the dropped sentinel sorts last and is cut off
__global__ __launch_bounds__(kBlock) void tg_emit_kernel(const uint64_t* __restrict__ uniq,
                                                         const int64_t* __restrict__ count, int64_t n,
                                                         int64_t* __restrict__ rowptr, int32_t* __restrict__ col,
                                                         int64_t col_cap) {
  int64_t M = *count;
  const uint64_t drop = static_cast<uint64_t>(n) * static_cast<uint64_t>(n);
  if (M > 0 && uniq[M - 1] == drop) --M;
  if (M > col_cap) M = col_cap;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;
  for (int64_t k = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; k <= M; k += stride) {
    const int64_t a = k == 0 ? -1 : static_cast<int64_t>(uniq[k - 1] / static_cast<uint64_t>(n));
    const int64_t b = k == M ? n : static_cast<int64_t>(uniq[k] / static_cast<uint64_t>(n));
    for (int64_t r = a + 1; r <= b; ++r) rowptr[r] = k;
    if (k < M) col[k] = static_cast<int32_t>(uniq[k] % static_cast<uint64_t>(n));
  }
}

struct TgWs {
  uint64_t* keys;
  uint64_t* sorted;
  uint64_t* uniq;
  int64_t* count;
  void* tmp;
  size_t tmp_bytes;
  int64_t total;
};

TgWs tg_carve(void* ws, int64_t P, int64_t n) {
  TgWs w;
  const int64_t M = 2 * (P > 0 ? P : 1);
  char* p = static_cast<char*>(ws);
  w.keys = reinterpret_cast<uint64_t*>(p);
  p += align_up(M * 8, 256);
  w.sorted = reinterpret_cast<uint64_t*>(p);
  p += align_up(M * 8, 256);
  w.uniq = reinterpret_cast<uint64_t*>(p);
  p += align_up(M * 8, 256);
  w.count = reinterpret_cast<int64_t*>(p);
  p += 256;
  w.tmp = p;
  size_t a = 0, b = 0;
  (void)rocprim::radix_sort_keys(nullptr, a, static_cast<uint64_t*>(nullptr), static_cast<uint64_t*>(nullptr),
                                 static_cast<size_t>(M), 0u, key_bits(n), hipStream_t(0), false);
  (void)rocprim::unique(nullptr, b, static_cast<uint64_t*>(nullptr), static_cast<uint64_t*>(nullptr),
                        static_cast<int64_t*>(nullptr), static_cast<size_t>(M), rocprim::equal_to<uint64_t>(),
                        hipStream_t(0), false);
  w.tmp_bytes = a > b ? a : b;
  w.total = (p - static_cast<char*>(ws)) + align_up(static_cast<int64_t>(w.tmp_bytes), 256);
  return w;
}
}  // namespace

LPG_EXPORT int64_t lpg_seal_train_csr_small_max_nodes(void) { return kMaxNodes; }

LPG_EXPORT int lpg_seal_train_csr_small(const int64_t* pairs, int64_t n_pairs, int64_t n_nodes, int directed, void* bits,
                                        int64_t bits_bytes, int64_t* rowptr, int32_t* col, int64_t col_cap,
                                        int32_t* err_flag, void* stream) {
  LPG_REQUIRE(n_pairs >= 0 && n_nodes >= 1 && n_nodes <= kMaxNodes && col_cap >= 0,
              "seal_train_csr_small: needs 1 <= n_nodes <= %d", kMaxNodes);
  LPG_REQUIRE(bits && rowptr && (col || col_cap == 0) && (pairs || n_pairs == 0), "seal_train_csr_small: null pointer");
  const int n = static_cast<int>(n_nodes), words = (n + 31) / 32;
  if (bits_bytes < static_cast<int64_t>(n) * words * 4) {
    set_error("seal_train_csr_small: bit matrix buffer too small");
    return LPG_E_WORKSPACE;
  }
  hipStream_t st = as_stream(stream);
  LPG_HIP(hipMemsetAsync(bits, 0, static_cast<size_t>(n) * words * 4, st));
  uint32_t* b = static_cast<uint32_t*>(bits);
  if (n_pairs > 0)
    hipLaunchKernelGGL(tg_set_bits_kernel, dim3(grid_for(n_pairs)), dim3(kBlock), 0, st, pairs, n_pairs, n, words, b, err_flag, directed);
  hipLaunchKernelGGL(tg_rowptr_kernel, dim3(1), dim3(kScanBlock), 0, st, b, n, words, rowptr);
  hipLaunchKernelGGL(tg_fill_kernel, dim3(grid_for(static_cast<int64_t>(n) * 64)), dim3(kBlock), 0, st, b, n, words, rowptr,
                     col, col_cap);
  return check_launch("seal_train_csr_small");
}

LPG_EXPORT int64_t lpg_seal_train_csr_bytes(int64_t n_pairs, int64_t n_nodes) {
  if (n_pairs < 0 || n_nodes < 1 || n_nodes >= (int64_t(1) << 31)) return LPG_E_INVALID;
  return tg_carve(nullptr, n_pairs, n_nodes).total;
}

LPG_EXPORT int lpg_seal_train_csr(const int64_t* pairs, int64_t n_pairs, int64_t n_nodes, int directed, int64_t* rowptr,
                                  int32_t* col, int64_t col_cap, void* workspace, int64_t workspace_bytes, int32_t* err_flag,
                                  void* stream) {
  LPG_REQUIRE(n_pairs >= 0 && n_pairs < (int64_t(1) << 30) && n_nodes >= 1 && n_nodes < (int64_t(1) << 31) && col_cap >= 0,
              "seal_train_csr: bad sizes");
  LPG_REQUIRE(rowptr && (col || col_cap == 0) && (pairs || n_pairs == 0), "seal_train_csr: null pointer");
  TgWs w = tg_carve(workspace, n_pairs, n_nodes);
  if (!workspace || workspace_bytes < w.total) {
    set_error("seal_train_csr: workspace %lld < %lld bytes", (long long)workspace_bytes, (long long)w.total);
    return LPG_E_WORKSPACE;
  }
  hipStream_t st = as_stream(stream);
  const int64_t M = 2 * n_pairs;
  LPG_HIP(hipMemsetAsync(w.count, 0, 8, st));
  if (M > 0) {
    hipLaunchKernelGGL(tg_keys_kernel, dim3(grid_for(n_pairs)), dim3(kBlock), 0, st, pairs, n_pairs, n_nodes, w.keys, err_flag, directed);
    size_t tmp = w.tmp_bytes;
    LPG_HIP(rocprim::radix_sort_keys(w.tmp, tmp, w.keys, w.sorted, static_cast<size_t>(M), 0u, key_bits(n_nodes), st, false));
    tmp = w.tmp_bytes;
    LPG_HIP(rocprim::unique(w.tmp, tmp, w.sorted, w.uniq, w.count, static_cast<size_t>(M), rocprim::equal_to<uint64_t>(),
                            st, false));
  }
  hipLaunchKernelGGL(tg_emit_kernel, dim3(grid_for(M + 1)), dim3(kBlock), 0, st, w.uniq, w.count, n_nodes, rowptr, col, col_cap);
  return check_launch("seal_train_csr");
}

// Once-per-graph structure work: COO -> dst-major CSR (with PyG's add_remaining_self_loops
// folded in), gcn_norm on the CSR, and the SpMM plan (row-id stream + empty-row list).
//
// Reference anchors: edge_index is [2,E] int64, row 0 = source, row 1 = destination
// (TwoWL/model/model.py:73,77 call GCNConv(x, edge_index); PyG flow source_to_target).
// The stable order inside a destination row (original edge order, appended self loop last) is
// the order PyG's scatter-add accumulates in on CPU.
//
// rocPRIM's radix sort is used for the one-time sort (library primitive, not on the timed path).
// Small lists sort unique 64-bit keys (row << 32 | edge id).  From 2^18 entries on the sort is a STABLE
// least-significant-digit radix sort of (row, payload) pairs over the ceil(log2 N) row bits only -- 3 passes of
// 8 bits at |V| = 5 M instead of 7 over the 55-bit unique key -- and stability keeps the edge order inside a row;
// without edge weights the payload is the source id itself, so the sorted payload IS `col` (no gather pass).
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>

#include "lpg_common.hpp"

namespace {
using namespace lpg;

struct CooWs {
  uint64_t* keys_in;
  uint64_t* keys_out;
  int32_t* loop_last;
  void* sort_tmp;
  size_t sort_tmp_bytes;
  int64_t total;
};

inline unsigned bits_for(int64_t n) {  // bits needed to represent values in [0, n]
  unsigned b = 1;
  while ((int64_t(1) << b) <= n) ++b;
  return b;
}

size_t sort_tmp_bytes(int64_t M, unsigned end_bit) {
  size_t bytes = 0;
  (void)rocprim::radix_sort_keys(nullptr, bytes, static_cast<uint64_t*>(nullptr), static_cast<uint64_t*>(nullptr),
                                 static_cast<size_t>(M), 0u, end_bit, hipStream_t(0), false);
  return bytes;
}

CooWs carve(void* ws, int64_t E, int64_t N, int loops) {
  CooWs w;
  const int64_t M = E + (loops ? N : 0);
  char* p = static_cast<char*>(ws);
  w.keys_in = reinterpret_cast<uint64_t*>(p);
  p += align_up(M * 8, 256);
  w.keys_out = reinterpret_cast<uint64_t*>(p);
  p += align_up(M * 8, 256);
  w.loop_last = reinterpret_cast<int32_t*>(p);
  p += align_up((N + 1) * 4, 256);
  w.sort_tmp = p;
  w.sort_tmp_bytes = sort_tmp_bytes(M > 0 ? M : 1, 32 + bits_for(N));
  w.total = (p - static_cast<char*>(ws)) + static_cast<int64_t>(align_up(w.sort_tmp_bytes, 256));
  return w;
}

__global__ __launch_bounds__(kBlock) void coo_keys_kernel(const int64_t* __restrict__ src,
                                                          const int64_t* __restrict__ dst,
                                                          const float* __restrict__ val, int64_t E, int64_t n_rows,
                                                          int64_t n_cols, int loops, int flip,
                                                          uint64_t* __restrict__ keys,
                                                          int32_t* __restrict__ loop_last, int32_t* err_flag) {
  const int64_t M = E + (loops ? n_rows : 0);
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;
  for (int64_t e = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; e < M; e += stride) {
    uint64_t row;
    if (e < E) {
      int64_t s = src ? src[e] : e, t = dst[e];     // src == nullptr: column = entry id
      if (flip & 1) s ^= 1;            // TwoWL reverse(): partner-direction tuple id (utils.py:71-78)
      if (flip & 2) t ^= 1;
      if (s < 0 || s >= n_cols || t < 0 || t >= n_rows) {
        flag_error(err_flag, LPG_F_INDEX_RANGE);
        row = static_cast<uint64_t>(n_rows);  // dropped bucket
      } else if (loops && s == t) {
        row = static_cast<uint64_t>(n_rows);  // existing self loop: replaced by the appended one
        if (val) atomicMax(&loop_last[s], static_cast<int32_t>(e));
      } else {
        row = static_cast<uint64_t>(t);
      }
    } else {
      row = static_cast<uint64_t>(e - E);  // appended self loop of node e-E
    }
    keys[e] = (row << 32) | static_cast<uint64_t>(e);
  }
}

constexpr int64_t kPairSortMin = int64_t(1) << 18;   // below: unique 64-bit keys (any sort algorithm gives the order)

struct CooWs32 {
  uint32_t* k_in;
  uint32_t* k_out;
  uint32_t* v_in;
  uint32_t* v_out;
  int32_t* loop_last;
  void* sort_tmp;
  size_t sort_tmp_bytes;
  int64_t total;
};

CooWs32 carve32(void* ws, int64_t E, int64_t N, int loops) {
  CooWs32 w;
  const int64_t M = E + (loops ? N : 0);
  char* p = static_cast<char*>(ws);
  uint32_t** slots[4] = {&w.k_in, &w.k_out, &w.v_in, &w.v_out};
  for (auto slot : slots) {
    *slot = reinterpret_cast<uint32_t*>(p);
    p += align_up(M * 4, 256);
  }
  w.loop_last = reinterpret_cast<int32_t*>(p);
  p += align_up((N + 1) * 4, 256);
  w.sort_tmp = p;
  size_t bytes = 0;
  (void)rocprim::radix_sort_pairs(nullptr, bytes, static_cast<uint32_t*>(nullptr), static_cast<uint32_t*>(nullptr),
                                  static_cast<uint32_t*>(nullptr), static_cast<uint32_t*>(nullptr),
                                  static_cast<size_t>(M > 0 ? M : 1), 0u, bits_for(N), hipStream_t(0), false);
  w.sort_tmp_bytes = bytes;
  w.total = (p - static_cast<char*>(ws)) + static_cast<int64_t>(align_up(bytes, 256));
  return w;
}

// the pair form of coo_keys_kernel: key = destination row (n_rows = dropped), payload = source id (by_col) or edge id
__global__ __launch_bounds__(kBlock) void coo_pairs_kernel(const int64_t* __restrict__ src,
                                                           const int64_t* __restrict__ dst,
                                                           const float* __restrict__ val, int64_t E, int64_t n_rows,
                                                           int64_t n_cols, int loops, int flip, int by_col,
                                                           uint32_t* __restrict__ keys, uint32_t* __restrict__ vals,
                                                           int32_t* __restrict__ loop_last, int32_t* err_flag) {
  const int64_t M = E + (loops ? n_rows : 0);
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;
  for (int64_t e = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; e < M; e += stride) {
    uint32_t row, pay = static_cast<uint32_t>(e);
    if (e < E) {
      int64_t s = src ? src[e] : e, t = dst[e];     // src == nullptr: column = entry id
      if (flip & 1) s ^= 1;
      if (flip & 2) t ^= 1;
      if (s < 0 || s >= n_cols || t < 0 || t >= n_rows) {
        flag_error(err_flag, LPG_F_INDEX_RANGE);
        row = static_cast<uint32_t>(n_rows);
        s = 0;
      } else if (loops && s == t) {
        row = static_cast<uint32_t>(n_rows);
        if (val) atomicMax(&loop_last[s], static_cast<int32_t>(e));
      } else {
        row = static_cast<uint32_t>(t);
      }
      if (by_col) pay = static_cast<uint32_t>(s);
    } else {
      row = static_cast<uint32_t>(e - E);
      if (by_col) pay = row;
    }
    keys[e] = row;
    vals[e] = pay;
  }
}

__global__ __launch_bounds__(kBlock) void rowptr_from_sorted32_kernel(const uint32_t* __restrict__ keys, int64_t M,
                                                                      int64_t n_rows, int64_t* __restrict__ rowptr) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;
  for (int64_t k = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; k <= M; k += stride) {
    const int64_t a = (k == 0) ? -1 : static_cast<int64_t>(keys[k - 1]);
    int64_t b = (k == M) ? n_rows : static_cast<int64_t>(keys[k]);
    if (b > n_rows) b = n_rows;
    for (int64_t r = a + 1; r <= b; ++r) rowptr[r] = k;
  }
}

// weighted / perm-keeping graphs: the sorted payload is the edge id
__global__ __launch_bounds__(kBlock) void coo_gather32_kernel(const uint32_t* __restrict__ ids, int64_t M, int64_t E,
                                                              const int64_t* __restrict__ src, int flip,
                                                              const float* __restrict__ val,
                                                              const int32_t* __restrict__ loop_last,
                                                              int32_t* __restrict__ col_out, float* __restrict__ val_out,
                                                              int32_t* __restrict__ perm_out) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;
  for (int64_t k = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; k < M; k += stride) {
    const int64_t e = static_cast<int64_t>(ids[k]);
    int32_t c;
    float w = 1.f;
    if (e < E) {
      {
        const int64_t sv = src ? src[e] : e;
        c = static_cast<int32_t>((flip & 1) ? (sv ^ 1) : sv);
      }
      if (val) w = val[e];
    } else {
      c = static_cast<int32_t>(e - E);
      if (val) {
        const int32_t last = loop_last[c];
        if (last >= 0) w = val[last];
      }
    }
    col_out[k] = c;
    if (val_out) val_out[k] = w;
    if (perm_out) perm_out[k] = static_cast<int32_t>(e);
  }
}

// rowptr[r] = first sorted position whose row >= r, for r in [0, n_rows]; thread k owns the
// rows in (row[k-1], row[k]].  No atomics, no scan.
__global__ __launch_bounds__(kBlock) void rowptr_from_sorted_kernel(const uint64_t* __restrict__ keys, int64_t M,
                                                                    int64_t n_rows, int64_t* __restrict__ rowptr) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;
  for (int64_t k = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; k <= M; k += stride) {
    const int64_t a = (k == 0) ? -1 : static_cast<int64_t>(keys[k - 1] >> 32);
    int64_t b = (k == M) ? n_rows : static_cast<int64_t>(keys[k] >> 32);
    if (b > n_rows) b = n_rows;
    for (int64_t r = a + 1; r <= b; ++r) rowptr[r] = k;
  }
}

__global__ __launch_bounds__(kBlock) void coo_gather_kernel(const uint64_t* __restrict__ keys, int64_t M, int64_t E,
                                                            const int64_t* __restrict__ src, int flip,
                                                            const float* __restrict__ val,
                                                            const int32_t* __restrict__ loop_last,
                                                            int32_t* __restrict__ col_out, float* __restrict__ val_out,
                                                            int32_t* __restrict__ perm_out) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;
  for (int64_t k = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; k < M; k += stride) {
    const int64_t e = static_cast<int64_t>(keys[k] & 0xffffffffull);
    int32_t c;
    float w = 1.f;
    if (e < E) {
      {
        const int64_t sv = src ? src[e] : e;
        c = static_cast<int32_t>((flip & 1) ? (sv ^ 1) : sv);
      }
      if (val) w = val[e];
    } else {
      c = static_cast<int32_t>(e - E);
      if (val) {
        const int32_t last = loop_last[c];
        if (last >= 0) w = val[last];
      }
    }
    col_out[k] = c;
    if (val_out) val_out[k] = w;
    if (perm_out) perm_out[k] = static_cast<int32_t>(e);
  }
}

// ---- gcn_norm: one wave per destination row; lanes stride the row, xor-shuffle tree reduce
// (fixed association => reproducible).
__global__ __launch_bounds__(kBlock) void row_weight_sum_kernel(const int64_t* __restrict__ rowptr,
                                                                const float* __restrict__ val, int64_t N,
                                                                float* __restrict__ dinv) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x) >> 6;
  const int64_t nwaves = (static_cast<int64_t>(gridDim.x) * kBlock) >> 6;
  for (int64_t r = wave; r < N; r += nwaves) {
    const int64_t b = rowptr[r], e = rowptr[r + 1];
    float s = 0.f;
    for (int64_t k = b + lane; k < e; k += 64) s += val ? val[k] : 1.f;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) {
      float di = 1.0f / sqrtf(s);      // deg^-1/2; deg == 0 -> inf -> 0 (masked_fill in gcn_norm)
      if (isinf(di)) di = 0.f;
      dinv[r] = di;
    }
  }
}

// unweighted graphs: deg[j] is just the row length (every stored weight is 1)
__global__ __launch_bounds__(kBlock) void row_len_dinv_kernel(const int64_t* __restrict__ rowptr, int64_t N,
                                                              float* __restrict__ dinv) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;
  for (int64_t r = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; r < N; r += stride) {
    const float deg = static_cast<float>(rowptr[r + 1] - rowptr[r]);
    float di = 1.0f / sqrtf(deg);
    if (isinf(di)) di = 0.f;
    dinv[r] = di;
  }
}

// edge-parallel: val[k] = dinv[col[k]] * w_k * dinv[row(k)], row ids from the SpMM plan
__global__ __launch_bounds__(kBlock) void gcn_scale_kernel(const int64_t* __restrict__ rowptr, int64_t N,
                                                           const int32_t* __restrict__ col,
                                                           const int32_t* __restrict__ rowid, float* __restrict__ val,
                                                           const float* __restrict__ dinv, int unit) {
  const int64_t E = rowptr[N];
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;
  for (int64_t k = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; k < E; k += stride) {
    const float w = unit ? 1.f : val[k];
    val[k] = (dinv[col[k]] * w) * dinv[rowid[k]];
  }
}

// ---- plan: per-edge row ids + list of empty rows + longest row (+ column range check)
// Two kernels.  plan_rows_kernel is row-parallel (a lane per row): validates rowptr, lists the empty rows (one
// atomic per wave reserves the slots) and records the longest row.  plan_edges_kernel is EDGE-balanced: a wave
// owns 1024 consecutive edges, finds the row of its first edge by one binary search in rowptr and then slides
// a window of 64 row starts over the rows its edges cover; the row of an edge is a 6-step search over the
// window (ds_bpermute), col reads and row-id writes are coalesced.  Balancing by rows does not work on RMAT:
// the 64 lowest ids hold ~3 % of all edges (26.8 ms for |E| = 100 M with a wave per row or per 64 rows).
__global__ __launch_bounds__(kBlock) void plan_rows_kernel(const int64_t* __restrict__ rowptr, int64_t N, int64_t e_cap,
                                                           int32_t* __restrict__ n_empty,
                                                           int32_t* __restrict__ empty_rows, int32_t* err_flag) {
  __shared__ int cnt[kBlock / 64];
  __shared__ int blk_base;
  const int lane = threadIdx.x & 63;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;
  int wave_max = 0;
  // (the loop bound is block-uniform: every thread of a workgroup reaches the barriers the same number of times)
  for (int64_t rb = static_cast<int64_t>(blockIdx.x) * kBlock; rb < N; rb += stride) {
    const int64_t r0 = rb + (threadIdx.x & ~63);
    const int64_t r = r0 + lane;
    const bool on = r < N;
    int64_t b = on ? rowptr[r] : 0, e = on ? rowptr[r + 1] : 0;
    if (on && (b < 0 || e < b || e > e_cap)) { flag_error(err_flag, LPG_F_INDEX_RANGE); b = e = 0; }
    const bool empty = on && e == b;
    const unsigned long long em = __ballot(empty);
    // slots for the empty rows: ONE atomic per workgroup (RMAT leaves 42 % of the rows empty, so every wave has
    // some; a returning atomic per wave on one address serialised 78 K waves: 4 of the 6 ms of this kernel)
    const int wv = threadIdx.x >> 6;
    if (lane == 0) cnt[wv] = __popcll(em);
    __syncthreads();
    if (threadIdx.x == 0) {
      int tot = 0;
#pragma unroll
      for (int i = 0; i < kBlock / 64; ++i) { const int c = cnt[i]; cnt[i] = tot; tot += c; }
      blk_base = tot ? atomicAdd(n_empty, tot) : 0;
    }
    __syncthreads();
    if (empty) empty_rows[blk_base + cnt[wv] + __popcll(em & ((1ull << lane) - 1ull))] = static_cast<int32_t>(r);
    int len = on ? static_cast<int>(e - b) : 0;       // the longest row: running maximum per wave ...
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) len = max(len, __shfl_xor(len, o, 64));
    wave_max = max(wave_max, len);
    __syncthreads();                                  // cnt / blk_base are reused by the next sweep
  }
  // ... folded into one of 32 header slots at the end (words 2..33; plan_edges_kernel reduces them into word 1).
  // One slot hammered by every wave at kernel start cost 0.1 ms of serialised atomics on a 1.5 M-row graph.
  if (lane == 0 && wave_max > 0) atomicMax(n_empty + 2 + (blockIdx.x & 31), wave_max);
}

constexpr int kPlanChunk = 1024;
__global__ __launch_bounds__(kBlock) void plan_edges_kernel(const int64_t* __restrict__ rowptr,
                                                            const int32_t* __restrict__ col, int64_t N, int64_t n_cols,
                                                            int64_t e_cap, int32_t* __restrict__ rowid,
                                                            int32_t* __restrict__ hdr, int32_t* err_flag) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x) >> 6;
  if (blockIdx.x == 0 && threadIdx.x == 0) {         // header word 1 = the longest row (32 slots of plan_rows_kernel)
    int m = 0;
    for (int i = 0; i < 32; ++i) m = max(m, hdr[2 + i]);
    hdr[1] = m;
  }
  int64_t E = rowptr[N];
  E = E < 0 ? 0 : (E > e_cap ? e_cap : E);
  int64_t k = wave * kPlanChunk;
  if (k >= E) return;
  const int64_t k_end = k + kPlanChunk < E ? k + kPlanChunk : E;
  int64_t lo = 0, hi = N;                            // last row whose start is <= k (wave-uniform search)
  while (hi - lo > 1) {
    const int64_t mid = (lo + hi) >> 1;
    if (rowptr[mid] <= k) lo = mid; else hi = mid;
  }
  int64_t r0 = lo;
  while (k < k_end && r0 < N) {
    const int64_t ri = r0 + lane;
    int64_t start = rowptr[ri < N ? ri : N];
    start = start < 0 ? 0 : (start > E ? E : start);            // a malformed rowptr was flagged by plan_rows_kernel
    const int64_t w_last = __shfl(start, 63, 64);
    if (w_last <= k) { r0 += 63; continue; }         // 63 rows without an edge at or after k (empty rows)
    const int64_t w_end = w_last < k_end ? w_last : k_end;
    for (int64_t kk = k; kk < w_end; kk += 64) {
      const int64_t ed = kk + lane;
      int i = 0;                                     // last i in [0, 62] with start[i] <= ed
#pragma unroll
      for (int step = 32; step > 0; step >>= 1) {
        const int j = i + step < 63 ? i + step : 62;
        const int64_t sv = __shfl(start, j, 64);
        if (i + step < 63 && sv <= ed) i += step;
      }
      if (ed < w_end) {
        rowid[ed] = static_cast<int32_t>(r0 + i);
        const int32_t c = col[ed];
        if (c < 0 || c >= n_cols) flag_error(err_flag, LPG_F_INDEX_RANGE);
      }
    }
    k = w_end;
    r0 += 63;
  }
}
}  // namespace

LPG_EXPORT int64_t lpg_coo_to_csr_bytes(int64_t E, int64_t N, int gcn_self_loops) {
  if (E < 0 || N < 0) return LPG_E_INVALID;
  const int64_t a = carve(nullptr, E, N, gcn_self_loops).total, b = carve32(nullptr, E, N, gcn_self_loops).total;
  return a > b ? a : b;
}

LPG_EXPORT int lpg_coo_to_csr(const int64_t* src, const int64_t* dst, const float* val, int64_t E, int64_t n_rows,
                              int64_t n_cols, int gcn_self_loops, int flip, int64_t* rowptr, int32_t* col_out, float* val_out,
                              int32_t* perm_out, void* workspace, int64_t workspace_bytes, int32_t* err_flag,
                              void* stream) {
  LPG_REQUIRE(E >= 0 && n_rows >= 0 && n_cols >= 0, "coo_to_csr: negative size");
  LPG_REQUIRE(!gcn_self_loops || n_rows == n_cols, "coo_to_csr: self loops need a square matrix");
  const int64_t M = E + (gcn_self_loops ? n_rows : 0);
  LPG_REQUIRE(M < (int64_t(1) << 31), "coo_to_csr: more than 2^31-1 edges");
  LPG_REQUIRE(rowptr && (M == 0 || col_out) && (E == 0 || dst), "coo_to_csr: null pointer");
  LPG_REQUIRE(!val || val_out, "coo_to_csr: val given without val_out");
  const bool pairs = M >= kPairSortMin;
  const int64_t need = pairs ? carve32(workspace, E, n_rows, gcn_self_loops).total
                             : carve(workspace, E, n_rows, gcn_self_loops).total;
  if (workspace_bytes < need || (!workspace && need > 0)) {
    set_error("coo_to_csr: workspace %lld < %lld bytes", (long long)workspace_bytes, (long long)need);
    return LPG_E_WORKSPACE;
  }
  hipStream_t st = as_stream(stream);
  if (pairs) {
    CooWs32 w = carve32(workspace, E, n_rows, gcn_self_loops);
    const int by_col = !val && !perm_out;
    if (gcn_self_loops && val) LPG_HIP(hipMemsetAsync(w.loop_last, 0xff, static_cast<size_t>(n_rows + 1) * 4, st));
    hipLaunchKernelGGL(coo_pairs_kernel, dim3(grid_for(M)), dim3(kBlock), 0, st, src, dst, val, E, n_rows, n_cols,
                       gcn_self_loops, flip, by_col, w.k_in, w.v_in, w.loop_last, err_flag);
    uint32_t* v_sorted = by_col ? reinterpret_cast<uint32_t*>(col_out) : w.v_out;
    size_t tmp = w.sort_tmp_bytes;
    LPG_HIP(rocprim::radix_sort_pairs(w.sort_tmp, tmp, w.k_in, w.k_out, w.v_in, v_sorted, static_cast<size_t>(M), 0u,
                                      bits_for(n_rows), st, false));
    hipLaunchKernelGGL(rowptr_from_sorted32_kernel, dim3(grid_for(M + 1)), dim3(kBlock), 0, st, w.k_out, M, n_rows,
                       rowptr);
    if (!by_col)
      hipLaunchKernelGGL(coo_gather32_kernel, dim3(grid_for(M)), dim3(kBlock), 0, st, w.v_out, M, E, src, flip, val,
                         w.loop_last, col_out, val_out, perm_out);
    return check_launch("coo_to_csr");
  }
  CooWs w = carve(workspace, E, n_rows, gcn_self_loops);
  if (gcn_self_loops && val) LPG_HIP(hipMemsetAsync(w.loop_last, 0xff, static_cast<size_t>(n_rows + 1) * 4, st));
  if (M > 0) {
    hipLaunchKernelGGL(coo_keys_kernel, dim3(grid_for(M)), dim3(kBlock), 0, st, src, dst, val, E, n_rows, n_cols,
                       gcn_self_loops, flip, w.keys_in, w.loop_last, err_flag);
    size_t tmp = w.sort_tmp_bytes;
    LPG_HIP(rocprim::radix_sort_keys(w.sort_tmp, tmp, w.keys_in, w.keys_out, static_cast<size_t>(M), 0u,
                                     32 + bits_for(n_rows), st, false));
  }
  hipLaunchKernelGGL(rowptr_from_sorted_kernel, dim3(grid_for(M + 1)), dim3(kBlock), 0, st, w.keys_out, M, n_rows,
                     rowptr);
  if (M > 0)
    hipLaunchKernelGGL(coo_gather_kernel, dim3(grid_for(M)), dim3(kBlock), 0, st, w.keys_out, M, E, src, flip, val,
                       w.loop_last, col_out, val_out, perm_out);
  return check_launch("coo_to_csr");
}

LPG_EXPORT int lpg_gcn_norm_csr(const void* plan, const int64_t* rowptr, const int32_t* col, float* val, float* dinv,
                                int64_t N, int64_t e_cap, int unit_weights, void* stream) {
  LPG_REQUIRE(N >= 0 && e_cap >= 0 && plan && rowptr && val && dinv && (e_cap == 0 || col), "gcn_norm_csr: bad argument");
  if (N == 0) return LPG_OK;
  hipStream_t st = as_stream(stream);
  const int32_t* rowid = reinterpret_cast<const int32_t*>(static_cast<const char*>(plan) + 256);
  if (unit_weights)
    hipLaunchKernelGGL(row_len_dinv_kernel, dim3(grid_for(N)), dim3(kBlock), 0, st, rowptr, N, dinv);
  else
    hipLaunchKernelGGL(row_weight_sum_kernel, dim3(grid_for(N * 64)), dim3(kBlock), 0, st, rowptr, val, N, dinv);
  if (e_cap > 0)
    hipLaunchKernelGGL(gcn_scale_kernel, dim3(grid_for(e_cap)), dim3(kBlock), 0, st, rowptr, N, col, rowid, val, dinv,
                       unit_weights);
  return check_launch("gcn_norm_csr");
}

LPG_EXPORT int64_t lpg_csr_plan_bytes(int64_t N, int64_t e_cap) {
  if (N < 0 || e_cap < 0) return LPG_E_INVALID;
  return 256 + align_up((e_cap + 1) * 4, 256) + align_up((N + 1) * 4, 256);
}

LPG_EXPORT int lpg_csr_plan_build(const int64_t* rowptr, const int32_t* col, int64_t N, int64_t n_cols,
                                  int64_t e_cap, void* plan, int64_t plan_bytes, int32_t* err_flag, void* stream) {
  LPG_REQUIRE(N >= 0 && e_cap >= 0 && rowptr && plan, "csr_plan_build: bad argument");
  LPG_REQUIRE(e_cap < (int64_t(1) << 31) && N < (int64_t(1) << 31), "csr_plan_build: sizes exceed int32");
  if (plan_bytes < lpg_csr_plan_bytes(N, e_cap)) {
    set_error("csr_plan_build: plan buffer too small");
    return LPG_E_WORKSPACE;
  }
  hipStream_t st = as_stream(stream);
  char* p = static_cast<char*>(plan);
  int32_t* n_empty = reinterpret_cast<int32_t*>(p);
  int32_t* rowid = reinterpret_cast<int32_t*>(p + 256);
  int32_t* empty_rows = reinterpret_cast<int32_t*>(p + 256 + align_up((e_cap + 1) * 4, 256));
  LPG_HIP(hipMemsetAsync(p, 0, 256, st));
  if (N > 0) {
    hipLaunchKernelGGL(plan_rows_kernel, dim3(grid_for(N)), dim3(kBlock), 0, st, rowptr, N, e_cap, n_empty, empty_rows,
                       err_flag);
    if (e_cap > 0) {
      const int64_t waves = ceil_div(e_cap, kPlanChunk);
      hipLaunchKernelGGL(plan_edges_kernel, dim3(static_cast<unsigned>(ceil_div(waves, kBlock / kWave))), dim3(kBlock), 0,
                         st, rowptr, col, N, n_cols, e_cap, rowid, n_empty, err_flag);
    }
  }
  return check_launch("csr_plan_build");
}

// CSR SpMM  Y = act(row_scale * (A @ X) + bias)  -- the hot kernel (K4 / K10).
//
// Reference anchors: the propagate step of GCNConv (TwoWL/model/model.py:73,77, [3P] PyG:
// out[dst] += w_e * h[src]) and SEAL's tanh(D^-1 (A~ (X W))) (SEAL/model/gnn.py:47-60).
//
// Design (gfx950, 64-lane waves, HBM-bound gather):
//  * The nnz array is cut into fixed chunks of kChunk edges.  A *group* of G lanes owns one chunk;
//    the G lanes span the feature row with one 16-byte load each (fp32: G = d/4, bf16: G = d/8),
//    so a wave gathers 64/G source rows per load instruction and every load is a full 16 B/lane.
//  * Column indices, edge weights and the per-edge row id are read coalesced (one per lane per
//    G edges) and broadcast inside the group with ds_bpermute; U = 8 independent row gathers are
//    issued back to back before the first accumulate, so each wave keeps 8 KiB in flight.
//  * Work is balanced by EDGES, not rows: a 180 K-edge hub row is just 700 chunks.  A segment
//    (run of edges of one row) that lies inside its chunk is reduced in registers and stored once;
//    a segment cut by a chunk border is written to a carry slot and finished by the fix-up kernel,
//    which adds the partials of that row in ascending chunk order.  No atomics anywhere, so the
//    result is bitwise reproducible run to run.
//  * Rows with no edges never appear in the edge stream; the plan lists them and a third tiny
//    kernel writes act(bias) there.
//  * The actual nnz is read from rowptr[N] on the device: launch geometry depends only on the
//    capacity, so no host sync is needed between graph build and SpMM.
#include <stdlib.h>

#include "lpg_common.hpp"

namespace {
using namespace lpg;

constexpr int kDefaultChunk = 512;  // edges per group chunk

// Kernel-selection options (lpg_common.hpp `Options`, csrc/options.hip): spmm_chunk = edges per chunk, spmm_lds = bytes of
// LDS reserved per workgroup (occupancy limiter; -1 = automatic), spmm_block = threads per workgroup, spmm_mode = 1
// index-prefetch variant; spmm_mode = 2 (gather-only upper bound, WRONG results) exists in -DLPG_TUNING builds only.
struct Tune {
  int chunk, mode, lds, block, off32;
  Tune() {
    const Options& o = options();
    chunk = static_cast<int>(o.spmm_chunk);
    if (chunk < 0 || chunk % 16 != 0) chunk = 0;
    mode = static_cast<int>(o.spmm_mode);
    lds = static_cast<int>(o.spmm_lds);
    off32 = static_cast<int>(o.spmm_off32);
    block = static_cast<int>(o.spmm_block);               // 0 = automatic (launch_g)
    if (block != 0 && block != 64 && block != 128 && block != 256) block = 0;
  }
};
inline Tune tune() { return Tune(); }

// Edges per chunk: 512 on big graphs (flat between 256 and 2048, sweep_notes.md); small graphs get
// smaller chunks (down to 16) so that a 70 K-edge tuple graph still spreads over thousands of
// groups instead of running 512 dependent gathers in a handful of waves (243 us -> ~10 us).
inline int chunk_for(int64_t e_cap) {
  const int forced = tune().chunk;
  if (forced) return forced;
  int c = 16;
  // enough chunks to put >= 16 waves on every SIMD even when a wave covers 64 chunks (d <= 4): the big
  // HBM-bound graphs end at 512 (|E| = 100 M: 8.55 ms vs 8.67 at 256, 8.89 at 128), a batch of SEAL
  // subgraphs (8.5 M edges, d = 32 and d = 1) at 64 (4-layer stack 2.35 ms at 512 -> 1.6 ms)
  while (c < kDefaultChunk && static_cast<int64_t>(c) * (int64_t(1) << 18) <= e_cap) c *= 2;
  // mid-size graphs (round 3, profiles/spmm_mid_graph.py: |V| = 236 K, 2.8 M edges, d = 32): at 16-edge chunks nearly every
  // chunk border cuts a row and the fix-up kernel costs as much as the main one -- 93 us at 16, 78 at 32, 70 at 64,
  // 62 at 128, 84 at 256; the 8.5 M-edge SEAL batch was already best at 64
  if (e_cap >= (int64_t(1) << 21) && c < 64) c = 64;
  else if (e_cap >= (int64_t(1) << 19) && c < 32) c = 32;
  return c;
}

// Per-phase timing of one SpMM call (option spmm_phase_timing = 1; bench.py's per-kernel split of the step): four
// library-owned events recorded on the launch stream around {main kernel | fix-up | empty rows}.  Results are not
// affected; the records cost a few microseconds, so the option is off outside the measuring loop.
// (thread-local like SideStream: two host threads timing their own calls do not share events)
thread_local hipEvent_t g_phase_ev[4];
thread_local bool g_phase_ev_ready = false;
thread_local bool g_phase_recorded = false;
inline void phase_mark(int i, hipStream_t st) {
  if (!options().spmm_phase_timing) return;
  if (!g_phase_ev_ready) {
    for (auto& e : g_phase_ev)
      if (hipEventCreate(&e) != hipSuccess) return;
    g_phase_ev_ready = true;
  }
  (void)hipEventRecord(g_phase_ev[i], st);
  if (i == 3) g_phase_recorded = true;
}

// A second stream for the empty-row fill of big graphs.  The main kernel runs one workgroup per CU (LDS reservation)
// and is bound by the row gather at ~5.6 TB/s of the 8 TB/s the memory can move; the fill is a pure write stream to
// rows the main and fix-up kernels never touch.  Forked before the main kernel and joined after the fix-up, about half
// of its 0.28 ms hides behind the gather (|E| = 100 M: 9.01 -> 8.86 ms, profiles/r02/spmm_overlap_probe.log).
// One stream + two events per host thread and device, created on first use; never inside a stream capture.
struct SideStream {
  hipStream_t s = nullptr;
  hipEvent_t fork = nullptr, join = nullptr;
  bool tried = false;
};
SideStream* side_stream(hipStream_t st) {
  constexpr int kMaxDev = 16;
  thread_local SideStream side[kMaxDev];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDev) return nullptr;
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(st, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) {
    (void)hipGetLastError();
    return nullptr;
  }
  SideStream& ss = side[dev];
  if (!ss.tried) {
    ss.tried = true;
    if (hipStreamCreateWithFlags(&ss.s, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&ss.fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ss.join, hipEventDisableTiming) != hipSuccess) {
      (void)hipGetLastError();
      ss.s = nullptr;
    }
  }
  return ss.s ? &ss : nullptr;
}

struct SpmmArgs {
  const int64_t* rowptr;
  const int32_t* col;
  const float* val;
  const int32_t* rowid;
  const int32_t* n_empty;
  const int32_t* empty_rows;
  const void* x;
  void* y;
  int64_t ldx, ldy, N, d, n_cols;
  const float* row_scale;
  const float* bias;
  int act;
  float* carry;   // [2 * n_chunks_cap][dpad]
  int64_t dpad;
  int chunk;      // edges per chunk
  int off32;      // x spans < 4 GiB: gather addresses are base + 32-bit byte offset (1 VALU instead of 6)
};

// ---- 16-byte (or scalar) row pieces ---------------------------------------------------------
// Raw = what one lane loads from a source row; it stays raw in registers while U gathers are in
// flight (bf16: 4 VGPRs per piece instead of 8) and is widened to fp32 at the accumulate.
template <typename T, int VEC>
struct Piece;
template <>
struct Piece<float, 4> {
  using Raw = float4;
  static __device__ __forceinline__ Raw load(const float* p) { return *reinterpret_cast<const float4*>(p); }
  static __device__ __forceinline__ void widen(const Raw& t, float (&v)[4]) { v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w; }
  // Y is written once and is far larger than the caches: a streaming store keeps its lines from displacing the X rows
  // the gathers re-use (|E| = 100 M, d = 128: 8.775 -> 8.721 ms on the same box, d = 64 and 256 the same 0.5 %)
  static __device__ __forceinline__ void store(float* p, const float (&v)[4]) {
    store_stream(p, make_float4(v[0], v[1], v[2], v[3]));
  }
};
template <>
struct Piece<float, 1> {
  using Raw = float;
  static __device__ __forceinline__ Raw load(const float* p) { return *p; }
  static __device__ __forceinline__ void widen(const Raw& t, float (&v)[1]) { v[0] = t; }
  static __device__ __forceinline__ void store(float* p, const float (&v)[1]) { *p = v[0]; }
};
template <>
struct Piece<uint16_t, 8> {
  using Raw = uint4;
  static __device__ __forceinline__ Raw load(const uint16_t* p) { return *reinterpret_cast<const uint4*>(p); }
  static __device__ __forceinline__ void widen(const Raw& t, float (&v)[8]) {
    const uint32_t w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      v[2 * i] = __uint_as_float(w[i] << 16);
      v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
    }
  }
  static __device__ __forceinline__ void store(uint16_t* p, const float (&v)[8]) {
    uint32_t w[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
      w[i] = static_cast<uint32_t>(f32_to_bf16(v[2 * i])) | (static_cast<uint32_t>(f32_to_bf16(v[2 * i + 1])) << 16);
    using u4 = __attribute__((ext_vector_type(4))) uint32_t;
    const u4 q = {w[0], w[1], w[2], w[3]};
    __builtin_nontemporal_store(q, reinterpret_cast<u4*>(p));
  }
};
template <>
struct Piece<uint16_t, 1> {
  using Raw = uint16_t;
  static __device__ __forceinline__ Raw load(const uint16_t* p) { return *p; }
  static __device__ __forceinline__ void widen(const Raw& t, float (&v)[1]) { v[0] = bf16_to_f32(t); }
  static __device__ __forceinline__ void store(uint16_t* p, const float (&v)[1]) { *p = f32_to_bf16(v[0]); }
};

template <int VEC>
__device__ __forceinline__ void carry_store(float* p, const float (&v)[VEC]) {
  if constexpr (VEC % 4 == 0) {
#pragma unroll
    for (int i = 0; i < VEC; i += 4) *reinterpret_cast<float4*>(p + i) = make_float4(v[i], v[i + 1], v[i + 2], v[i + 3]);
  } else {
#pragma unroll
    for (int i = 0; i < VEC; ++i) p[i] = v[i];
  }
}
template <int VEC>
__device__ __forceinline__ void carry_add(const float* p, float (&acc)[VEC]) {
  if constexpr (VEC % 4 == 0) {
#pragma unroll
    for (int i = 0; i < VEC; i += 4) {
      const float4 t = *reinterpret_cast<const float4*>(p + i);
      acc[i] += t.x; acc[i + 1] += t.y; acc[i + 2] += t.z; acc[i + 3] += t.w;
    }
  } else {
#pragma unroll
    for (int i = 0; i < VEC; ++i) acc[i] += p[i];
  }
}

// Main-kernel flush: bias lives in registers (loaded once per kernel) and the row scale arrives by
// shuffle from the per-batch prefetch, so a flush issues NO loads -- a load at a flush site makes
// the compiler wait vmcnt(0) there, draining the gathers in flight (+38 % on the GCN shape).
template <typename T, int VEC, int ACT>
__device__ __forceinline__ void store_out_regs(const SpmmArgs& a, int64_t row, int64_t c0, const float (&acc)[VEC],
                                               const float (&bias)[VEC], float rs) {
  float o[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) o[i] = apply_act(fmaf(acc[i], rs, bias[i]), ACT);   // ACT is a compile-time constant
  Piece<T, VEC>::store(static_cast<T*>(a.y) + row * a.ldy + c0, o);
}

template <typename T, int VEC, bool EPI = true>
__device__ __forceinline__ void store_out(const SpmmArgs& a, int64_t row, int64_t c0, const float (&acc)[VEC]) {
  if constexpr (!EPI) {      // plain Y = A @ X: no row scale, bias or activation
    Piece<T, VEC>::store(static_cast<T*>(a.y) + row * a.ldy + c0, acc);
  } else {
    float o[VEC];
    const float rs = a.row_scale ? a.row_scale[row] : 1.f;
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      float v = acc[i];
      if (a.row_scale) v *= rs;
      if (a.bias) v += a.bias[c0 + i];
      o[i] = apply_act(v, a.act);
    }
    Piece<T, VEC>::store(static_cast<T*>(a.y) + row * a.ldy + c0, o);
  }
}

// ---- main kernel -----------------------------------------------------------------------------
// EPIK: 0 = plain A @ X; 1/2/3 = epilogue (row scale, bias) with activation none / relu / tanh baked in,
// so a flush site carries only the code it needs (the flush code is replicated at 8 unrolled sites).
template <typename T, int VEC, int G, bool WEIGHTED, int EPIK, int UREQ, int MODE>
__global__ __launch_bounds__(kBlock) void spmm_main_kernel(const SpmmArgs a) {
  constexpr bool EPI = EPIK != 0;
  constexpr int GPW = kWave / G;            // groups (chunks) per wave
  constexpr int U = G < UREQ ? G : UREQ;    // row gathers in flight per group step
  const int kChunk = a.chunk;
  const int lane = threadIdx.x & 63;
  const int gl = lane & (G - 1);
  const int64_t wave = static_cast<int64_t>(blockIdx.x) * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int64_t E = a.rowptr[a.N];
  if (wave * GPW * kChunk >= E) return;     // wave-uniform
  const int64_t chunk = wave * GPW + lane / G;
  const int64_t start = chunk * kChunk;
  int64_t rem = E - start;
  const int n_e = rem <= 0 ? 0 : (rem > kChunk ? kChunk : static_cast<int>(rem));

  int64_t c0 = static_cast<int64_t>(blockIdx.y) * (G * VEC) + gl * VEC;
  const bool lane_on = c0 < a.d;
  if (!lane_on) c0 = 0;                     // keep the (ignored) loads in bounds
  const T* xb = static_cast<const T*>(a.x) + c0;

  bool open_left = false, open_right = false;
  if (n_e > 0) {
    open_left = start > 0 && a.rowid[start - 1] == a.rowid[start];
    const int64_t last = start + n_e - 1;
    open_right = last + 1 < E && a.rowid[last] == a.rowid[last + 1];
  }
  bool first_seg = true;
  float acc[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) acc[i] = 0.f;
  float* carry0 = a.carry + (2 * chunk) * a.dpad + c0;
  float bias_r[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) bias_r[i] = 0.f;
  if constexpr (EPI) {
    if (a.bias && lane_on) {
      if constexpr (VEC % 4 == 0) {               // c0 % 4 == 0 and torch allocations are 16-byte aligned
#pragma unroll
        for (int i = 0; i < VEC; i += 4) {
          const float4 b4 = *reinterpret_cast<const float4*>(a.bias + c0 + i);
          bias_r[i] = b4.x; bias_r[i + 1] = b4.y; bias_r[i + 2] = b4.z; bias_r[i + 3] = b4.w;
        }
      } else {
#pragma unroll
        for (int i = 0; i < VEC; ++i) bias_r[i] = a.bias[c0 + i];
      }
    }
    // pin the values here: hipcc would otherwise sink the loads into every flush site
#pragma unroll
    for (int i = 0; i < VEC; ++i) asm volatile("" : "+v"(bias_r[i]));
  }

  // REP: bit pattern selecting steps [0, U) of every group in the 64-bit ballot
  unsigned long long rep = 0;
#pragma unroll
  for (int gi = 0; gi < GPW; ++gi) rep |= ((U >= 64 ? ~0ull : ((1ull << U) - 1ull))) << (gi * G);
  const bool wave_full = __all(n_e == kChunk);      // only the last wave has a ragged tail
  const int gshift = lane & ~(G - 1);

  // index words (column, row id, next row id, weight) of one G-edge batch; with PREFETCH the
  // next batch's words are requested before the current batch's gathers, so the gather of batch
  // b+1 never waits on an index round trip (one wave per SIMD: nobody else hides that latency)
  constexpr bool PREFETCH = (MODE & 1) != 0;
  auto load_idx = [&](int base, int& c, int& r, int& rn, float& w, float& rs) {
    const bool in = base + gl < n_e;
    const int64_t e = start + base + gl;
    c = in ? a.col[e] : 0;
    r = in ? a.rowid[e] : -1;
    rn = (in && e + 1 < E) ? a.rowid[e + 1] : -2;
    w = 0.f;
    if constexpr (WEIGHTED) w = in ? a.val[e] : 0.f;
    rs = 1.f;
  };
  int c, r, rn, c_n = 0, r_n = -1, rn_n = -2;
  float w, w_n = 0.f, rs, rs_n = 1.f;
  load_idx(0, c, r, rn, w, rs);

#pragma unroll 1
  for (int base = 0; base < kChunk; base += G) {
    if (!__any(base < n_e)) break;
    if constexpr (PREFETCH) {
      if (base + G < kChunk) load_idx(base + G, c_n, r_n, rn_n, w_n, rs_n);
    } else if (base > 0) {
      load_idx(base, c, r, rn, w, rs);
    }
    const bool in = base + gl < n_e;
    const bool fl = in && r != rn;                   // a row ends at this lane's edge
    const unsigned long long fmask = __ballot(fl);   // wave-uniform: one bit per (group, step)
    const unsigned long long mym = fmask >> gshift;  // this group's bits at [0, G)

#pragma unroll 1
    for (int j0 = 0; j0 < G; j0 += U) {
      if (!__any(base + j0 < n_e)) break;            // chunk shorter than the group width
      typename Piece<T, VEC>::Raw raw[U];
      if (a.off32) {          // wave-uniform: with one wave per SIMD the address VALU is on the critical path
        const char* xbytes = static_cast<const char*>(a.x);
        const uint32_t row_bytes = static_cast<uint32_t>(a.ldx) * sizeof(T);
        const uint32_t lane_bytes = static_cast<uint32_t>(c0) * sizeof(T);
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const uint32_t cj = static_cast<uint32_t>(__shfl(c, j0 + u, G));
          raw[u] = Piece<T, VEC>::load(reinterpret_cast<const T*>(xbytes + (cj * row_bytes + lane_bytes)));
        }
      } else {
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int cj = __shfl(c, j0 + u, G);
          raw[u] = Piece<T, VEC>::load(xb + static_cast<int64_t>(cj) * a.ldx);
        }
      }
      const bool any_flush = (fmask & (rep << j0)) != 0ull;      // scalar test
      if ((MODE & 2) != 0 || (wave_full && !any_flush)) {
        // fast path (most sub-batches): no row ends here in any group -> straight accumulate
#pragma unroll
        for (int u = 0; u < U; ++u) {
          float v[VEC];
          Piece<T, VEC>::widen(raw[u], v);
          if constexpr (WEIGHTED) {
            const float wj = __shfl(w, j0 + u, G);
#pragma unroll
            for (int i = 0; i < VEC; ++i) acc[i] = fmaf(wj, v[i], acc[i]);
          } else {
#pragma unroll
            for (int i = 0; i < VEC; ++i) acc[i] += v[i];
          }
        }
      } else {
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const bool ok = base + j0 + u < n_e;
          float v[VEC];
          Piece<T, VEC>::widen(raw[u], v);
          if constexpr (WEIGHTED) {
            const float wj = __shfl(w, j0 + u, G);   // 0 for edges past the chunk end
#pragma unroll
            for (int i = 0; i < VEC; ++i) acc[i] = fmaf(wj, ok ? v[i] : 0.f, acc[i]);
          } else {
#pragma unroll
            for (int i = 0; i < VEC; ++i) acc[i] += ok ? v[i] : 0.f;
          }
          if ((mym >> (j0 + u)) & 1ull) {            // a row ends at this edge (group-uniform)
            const int rj = __shfl(r, j0 + u, G);
            float rsj = 1.f;                         // the row scale is read at the (rare) flush itself; callers that
            if constexpr (EPI)                       // can fold it into the edge weights instead (SEAL's D^-1) do so
              if (a.row_scale) rsj = a.row_scale[rj];
            if (lane_on) {
              if (first_seg && open_left)
                carry_store<VEC>(carry0, acc);        // last piece of a row begun in an earlier chunk
              else if constexpr (EPI)
                store_out_regs<T, VEC, EPIK - 1>(a, rj, c0, acc, bias_r, rsj);
              else
                store_out<T, VEC, false>(a, rj, c0, acc);
            }
#pragma unroll
            for (int i = 0; i < VEC; ++i) acc[i] = 0.f;
            first_seg = false;
          }
        }
      }
    }
    if constexpr (PREFETCH) { c = c_n; r = r_n; rn = rn_n; w = w_n; rs = rs_n; }
  }
  if (open_right && lane_on) carry_store<VEC>(carry0 + ((first_seg && open_left) ? 0 : a.dpad), acc);
}

// ---- short rows, known on the host: one group of G lanes per ROW ---------------------------------------
// When the plan's longest row is short (max_row_hint <= 32) a row never needs to be cut: a group walks its row left to right, U gathers
// in flight, stores once with the epilogue -- no carry slots, no fix-up launch, and empty rows get act(bias)
// from the same kernel.  One launch instead of two or three: the fb-pages-food forward is launch-bound.
// Sum order: CSR order inside the row == oracle/spmm_ref.c with a chunk at least as long as the row.
template <typename T, int VEC, int G, bool WEIGHTED>
__global__ __launch_bounds__(kBlock) void spmm_rows_kernel(const SpmmArgs a) {
  constexpr int GPW = kWave / G;
  constexpr int U = G < 8 ? G : 8;
  const int lane = threadIdx.x & 63;
  const int gl = lane & (G - 1);
  const int64_t wave = static_cast<int64_t>(blockIdx.x) * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int64_t row = wave * GPW + lane / G;
  const bool row_on = row < a.N;
  int64_t c0 = static_cast<int64_t>(blockIdx.y) * (G * VEC) + gl * VEC;
  const bool lane_on = row_on && c0 < a.d;
  if (c0 >= a.d) c0 = 0;
  const T* xb = static_cast<const T*>(a.x) + c0;
  const int64_t beg = row_on ? a.rowptr[row] : 0;
  const int64_t end = row_on ? a.rowptr[row + 1] : 0;
  float acc[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) acc[i] = 0.f;
  for (int64_t e0 = beg; __any(e0 < end); e0 += G) {
    const int64_t e = e0 + gl;
    const bool in = e < end;
    const int c = in ? a.col[e] : 0;
    float w = 0.f;
    if constexpr (WEIGHTED) w = in ? a.val[e] : 0.f;
#pragma unroll 1
    for (int j0 = 0; j0 < G; j0 += U) {
      if (!__any(e0 + j0 < end)) break;
      typename Piece<T, VEC>::Raw raw[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int cj = __shfl(c, j0 + u, G);
        raw[u] = Piece<T, VEC>::load(xb + static_cast<int64_t>(cj) * a.ldx);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const bool ok = e0 + j0 + u < end;
        float v[VEC];
        Piece<T, VEC>::widen(raw[u], v);
        if constexpr (WEIGHTED) {
          const float wj = __shfl(w, j0 + u, G);       // 0 past the row's end
#pragma unroll
          for (int i = 0; i < VEC; ++i) acc[i] = fmaf(wj, ok ? v[i] : 0.f, acc[i]);
        } else {
#pragma unroll
          for (int i = 0; i < VEC; ++i) acc[i] += ok ? v[i] : 0.f;
        }
      }
    }
  }
  if (lane_on) {
    if (a.row_scale || a.bias || a.act != LPG_ACT_NONE) store_out<T, VEC, true>(a, row, c0, acc);
    else store_out<T, VEC, false>(a, row, c0, acc);
  }
}

template <typename T, int VEC, int G>
int launch_rows_g(const SpmmArgs& a, hipStream_t st) {
  constexpr int GPW = kWave / G;
  const int64_t n_blocks = ceil_div(ceil_div(a.N, GPW), kBlock / kWave);
  const int64_t tiles = ceil_div(a.d, G * VEC);
  LPG_REQUIRE(n_blocks < (int64_t(1) << 31) && tiles < 65536, "csr_spmm: grid too large");
  const dim3 grid(static_cast<unsigned>(n_blocks), static_cast<unsigned>(tiles));
  if (a.val) hipLaunchKernelGGL((spmm_rows_kernel<T, VEC, G, true>), grid, dim3(kBlock), 0, st, a);
  else hipLaunchKernelGGL((spmm_rows_kernel<T, VEC, G, false>), grid, dim3(kBlock), 0, st, a);
  return LPG_OK;
}
template <typename T, int VEC>
int launch_rows(const SpmmArgs& a, hipStream_t st) {
  const int64_t lanes = ceil_div(a.d, VEC);
  if (lanes <= 2) return launch_rows_g<T, VEC, 2>(a, st);
  if (lanes <= 4) return launch_rows_g<T, VEC, 4>(a, st);
  if (lanes <= 8) return launch_rows_g<T, VEC, 8>(a, st);
  if (lanes <= 16) return launch_rows_g<T, VEC, 16>(a, st);
  if (lanes <= 32) return launch_rows_g<T, VEC, 32>(a, st);
  return launch_rows_g<T, VEC, 64>(a, st);
}

// ---- narrow features (d = 1, 2, 4, 8, 16): lanes are EDGES, not feature columns ---------------------------
// With d <= 4 the main kernel's "G lanes per chunk" degenerates to one lane walking a chunk alone (64 chunks
// per wave, every index load 64 cache lines wide, a row flush on almost every step): 0.5 ms for the 8.5 M-edge
// SEAL batch at d = 1.  Here a wave owns one chunk and takes 64 consecutive edges per step -- col / rowid / val
// loads are fully coalesced -- multiplies, and reduces rows with a segmented inclusive scan across the lanes
// (rows are contiguous in CSR order, so "same row as the lane o places back" is the segment test).  The lane
// holding a row's last edge stores it; a row still open at the end of a 64-edge step continues in `run`.
// Chunk borders use the same carry slots and the same fix-up kernel as the main kernel.  The order of the
// additions inside a row is a fixed tree, so results are bitwise reproducible, but it is NOT the left-to-right
// order of oracle/spmm_ref.c (the bit-exact-vs-C test covers the main kernel; this path is tolerance-tested).
template <typename T> __device__ __forceinline__ float elt_to_f32(T v);
template <> __device__ __forceinline__ float elt_to_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ float elt_to_f32<uint16_t>(uint16_t v) { return bf16_to_f32(v); }
template <typename T> __device__ __forceinline__ T f32_to_elt(float v);
template <> __device__ __forceinline__ float f32_to_elt<float>(float v) { return v; }
template <> __device__ __forceinline__ uint16_t f32_to_elt<uint16_t>(float v) { return f32_to_bf16(v); }

// DV consecutive elements of a row; VIO: 16-byte pieces (the caller checked alignment and DV % pieces == 0)
template <typename T, int DV, bool VIO>
__device__ __forceinline__ void load_elts(const T* p, float (&v)[DV]) {
  if constexpr (VIO) {
    constexpr int PV = 16 / sizeof(T);
#pragma unroll
    for (int i = 0; i < DV; i += PV) {
      float t[PV];
      Piece<T, PV>::widen(Piece<T, PV>::load(p + i), t);
#pragma unroll
      for (int k = 0; k < PV; ++k) v[i + k] = t[k];
    }
  } else {
#pragma unroll
    for (int i = 0; i < DV; ++i) v[i] = elt_to_f32<T>(p[i]);
  }
}
template <typename T, int DV, bool VIO>
__device__ __forceinline__ void store_elts(T* p, const float (&v)[DV]) {
  if constexpr (VIO) {
    constexpr int PV = 16 / sizeof(T);
#pragma unroll
    for (int i = 0; i < DV; i += PV) {
      float t[PV];
#pragma unroll
      for (int k = 0; k < PV; ++k) t[k] = v[i + k];
      Piece<T, PV>::store(p + i, t);
    }
  } else {
#pragma unroll
    for (int i = 0; i < DV; ++i) p[i] = f32_to_elt<T>(v[i]);
  }
}

template <typename T, int DV, bool WEIGHTED, bool VIO>
__global__ __launch_bounds__(kBlock) void spmm_edge_kernel(const SpmmArgs a) {
  const int lane = threadIdx.x & 63;
  const int64_t chunk = static_cast<int64_t>(blockIdx.x) * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int kChunk = a.chunk;
  const int64_t E = a.rowptr[a.N];
  const int64_t start = chunk * kChunk;
  if (start >= E) return;                           // wave-uniform
  const int n_e = E - start > kChunk ? kChunk : static_cast<int>(E - start);
  const int r_start = a.rowid[start];
  const bool open_left = start > 0 && a.rowid[start - 1] == r_start;
  const int64_t last = start + n_e - 1;
  const bool open_right = last + 1 < E && a.rowid[last] == a.rowid[last + 1];
  float* carry0 = a.carry + (2 * chunk) * a.dpad;
  const T* xb = static_cast<const T*>(a.x);
  T* yb = static_cast<T*>(a.y);
  float run[DV];                                    // wave-uniform: partial sum of the row open at the step start
#pragma unroll
  for (int i = 0; i < DV; ++i) run[i] = 0.f;
  int run_row = -1;

  for (int base = 0; base < n_e; base += 64) {
    const bool in = base + lane < n_e;
    const int64_t e = start + base + lane;
    const int c = in ? a.col[e] : 0;
    const int r = in ? a.rowid[e] : -1;
    const int rn = (in && e + 1 < E) ? a.rowid[e + 1] : -2;
    float w = 1.f;
    if constexpr (WEIGHTED) w = in ? a.val[e] : 0.f;
    float v[DV];
    load_elts<T, DV, VIO>(xb + static_cast<int64_t>(c) * a.ldx, v);
#pragma unroll
    for (int i = 0; i < DV; ++i) v[i] = in ? (WEIGHTED ? w * v[i] : v[i]) : 0.f;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {              // segmented inclusive scan, lane order = edge order
      const int rr = __shfl_up(r, o, 64);
      const bool same = lane >= o && rr == r;
#pragma unroll
      for (int i = 0; i < DV; ++i) {
        const float t = __shfl_up(v[i], o, 64);
        if (same) v[i] += t;
      }
    }
    if (in && r != rn) {                            // this lane holds the row's last edge
      float tot[DV];
#pragma unroll
      for (int i = 0; i < DV; ++i) tot[i] = r == run_row ? run[i] + v[i] : v[i];
      if (open_left && r == r_start) {              // tail of a row begun in an earlier chunk: the fix-up finishes it
#pragma unroll
        for (int i = 0; i < DV; ++i) carry0[i] = tot[i];
      } else {
        const float rs = a.row_scale ? a.row_scale[r] : 1.f;
        float o[DV];
#pragma unroll
        for (int i = 0; i < DV; ++i) {
          o[i] = tot[i] * rs;
          if (a.bias) o[i] += a.bias[i];
          o[i] = apply_act(o[i], a.act);
        }
        store_elts<T, DV, VIO>(yb + static_cast<int64_t>(r) * a.ldy, o);
      }
    }
    // the row open at the end of this step (if any) continues
    const int L = n_e - base > 64 ? 63 : n_e - base - 1;
    const int rL = __shfl(r, L, 64), rnL = __shfl(rn, L, 64);
    if (rL != rnL) {
      run_row = -1;
    } else {
#pragma unroll
      for (int i = 0; i < DV; ++i) {
        const float vL = __shfl(v[i], L, 64);
        run[i] = rL == run_row ? run[i] + vL : vL;
      }
      run_row = rL;
    }
  }
  if (open_right && lane == 0) {
    float* slot = carry0 + ((open_left && run_row == r_start) ? 0 : a.dpad);   // one row through the whole chunk: slot 0
#pragma unroll
    for (int i = 0; i < DV; ++i) slot[i] = run[i];
  }
}

constexpr int kFixupLongRow = 64;   // pieces after the first beyond which a row's fix-up is shared by the wave's groups

// ---- fix-up: finish the rows that straddle chunk borders --------------------------------------
// The group that owns a row's LAST chunk adds the row's partial sums (carry slots) in chunk order.  A row of more than
// kLongRow chunks (a hub: thousands of partials) is not walked by its 1..16 lanes alone: every group of the wave takes
// every GPW-th partial and the groups' sums are folded by shuffles (fixed order).  On the collab-shaped graph (d = 32)
// the one-group walk was 16 of the fix-up's 18 us.
template <typename T, int VEC, int G>
__global__ __launch_bounds__(kBlock) void spmm_fixup_kernel(const SpmmArgs a) {
  constexpr int GPW = kWave / G;
  constexpr int kLongRow = kFixupLongRow;
  const int lane = threadIdx.x & 63;
  const int gl = lane & (G - 1);
  const int64_t wave = static_cast<int64_t>(blockIdx.x) * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int kChunk = a.chunk;
  const int64_t E = a.rowptr[a.N];
  if (wave * GPW * kChunk >= E) return;             // wave-uniform
  const int64_t chunk = wave * GPW + lane / G;
  const int64_t start = chunk * kChunk;
  const int64_t c0 = static_cast<int64_t>(blockIdx.y) * (G * VEC) + gl * VEC;
  const bool cols = c0 < a.d;                        // this lane's columns exist
  bool active = start < E && start != 0;
  int r0 = 0;
  int64_t first = 0;
  if (active) {
    r0 = a.rowid[start];
    if (a.rowid[start - 1] != r0) active = false;    // first segment starts here: nothing pending
  }
  if (active) {
    const int64_t end = start + kChunk;
    if (end < E && a.rowid[end] == r0) active = false;   // row continues: a later chunk finishes it
  }
  if (active) first = a.rowptr[r0] / kChunk;         // chunk holding the row's first edge
  const bool lng = GPW > 1 && active && chunk - first > kLongRow;
  float acc[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) acc[i] = 0.f;
  const float* p = a.carry + c0;
  constexpr int FU = VEC >= 4 ? 16 : 8;             // partial rows in flight (a star row is a 40 K-long chain)
  if (active && cols) {
    carry_add<VEC>(a.carry + (2 * first + 1) * a.dpad + c0, acc);
    if (!lng) {
      int64_t c = first + 1;
      for (; c + FU <= chunk + 1; c += FU) {         // loaded together, added in chunk order
        float t[FU][VEC];
#pragma unroll
        for (int u = 0; u < FU; ++u) {
#pragma unroll
          for (int i = 0; i < VEC; ++i) t[u][i] = 0.f;
          carry_add<VEC>(p + 2 * (c + u) * a.dpad, t[u]);
        }
#pragma unroll
        for (int u = 0; u < FU; ++u)
#pragma unroll
          for (int i = 0; i < VEC; ++i) acc[i] += t[u][i];
      }
      for (; c <= chunk; ++c) carry_add<VEC>(p + 2 * c * a.dpad, acc);
    }
  }
  if constexpr (GPW > 1) {
    uint64_t todo = __ballot(lng);                   // every lane of a long group has its bit set
    while (todo) {                                   // wave-uniform
      const int src = __builtin_ctzll(todo);         // lowest lane of the group
      todo &= ~(((uint64_t(1) << G) - 1) << src);
      const int64_t f = __shfl(first, src), ch = __shfl(chunk, src);
      float part[VEC];
#pragma unroll
      for (int i = 0; i < VEC; ++i) part[i] = 0.f;
      if (cols) {
        constexpr int HU = FU / 2;
        int64_t c = f + 1 + lane / G;
        for (; c + static_cast<int64_t>(HU - 1) * GPW <= ch; c += static_cast<int64_t>(HU) * GPW) {
          float t[HU][VEC];
#pragma unroll
          for (int u = 0; u < HU; ++u) {
#pragma unroll
            for (int i = 0; i < VEC; ++i) t[u][i] = 0.f;
            carry_add<VEC>(p + 2 * (c + static_cast<int64_t>(u) * GPW) * a.dpad, t[u]);
          }
#pragma unroll
          for (int u = 0; u < HU; ++u)
#pragma unroll
            for (int i = 0; i < VEC; ++i) part[i] += t[u][i];
        }
        for (; c <= ch; c += GPW) carry_add<VEC>(p + 2 * c * a.dpad, part);
      }
#pragma unroll
      for (int off = G; off < kWave; off <<= 1)
#pragma unroll
        for (int i = 0; i < VEC; ++i) part[i] += __shfl_xor(part[i], off);
      if (lane / G == src / G)
#pragma unroll
        for (int i = 0; i < VEC; ++i) acc[i] += part[i];
    }
  }
  if (active && cols) store_out<T, VEC>(a, r0, c0, acc);
}

// ---- rows without edges: act(bias) -------------------------------------------------------------
template <typename T, int VEC>
__global__ __launch_bounds__(kBlock) void spmm_empty_rows_kernel(const SpmmArgs a) {
  const int64_t dv = a.d / VEC;
  const int64_t total = static_cast<int64_t>(*a.n_empty) * dv;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;
  for (int64_t t = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; t < total; t += stride) {
    const int64_t i = t / dv, c = (t - i * dv) * VEC;
    const int64_t row = a.empty_rows[i];
    float acc[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) acc[k] = 0.f;
    store_out<T, VEC>(a, row, c, acc);
  }
}

template <typename T, int VEC, int G>
int launch_g(const SpmmArgs& a, int64_t e_cap, hipStream_t st) {
  constexpr int GPW = kWave / G;
  const int kChunk = a.chunk;
  const int64_t n_chunks = ceil_div(e_cap, kChunk);
  const Tune tn = tune();
  // Occupancy limiter.  On big graphs the row gather runs FASTER with one or two workgroups per CU
  // than at full occupancy (profiles/r01/sweep_notes.md): fp32 d=128  9.27 ms full -> 8.94 (2/CU) ->
  // 8.28 (1/CU); fp32 d=64  4.65 -> 4.20 (2/CU) | 4.61 (1/CU); bf16 d=128  5.18 -> 4.76 (2/CU) |
  // 6.17 (1/CU: the bf16 widen leaves too little overlap with one wave per SIMD).  Reserving LDS
  // per workgroup admits exactly that many: > 80 KiB -> one, 64 KiB -> two of the CU's 160 KiB.
  // Small graphs are latency bound and keep the default.
  // Round 2: at the same number of waves per CU, 64-thread workgroups beat one or two 256-thread ones (a wave that has
  // finished its chunks frees its slot instead of waiting for three siblings): fp32 d = 128 four waves per CU 8.31 ->
  // 8.01 ms, d = 256 33.7 -> 33.3, bf16 d = 256 20.8 -> 19.7; fp32 d = 64 six waves per CU 4.02 -> 3.89 (bf16 d = 128 and
  // narrower rows: no change) -- same-box sweeps of spmm_lds / spmm_block.
  int lds = tn.lds, block = tn.block ? tn.block : kBlock;
  if (lds < 0) {
    lds = 0;
    // ... only where the gathers actually go to HBM: X must be much larger than the 256 MB Infinity Cache.
    // A block-diagonal batch of SEAL subgraphs (X = 187 MB, every gather lands in its own subgraph's few
    // KB) is L2-latency bound instead and wants every wave it can get: 3.4 -> 1.9 ms for the 4-layer stack.
    const int64_t x_bytes = a.n_cols * a.ldx * static_cast<int64_t>(sizeof(T));
    if (e_cap >= (int64_t(1) << 20) && x_bytes >= (int64_t(384) << 20)) {
      constexpr int tile_bytes = G * VEC * static_cast<int>(sizeof(T));     // bytes of an X row one group gathers
      if (tn.block == 0 && tile_bytes >= 512) { block = 64; lds = 40960; }             // four waves per CU
      else if (tn.block == 0 && tile_bytes >= 256) { block = 64; lds = 26000; }        // six
      else lds = (sizeof(T) == 4 && tile_bytes >= 512) ? 90000 : 65536;                // (a forced block size: the 256-thread limits)
    }
  }
  const int64_t n_waves = ceil_div(n_chunks, GPW);
  const int64_t n_blocks = ceil_div(n_waves, block / kWave);
  const dim3 blk(block);
  const int64_t tiles = ceil_div(a.d, G * VEC);
  LPG_REQUIRE(n_blocks < (int64_t(1) << 31) && tiles < 65536, "csr_spmm: grid too large");
  dim3 grid(static_cast<unsigned>(n_blocks), static_cast<unsigned>(tiles));
  const bool epi = a.row_scale || a.bias || a.act != LPG_ACT_NONE;
#define LPG_LAUNCH_MAIN(W, EPK, MD) \
  hipLaunchKernelGGL((spmm_main_kernel<T, VEC, G, W, EPK, 8, MD>), grid, blk, lds, st, a)
  static_assert(LPG_ACT_NONE == 0 && LPG_ACT_RELU == 1 && LPG_ACT_TANH == 2, "EPIK = 1 + act");
#ifdef LPG_TUNING
  if (tn.mode == 2 && !a.val && !epi) {
    LPG_LAUNCH_MAIN(false, 0, 2);              // measurement only: gather-only upper bound (wrong results)
  } else
#endif
  if (tn.mode == 1 && !a.val && !epi) {
    LPG_LAUNCH_MAIN(false, 0, 1);              // index prefetch variant
  } else if (!epi) {
    if (a.val) LPG_LAUNCH_MAIN(true, 0, 0); else LPG_LAUNCH_MAIN(false, 0, 0);
  } else if (a.act == LPG_ACT_NONE) {
    if (a.val) LPG_LAUNCH_MAIN(true, 1, 0); else LPG_LAUNCH_MAIN(false, 1, 0);
  } else if (a.act == LPG_ACT_RELU) {
    if (a.val) LPG_LAUNCH_MAIN(true, 2, 0); else LPG_LAUNCH_MAIN(false, 2, 0);
  } else {
    if (a.val) LPG_LAUNCH_MAIN(true, 3, 0); else LPG_LAUNCH_MAIN(false, 3, 0);
  }
#undef LPG_LAUNCH_MAIN
  phase_mark(1, st);
  hipLaunchKernelGGL((spmm_fixup_kernel<T, VEC, G>), grid, blk, 0, st, a);
  return LPG_OK;
}

template <typename T, int VEC>
int launch_vec(const SpmmArgs& a, int64_t e_cap, hipStream_t st) {
  const int64_t lanes = ceil_div(a.d, VEC);
  if (lanes <= 1) return launch_g<T, VEC, 1>(a, e_cap, st);
  if (lanes <= 2) return launch_g<T, VEC, 2>(a, e_cap, st);
  if (lanes <= 4) return launch_g<T, VEC, 4>(a, e_cap, st);
  if (lanes <= 8) return launch_g<T, VEC, 8>(a, e_cap, st);
  if (lanes <= 16) return launch_g<T, VEC, 16>(a, e_cap, st);
  if (lanes <= 32) return launch_g<T, VEC, 32>(a, e_cap, st);
  return launch_g<T, VEC, 64>(a, e_cap, st);
}

template <typename T, int WIDE>
int spmm_entry(const void* plan, const int64_t* rowptr, const int32_t* col, const float* val, const T* x,
               int64_t ldx, T* y, int64_t ldy, int64_t N, int64_t n_cols, int64_t e_cap, int64_t d, const float* row_scale,
               const float* bias, int act, int64_t n_empty_hint, int64_t max_row_hint, void* workspace,
               int64_t workspace_bytes, void* stream) {
  LPG_REQUIRE(N >= 0 && e_cap >= 0 && d >= 1, "csr_spmm: bad sizes N=%lld e_cap=%lld d=%lld", (long long)N,
              (long long)e_cap, (long long)d);
  if (N == 0) return LPG_OK;
  LPG_REQUIRE(plan && rowptr && x && y && (e_cap == 0 || col), "csr_spmm: null pointer");
  LPG_REQUIRE(ldx >= d && ldy >= d, "csr_spmm: leading dimension smaller than d");
  LPG_REQUIRE(act >= LPG_ACT_NONE && act <= LPG_ACT_TANH, "csr_spmm: unknown activation %d", act);
  const int64_t need = lpg_csr_spmm_bytes(e_cap, d);
  if (workspace_bytes < need || (!workspace && need > 0)) {
    set_error("csr_spmm: workspace %lld < %lld bytes", (long long)workspace_bytes, (long long)need);
    return LPG_E_WORKSPACE;
  }
  hipStream_t st = as_stream(stream);
  const char* p = static_cast<const char*>(plan);
  SpmmArgs a;
  a.rowptr = rowptr; a.col = col; a.val = val;
  a.n_empty = reinterpret_cast<const int32_t*>(p);
  a.rowid = reinterpret_cast<const int32_t*>(p + 256);
  a.empty_rows = reinterpret_cast<const int32_t*>(p + 256 + align_up((e_cap + 1) * 4, 256));
  a.x = x; a.y = y; a.ldx = ldx; a.ldy = ldy; a.N = N; a.d = d; a.n_cols = n_cols;
  a.row_scale = row_scale; a.bias = bias; a.act = act;
  a.carry = static_cast<float*>(workspace);
  a.dpad = align_up(d, 4);
  a.chunk = chunk_for(e_cap);
  a.off32 = (n_cols > 0 && static_cast<uint64_t>(n_cols) * static_cast<uint64_t>(ldx) * sizeof(T) <= (uint64_t(1) << 32)) ? 1 : 0;
  if (!tune().off32) a.off32 = 0;
  // short rows (host knows the longest), 16-byte rows, not a huge graph: one launch, a group per row
  const bool wide_all = d % WIDE == 0 && ldx % WIDE == 0 && ldy % WIDE == 0 &&
                        reinterpret_cast<uintptr_t>(x) % 16 == 0 && reinterpret_cast<uintptr_t>(y) % 16 == 0;
  // a group walks its row alone, so the kernel takes as long as the LONGEST row: worth it only for really short
  // rows (fb-pages-food's graphs have rows of ~110 edges and are faster on the chunked kernels, 110 vs 116 us / forward)
  const int64_t rows_max = options().spmm_rows_max;    // 0 disables
  const int64_t rows_ecap = options().spmm_rows_ecap;
  if (max_row_hint >= 0 && max_row_hint <= rows_max && wide_all && d / WIDE >= 2 && e_cap <= rows_ecap &&
      tune().mode == 0) {
    const int rc = launch_rows<T, WIDE>(a, st);
    if (rc != LPG_OK) return rc;
    return check_launch("csr_spmm (rows)");
  }
  const bool wide_out = d % WIDE == 0 && ldy % WIDE == 0 && reinterpret_cast<uintptr_t>(y) % 16 == 0;
  auto launch_fill = [&](hipStream_t s) {
    if (wide_out)
      hipLaunchKernelGGL((spmm_empty_rows_kernel<T, WIDE>), dim3(grid_for(N * (d < 64 ? d : 64) / WIDE)), dim3(kBlock), 0, s, a);
    else
      hipLaunchKernelGGL((spmm_empty_rows_kernel<T, 1>), dim3(grid_for(N * (d < 64 ? d : 64))), dim3(kBlock), 0, s, a);
  };
  // big graphs: the empty rows are filled on a second stream while the main kernel gathers (see SideStream)
  SideStream* side = nullptr;
  if (n_empty_hint != 0 && options().spmm_fill_overlap && e_cap >= (int64_t(1) << 24) &&
      n_empty_hint * d * static_cast<int64_t>(sizeof(T)) >= (int64_t(64) << 20))
    side = side_stream(st);
  if (side) {
    if (hipEventRecord(side->fork, st) == hipSuccess && hipStreamWaitEvent(side->s, side->fork, 0) == hipSuccess) {
      launch_fill(side->s);
      if (hipEventRecord(side->join, side->s) != hipSuccess) {
        set_error("csr_spmm: cannot record the fill's completion");
        return LPG_E_LAUNCH;
      }
    } else {
      (void)hipGetLastError();
      side = nullptr;
    }
  }
  phase_mark(0, st);
  phase_mark(1, st);
  // d <= 16 in a power of two: the edge-parallel kernel (lanes = edges); wider rows: lanes = feature columns
  const bool narrow = (d == 1 || d == 2 || d == 4 || d == 8 || d == 16) && tune().mode == 0 && !options().spmm_no_edge;
  if (e_cap > 0 && narrow) {
    const int64_t n_chunks = ceil_div(e_cap, a.chunk);
    const int64_t gx = ceil_div(n_chunks, kBlock / kWave);
    if (gx >= (int64_t(1) << 31)) {
      if (side) (void)hipStreamWaitEvent(st, side->join, 0);      // never return while the hidden stream still writes y
      set_error("csr_spmm: grid too large");
      return LPG_E_INVALID;
    }
    constexpr int PV = 16 / sizeof(T);
    const bool vio = d % PV == 0 && ldx % PV == 0 && ldy % PV == 0 && reinterpret_cast<uintptr_t>(x) % 16 == 0 &&
                     reinterpret_cast<uintptr_t>(y) % 16 == 0;
    const dim3 eg(static_cast<unsigned>(gx)), blk(kBlock);
#define LPG_EDGE_IO(DVV, VIOV)                                                                      \
  do {                                                                                              \
    if (a.val) hipLaunchKernelGGL((spmm_edge_kernel<T, DVV, true, VIOV>), eg, blk, 0, st, a);       \
    else hipLaunchKernelGGL((spmm_edge_kernel<T, DVV, false, VIOV>), eg, blk, 0, st, a);            \
  } while (0)
#define LPG_EDGE(DVV)                                                                               \
  do {                                                                                              \
    if constexpr (DVV % PV == 0) { if (vio) LPG_EDGE_IO(DVV, true); else LPG_EDGE_IO(DVV, false); } \
    else LPG_EDGE_IO(DVV, false);                                                                   \
    constexpr int FG = DVV;                                                                         \
    const int64_t fw = ceil_div(n_chunks, kWave / FG);                                              \
    hipLaunchKernelGGL((spmm_fixup_kernel<T, 1, FG>), dim3(static_cast<unsigned>(ceil_div(fw, kBlock / kWave))), blk, 0, st, a); \
  } while (0)
    if (d == 1) LPG_EDGE(1); else if (d == 2) LPG_EDGE(2); else if (d == 4) LPG_EDGE(4);
    else if (d == 8) LPG_EDGE(8); else LPG_EDGE(16);
#undef LPG_EDGE
#undef LPG_EDGE_IO
  } else if (e_cap > 0) {
    const bool wide = d % WIDE == 0 && ldx % WIDE == 0 && ldy % WIDE == 0 &&
                      reinterpret_cast<uintptr_t>(x) % 16 == 0 && reinterpret_cast<uintptr_t>(y) % 16 == 0;
    const int rc = wide ? launch_vec<T, WIDE>(a, e_cap, st) : launch_vec<T, 1>(a, e_cap, st);
    if (rc != LPG_OK) {
      if (side) (void)hipStreamWaitEvent(st, side->join, 0);      // the caller's stream owns y again before the error returns
      return rc;
    }
  }
  phase_mark(2, st);
  if (n_empty_hint == 0) {
    phase_mark(3, st);
    return check_launch("csr_spmm");
  }
  if (side) {
    if (hipStreamWaitEvent(st, side->join, 0) != hipSuccess) {
      set_error("csr_spmm: cannot join the fill stream");
      return LPG_E_LAUNCH;
    }
  } else {
    launch_fill(st);
  }
  phase_mark(3, st);
  return check_launch("csr_spmm");
}
}  // namespace

LPG_EXPORT int lpg_csr_spmm_phase_ms(float* out_ms) {
  LPG_REQUIRE(out_ms, "csr_spmm_phase_ms: null pointer");
  LPG_REQUIRE(g_phase_recorded, "csr_spmm_phase_ms: no call was timed (set option spmm_phase_timing = 1 first)");
  LPG_HIP(hipEventSynchronize(g_phase_ev[3]));
  for (int i = 0; i < 3; ++i) LPG_HIP(hipEventElapsedTime(out_ms + i, g_phase_ev[i], g_phase_ev[i + 1]));
  return LPG_OK;
}

LPG_EXPORT int lpg_spmm_chunk_edges(int64_t e_cap) { return chunk_for(e_cap); }

// the association order of the fix-up, for the same-order oracle (oracle/spmm_ref.c): a row whose pieces after the first
// number more than lpg_spmm_fixup_long_row() is summed as `groups` interleaved left-to-right sub-sums (piece j goes to
// sub-sum j % groups) folded by a balanced binary tree; shorter rows left to right.  `wide` = d, ldx, ldy multiples of
// 16 bytes and x, y 16-byte aligned (what selects the 16-byte-per-lane kernels).
LPG_EXPORT int lpg_spmm_fixup_groups(int64_t d, int elem_bytes, int wide) {
  if (d < 1 || (elem_bytes != 2 && elem_bytes != 4)) return LPG_E_INVALID;
  if (d == 1 || d == 2 || d == 4 || d == 8 || d == 16) {
    if (!options().spmm_no_edge) return kWave / static_cast<int>(d);
  }
  const int vec = wide ? 16 / elem_bytes : 1;
  const int64_t lanes = ceil_div(d, vec);
  int g = 1;
  while (g < lanes && g < kWave) g *= 2;
  return kWave / g;
}
LPG_EXPORT int lpg_spmm_fixup_long_row(void) { return kFixupLongRow; }

LPG_EXPORT int64_t lpg_csr_spmm_bytes(int64_t e_cap, int64_t d) {
  if (e_cap < 0 || d < 1) return LPG_E_INVALID;
  return 2 * lpg::ceil_div(e_cap, chunk_for(e_cap)) * lpg::align_up(d, 4) * 4 + 256;
}

LPG_EXPORT int lpg_csr_spmm_f32(const void* plan, const int64_t* rowptr, const int32_t* col, const float* val,
                                const float* x, int64_t ldx, float* y, int64_t ldy, int64_t N, int64_t n_cols,
                                int64_t e_cap, int64_t d, const float* row_scale, const float* bias, int act,
                                int64_t n_empty_hint, int64_t max_row_hint, void* workspace, int64_t workspace_bytes,
                                void* stream) {
  return spmm_entry<float, 4>(plan, rowptr, col, val, x, ldx, y, ldy, N, n_cols, e_cap, d, row_scale, bias, act, n_empty_hint,
                              max_row_hint, workspace, workspace_bytes, stream);
}

LPG_EXPORT int lpg_csr_spmm_bf16(const void* plan, const int64_t* rowptr, const int32_t* col, const float* val,
                                 const uint16_t* x, int64_t ldx, uint16_t* y, int64_t ldy, int64_t N, int64_t n_cols,
                                 int64_t e_cap, int64_t d, const float* row_scale, const float* bias, int act,
                                 int64_t n_empty_hint, int64_t max_row_hint, void* workspace, int64_t workspace_bytes,
                                 void* stream) {
  return spmm_entry<uint16_t, 8>(plan, rowptr, col, val, x, ldx, y, ldy, N, n_cols, e_cap, d, row_scale, bias, act,
                                 n_empty_hint, max_row_hint, workspace, workspace_bytes, stream);
}

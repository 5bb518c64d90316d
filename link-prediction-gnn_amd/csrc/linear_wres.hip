// Dense feature linear for BIG n with a SMALL weight matrix (the GNN shape: n = |V| or the rows of a
// whole subgraph batch, k and m <= 256):  Y[n,m] = act(X[n,k] @ W^T + bias).
// Reference anchor: same op as linear.hip / linear_bf16.hip (GCNConv.lin, TwoWL/model/model.py:37
// [3P PyG]; SEAL's tf.matmul(X, W), SEAL/model/gnn.py:46-58); this file only changes HOW it runs.
//
// Persistent, W-resident design.  The tiled kernels of linear.hip re-stage W from L2 for every 128
// rows and meet at two workgroup barriers per 64-deep K step; at n = 5 M that is 39 K workgroups each
// re-reading the same 64 KB and waves that spend their time waiting for each other.  Here:
//   * a workgroup copies W into LDS ONCE ([NT*32 columns][k], row-major, rows padded by 16 B) and then
//     never synchronises again;
//   * each wave walks 32-row tiles of X on its own (tile = wave id, + number of waves, ...): the
//     16-byte pieces of the NEXT tile are loaded into registers before the MFMAs of the current one,
//     so 8-16 KB per wave is always in flight (x 8 or 4 waves per CU: enough to cover HBM latency);
//   * the tile is transposed through a wave-private LDS slice (coalesced 16-byte global loads in, one
//     ds_read_b128 per fragment out), the result tile goes back through the same slice and leaves as
//     whole 16-byte pieces of a row.
// Fragments.  Both operands are read as 16 contiguous bytes of a row of [rows][k]:
//   bf16: 8 values = exactly the A / B fragment of v_mfma_f32_32x32x16_bf16 (lane = row | k-half);
//   fp32: 4 values feed FOUR v_mfma_f32_32x32x2_f32 steps -- lane half h of step i carries
//         k = 8g + 4h + i.  The contraction order inside a group of 8 is permuted identically for X
//         and W, which a sum over k does not care about (fp32 products are exact, fp32 accumulate).
// Bounds: fp32 k = m = 128 is MFMA-bound (2*n*k*m flop at 157 TF = 1.04 ms for n = 5 M, the HBM side
// is 0.81 ms); bf16 is HBM-bound (x + y = 2.56 GB).
#include <stdlib.h>

#include "lpg_common.hpp"

namespace {
using namespace lpg;

using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) short;

template <typename T> struct Elt;
template <> struct Elt<float> {
  static constexpr int EPB = 4;    // elements per 16-byte piece
  static constexpr int GK = 8;     // k values consumed per fragment read (both lane halves)
  static __device__ __forceinline__ float from_f32(float v) { return v; }
};
template <> struct Elt<uint16_t> {
  static constexpr int EPB = 8;
  static constexpr int GK = 16;
  static __device__ __forceinline__ uint16_t from_f32(float v) { return f32_to_bf16(v); }
};

__device__ __forceinline__ void mfma_group(const uint4& a, const uint4& b, f32x16& acc, float*) {
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.x), __uint_as_float(b.x), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.y), __uint_as_float(b.y), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.z), __uint_as_float(b.z), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.w), __uint_as_float(b.w), acc, 0, 0, 0);
}
__device__ __forceinline__ void mfma_group(const uint4& a, const uint4& b, f32x16& acc, uint16_t*) {
  union { uint4 u; bf16x8 v; } ua, ub;
  ua.u = a; ub.u = b;
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ua.v, ub.v, acc, 0, 0, 0);
}

// one fragment group for all NT column tiles.  fp32: the four k-steps are the OUTER loop so that consecutive
// MFMAs write different accumulators (a dependent chain on one accumulator cannot issue back to back).
template <int NT>
__device__ __forceinline__ void mfma_groups(const uint4& a, const uint4 (&b)[NT], f32x16 (&acc)[NT], float*) {
  const uint32_t av[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const uint32_t bv = i == 0 ? b[t].x : (i == 1 ? b[t].y : (i == 2 ? b[t].z : b[t].w));
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(av[i]), __uint_as_float(bv), acc[t], 0, 0, 0);
    }
}
template <int NT>
__device__ __forceinline__ void mfma_groups(const uint4& a, const uint4 (&b)[NT], f32x16 (&acc)[NT], uint16_t*) {
#pragma unroll
  for (int t = 0; t < NT; ++t) mfma_group(a, b[t], acc[t], static_cast<uint16_t*>(nullptr));
}

// zero the elements >= keep of a 16-byte piece (keep in [0, EPB])
__device__ __forceinline__ uint4 keep_first(uint4 v, int keep, float*) {
  v.x = keep > 0 ? v.x : 0u; v.y = keep > 1 ? v.y : 0u; v.z = keep > 2 ? v.z : 0u; v.w = keep > 3 ? v.w : 0u;
  return v;
}
__device__ __forceinline__ uint4 keep_first(uint4 v, int keep, uint16_t*) {
  auto two = [&](uint32_t d, int i) {   // d holds elements 2i (low half) and 2i+1 (high half)
    return (keep > 2 * i ? d & 0xffffu : 0u) | (keep > 2 * i + 1 ? d & 0xffff0000u : 0u);
  };
  v.x = two(v.x, 0); v.y = two(v.y, 1); v.z = two(v.z, 2); v.w = two(v.w, 3);
  return v;
}

// W -> LDS, once per workgroup: [NT*32 columns][KPAD + EPB], zero outside [m, k].  Batches of 8 loads
// from clamped addresses, then the selects (written as `cond ? load : 0` hipcc branches around every
// load and waits for each one; see linear.hip).
template <typename T, int NT, int KPAD, int BLOCK = kBlock>
__device__ __forceinline__ void stage_w(T* Ws, const T* __restrict__ w, int w_is_km, int j0, int k, int m, int tid) {
  constexpr int LD = KPAD + Elt<T>::EPB, TOTAL = NT * 32 * KPAD;
  for (int base = 0; base < TOTAL; base += 8 * BLOCK) {
    T v[8];
    int jj[8], kk[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int idx = min(base + u * BLOCK + tid, TOTAL - 1);
      if (w_is_km) { kk[u] = idx / (NT * 32); jj[u] = idx % (NT * 32); }   // lanes walk m
      else { jj[u] = idx / KPAD; kk[u] = idx % KPAD; }                     // lanes walk k
      const int jc = min(j0 + jj[u], m - 1), kc = min(kk[u], k - 1);
      v[u] = w[w_is_km ? static_cast<int64_t>(kc) * m + jc : static_cast<int64_t>(jc) * k + kc];
    }
    asm volatile("" ::: "memory");
#pragma unroll
    for (int u = 0; u < 8; ++u)     // (a clamped tail index rewrites the last element with its own value)
      Ws[jj[u] * LD + kk[u]] = (j0 + jj[u] < m && kk[u] < k) ? v[u] : T(0);
  }
}

__device__ __forceinline__ uint4 and4(uint4 v, uint32_t mask) {
  v.x &= mask; v.y &= mask; v.z &= mask; v.w &= mask;
  return v;
}

// ---------------------------------------------------------------------------------------------------
// Variant 1 (HBM-bound shapes): X tile through a wave-private LDS slice, coalesced 16-byte loads and stores.
// PG = KPAD / GK = 16-byte pieces of a 32-row X tile per lane; k is padded to KPAD with zeros in LDS.
// (NT = 4, PG = 16 needs 135 KB of LDS: one workgroup per CU anyway, so it may use the whole register file.)
template <typename T, int NT, int PG, int ACT>
__global__ __launch_bounds__(kBlock, (NT == 4 && PG == 16) ? 1 : 2) void linear_wres_kernel(
    const T* __restrict__ x, int64_t ldx, const T* __restrict__ w, int w_is_km, const float* __restrict__ bias,
    T* __restrict__ y, int64_t ldy, int64_t n, int k, int m, int64_t n_tiles, int y_vec) {
  constexpr int EPB = Elt<T>::EPB, GK = Elt<T>::GK, KPAD = PG * GK;
  constexpr int LD = KPAD + EPB;                    // LDS row of X / W (elements); a 16-byte multiple
  constexpr int LDO = NT * 32 + EPB;                // LDS row of the output tile
  constexpr int TLD = LD > LDO ? LD : LDO;
  constexpr int PPR = KPAD / EPB;                   // pieces per row: 8, 16 or 32
  extern __shared__ uint4 smem_raw[];
  T* smem = reinterpret_cast<T*>(smem_raw);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  T* Ws = smem;                                                     // [NT*32][LD]
  T* Xs = smem + NT * 32 * LD + wave * 32 * TLD;                    // wave-private [32][TLD]
  const int j0 = blockIdx.y * (NT * 32);
  const int fr = lane & 31, fk = lane >> 5;
  stage_w<T, NT, KPAD>(Ws, w, w_is_km, j0, k, m, tid);
  __syncthreads();                                  // the only workgroup barrier

  float bv[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) bv[t] = (bias && j0 + t * 32 + fr < m) ? bias[j0 + t * 32 + fr] : 0.f;
  const bool ktail = (k % EPB) != 0;
  const int64_t n_waves = static_cast<int64_t>(gridDim.x) * (kBlock / 64);
  int64_t tile = static_cast<int64_t>(blockIdx.x) * (kBlock / 64) + wave;

  // piece p = it * 64 + lane of a tile: row p / PPR, 16-byte segment p % PPR.  Loads always come from a
  // valid address (row clamped to n - 1, segments at or past k read segment 0) and are zeroed in registers.
  uint4 xr[PG];
  auto load_tile = [&](int64_t tl) {
    const int64_t row0 = tl * 32;
#pragma unroll
    for (int it = 0; it < PG; ++it) {
      const int p = it * 64 + lane, r = p / PPR, seg = p % PPR;
      int64_t row = row0 + r;
      row = row < n ? row : n - 1;
      xr[it] = *reinterpret_cast<const uint4*>(x + row * ldx + (seg * EPB < k ? seg * EPB : 0));
    }
  };
  if (tile < n_tiles) load_tile(tile);

  for (; tile < n_tiles; tile += n_waves) {
    const int64_t row0 = tile * 32;
    asm volatile("" ::: "memory");
#pragma unroll
    for (int it = 0; it < PG; ++it) {
      const int p = it * 64 + lane, r = p / PPR, seg = p % PPR;
      const int keep = k - seg * EPB;
      uint4 v = and4(xr[it], (row0 + r < n && keep > 0) ? 0xffffffffu : 0u);
      if (ktail) v = keep_first(v, keep, static_cast<T*>(nullptr));      // uniform branch
      *reinterpret_cast<uint4*>(Xs + r * TLD + seg * EPB) = v;
    }
    {
      const int64_t nx = tile + n_waves;
      load_tile(nx < n_tiles ? nx : tile);           // in flight during the MFMAs below (last tile: reloads itself)
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);              // lgkmcnt(0): the slice is wave-private
    __builtin_amdgcn_wave_barrier();

    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
    const T* ap = Xs + fr * TLD + fk * (GK / 2);
    const T* bp = Ws + fr * LD + fk * (GK / 2);
#pragma unroll
    for (int g = 0; g < PG; ++g) {
      const uint4 a = *reinterpret_cast<const uint4*>(ap + g * GK);
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const uint4 b = *reinterpret_cast<const uint4*>(bp + t * 32 * LD + g * GK);
        mfma_group(a, b, acc[t], static_cast<T*>(nullptr));
      }
    }
    __builtin_amdgcn_wave_barrier();                 // every fragment read precedes the overwrite below

    // ---- result tile -> the same LDS slice -> 16-byte row pieces
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int c = t * 32 + fr;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * fk;
        Xs[row * TLD + c] = Elt<T>::from_f32(apply_act(acc[t][r] + bv[t], ACT));
      }
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_wave_barrier();
    constexpr int OPR = NT * 32 / EPB;               // output pieces per row
#pragma unroll
    for (int it = 0; it < OPR / 2; ++it) {           // 32 * OPR pieces, 64 lanes
      const int p = it * 64 + lane, r = p / OPR, c = (p % OPR) * EPB;
      const int64_t row = row0 + r;
      const int col = j0 + c;
      if (row < n && col < m) {
        const T* src = Xs + r * TLD + c;
        T* dst = y + row * ldy + col;
        if (y_vec && col + EPB <= m) {
          store_stream_u4(dst, *reinterpret_cast<const uint4*>(src));
        } else {
#pragma unroll
          for (int e = 0; e < EPB; ++e)
            if (col + e < m) dst[e] = src[e];
        }
      }
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_wave_barrier();                 // the slice is free for the next tile
  }
}

// ---------------------------------------------------------------------------------------------------
// Variant 2 (MFMA-bound shapes): no LDS staging of X at all.  A lane loads its A fragments straight from
// global memory (16 bytes of row fr at k offset g*GK + fk*GK/2: two lanes cover 32 contiguous bytes of a
// row, the four groups of a 128-byte line are requested back to back and hit in L1), the fragment
// registers of tile i+1 are refilled in place right after the MFMAs of tile i have consumed them, and the
// result is stored from the accumulator layout (one store covers 32 consecutive columns of two rows).
// LDS holds only W, so twice the waves fit and there is no per-tile LDS round trip.
template <typename T, int NT, int PG, int ACT>
__global__ __launch_bounds__(kBlock, 2) void linear_wres2_kernel(
    const T* __restrict__ x, int64_t ldx, const T* __restrict__ w, int w_is_km, const float* __restrict__ bias,
    T* __restrict__ y, int64_t ldy, int64_t n, int k, int m, int64_t n_tiles) {
  constexpr int EPB = Elt<T>::EPB, GK = Elt<T>::GK, KPAD = PG * GK, LD = KPAD + EPB;
  extern __shared__ uint4 smem_raw[];
  T* Ws = reinterpret_cast<T*>(smem_raw);            // [NT*32][LD]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j0 = blockIdx.y * (NT * 32);
  const int fr = lane & 31, fk = lane >> 5;
  stage_w<T, NT, KPAD>(Ws, w, w_is_km, j0, k, m, tid);
  __syncthreads();

  float bv[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) bv[t] = (bias && j0 + t * 32 + fr < m) ? bias[j0 + t * 32 + fr] : 0.f;
  const bool kmask = k != KPAD;                      // some fragment reaches past k: mask it in registers
  const int64_t n_waves = static_cast<int64_t>(gridDim.x) * (kBlock / 64);
  int64_t tile = static_cast<int64_t>(blockIdx.x) * (kBlock / 64) + wave;
  const int ko0 = fk * (GK / 2);
  const T* bp = Ws + fr * LD + ko0;

  // rows >= n read row n-1 (their results are never stored); fragments at k offsets >= k read offset 0
  auto row_ptr = [&](int64_t tl) {
    int64_t r = tl * 32 + fr;
    r = r < n ? r : n - 1;
    return x + r * ldx;
  };
  uint4 xr[PG];
  if (tile < n_tiles) {
    const T* rp = row_ptr(tile);
#pragma unroll
    for (int g = 0; g < PG; ++g) xr[g] = *reinterpret_cast<const uint4*>(rp + (g * GK + ko0 < k ? g * GK + ko0 : 0));
  }
  for (; tile < n_tiles; tile += n_waves) {
    const int64_t row0 = tile * 32;
    const int64_t nx = tile + n_waves;
    const T* rp = row_ptr(nx < n_tiles ? nx : tile);  // last tile: reloads itself, unused
    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
#pragma unroll
    for (int g = 0; g < PG; ++g) {
      uint4 a = xr[g];
      const int ko = g * GK + ko0;
      if (kmask) {                                   // uniform branch
        a = and4(a, k - ko > 0 ? 0xffffffffu : 0u);
        a = keep_first(a, k - ko, static_cast<T*>(nullptr));
      }
      xr[g] = *reinterpret_cast<const uint4*>(rp + (ko < k ? ko : 0));   // refill in place for the next tile
      uint4 b[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) b[t] = *reinterpret_cast<const uint4*>(bp + t * 32 * LD + g * GK);
      mfma_groups<NT>(a, b, acc, static_cast<T*>(nullptr));
      // without this the scheduler hoists every W fragment read of the unrolled loop to the top
      // (PG * NT * 4 registers) and spills the accumulators
      if (g % 2 == 1) __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int col = j0 + t * 32 + fr;
      if (col < m) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int64_t row = row0 + (r & 3) + 8 * (r >> 2) + 4 * fk;
          if (row < n) y[row * ldy + col] = Elt<T>::from_f32(apply_act(acc[t][r] + bv[t], ACT));
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// Variant 3 = variant 2 with the result leaving through LDS.  Stores from the accumulator layout are 4 (2) bytes
// per lane, 64 store instructions per 32 x 128 tile; measured on fp32 k = m = 128 they cost 0.38 of the 1.51 ms
// although the bytes are few -- the memory pipeline is paid per instruction.  Here a wave writes half a tile
// (16 rows) into a private LDS slice and stores it as whole 16-byte row pieces, twice: 16 instructions per tile.
// The slices do not fit next to two W copies, so ONE 512-thread workgroup per CU shares one W (8 waves).
constexpr int kBlock3 = 512;
template <typename T, int NT, int PG, int ACT>
__global__ __launch_bounds__(kBlock3, 1) void linear_wres3_kernel(
    const T* __restrict__ x, int64_t ldx, const T* __restrict__ w, int w_is_km, const float* __restrict__ bias,
    T* __restrict__ y, int64_t ldy, int64_t n, int k, int m, int64_t n_tiles, int y_vec) {
  constexpr int EPB = Elt<T>::EPB, GK = Elt<T>::GK, KPAD = PG * GK, LD = KPAD + EPB;
  constexpr int LDO = NT * 32 + EPB;               // LDS row of the half output tile
  extern __shared__ uint4 smem_raw[];
  T* Ws = reinterpret_cast<T*>(smem_raw);            // [NT*32][LD]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  T* Os = Ws + NT * 32 * LD + wave * 16 * LDO;       // wave-private [16][LDO]
  const int j0 = blockIdx.y * (NT * 32);
  const int fr = lane & 31, fk = lane >> 5;
  stage_w<T, NT, KPAD, kBlock3>(Ws, w, w_is_km, j0, k, m, tid);
  __syncthreads();

  float bv[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) bv[t] = (bias && j0 + t * 32 + fr < m) ? bias[j0 + t * 32 + fr] : 0.f;
  const bool kmask = k != KPAD;
  const int64_t n_waves = static_cast<int64_t>(gridDim.x) * (kBlock3 / 64);
  int64_t tile = static_cast<int64_t>(blockIdx.x) * (kBlock3 / 64) + wave;
  const int ko0 = fk * (GK / 2);
  const T* bp = Ws + fr * LD + ko0;
  auto row_ptr = [&](int64_t tl) {
    int64_t r = tl * 32 + fr;
    r = r < n ? r : n - 1;
    return x + r * ldx;
  };
  uint4 xr[PG];
  if (tile < n_tiles) {
    const T* rp = row_ptr(tile);
#pragma unroll
    for (int g = 0; g < PG; ++g) xr[g] = *reinterpret_cast<const uint4*>(rp + (g * GK + ko0 < k ? g * GK + ko0 : 0));
  }
  for (; tile < n_tiles; tile += n_waves) {
    const int64_t row0 = tile * 32;
    const int64_t nx = tile + n_waves;
    const T* rp = row_ptr(nx < n_tiles ? nx : tile);
    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
#pragma unroll
    for (int g = 0; g < PG; ++g) {
      uint4 a = xr[g];
      const int ko = g * GK + ko0;
      if (kmask) {
        a = and4(a, k - ko > 0 ? 0xffffffffu : 0u);
        a = keep_first(a, k - ko, static_cast<T*>(nullptr));
      }
      xr[g] = *reinterpret_cast<const uint4*>(rp + (ko < k ? ko : 0));
      uint4 b[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) b[t] = *reinterpret_cast<const uint4*>(bp + t * 32 * LD + g * GK);
      mfma_groups<NT>(a, b, acc, static_cast<T*>(nullptr));
      if (g % 2 == 1) __builtin_amdgcn_sched_barrier(0);
    }
    constexpr int OPR = NT * 32 / EPB;               // 16-byte pieces per output row
#pragma unroll
    for (int half = 0; half < 2; ++half) {           // accumulator registers 8*half .. 8*half+7 hold rows 16*half .. +15
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int c = t * 32 + fr;
#pragma unroll
        for (int r8 = 0; r8 < 8; ++r8) {
          const int r = half * 8 + r8;
          const int rl = (r & 3) + 8 * ((r >> 2) & 1) + 4 * fk;      // row inside the half tile
          Os[rl * LDO + c] = Elt<T>::from_f32(apply_act(acc[t][r] + bv[t], ACT));
        }
      }
      __builtin_amdgcn_s_waitcnt(0xC07F);            // lgkmcnt(0): the slice is wave-private
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int it = 0; it < (16 * OPR) / 64; ++it) {
        const int p = it * 64 + lane, rr = p / OPR, c = (p % OPR) * EPB;
        const int64_t row = row0 + half * 16 + rr;
        const int col = j0 + c;
        if (row < n && col < m) {
          const T* src = Os + rr * LDO + c;
          T* dst = y + row * ldy + col;
          if (y_vec && col + EPB <= m) {
            store_stream_u4(dst, *reinterpret_cast<const uint4*>(src));
          } else {
#pragma unroll
            for (int e = 0; e < EPB; ++e)
              if (col + e < m) dst[e] = src[e];
          }
        }
      }
      __builtin_amdgcn_s_waitcnt(0xC07F);
      __builtin_amdgcn_wave_barrier();               // the slice is rewritten by the next half
    }
  }
}

template <typename K>
int blocks_per_cu(K kern, size_t lds, int block = kBlock) {
  int b = 0;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                          160 * 1024) != hipSuccess)
    return -1;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&b, kern, block, lds) != hipSuccess) return -1;
  b += static_cast<int>(options().linear_percu_delta);     // A/B option (the API can be one block high)
  return b < 1 ? 1 : b;
}

int cu_count() {
  static int cus = 0;
  if (!cus) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
      cus = 256;
  }
  return cus;
}

// workgroups per column strip.  All strips must be RESIDENT TOGETHER: with cus * per_cu workgroups per strip the grid's
// second strip only started when the first had finished, and X came from HBM once per strip (bf16 256 -> 256: 2.07 GB
// fetched for 1.02 GB of X, profiles/r03/linear_wres_traffic.txt).  cus * per_cu / strips of them, a multiple of 8, put
// workgroup (i, strip 0) and (i, strip 1) -- the same row tiles at the same time -- on one XCD (ids i and i + gx), so
// the second reader finds the rows in that XCD's L2 (10 M rows: 7.1 GB fetched for 5.1 GB of X -- the pairs drift apart
// over a long launch; cutting the rows into one launch per 2^18 .. 2^20 rows to re-align them cost more than it returned).
// Also measured, not kept: BOTH strips resident in one 512-thread workgroup (2 x 66 KB of W, 8-row output slices), waves w
// and w + 4 taking the same row tiles -- X reaches the CU exactly once, but only four row tiles are in flight per CU
// instead of eight: 3.11 ms at 10 M rows against 2.55 (the kernel lives on bytes in flight, not on fetch volume).
inline int64_t strip_grid_x(int per_cu, unsigned gy, int64_t n_tiles, int waves) {
  int64_t gx = static_cast<int64_t>(cu_count()) * per_cu / (gy < 1 ? 1 : gy);
  if (gy > 1 && gx >= 8) gx -= gx % 8;
  gx = std::max<int64_t>(gx, 1);
  return std::min<int64_t>(gx, ceil_div(n_tiles, waves));
}

template <typename T, int NT, int PG, int ACT>
int launch_wres(const T* x, int64_t ldx, const T* w, int w_is_km, const float* bias, T* y, int64_t ldy, int64_t n,
                int k, int m, int y_vec, int variant, hipStream_t st) {
  constexpr int EPB = Elt<T>::EPB, KPAD = PG * Elt<T>::GK, LD = KPAD + EPB, LDO = NT * 32 + EPB;
  const int64_t n_tiles = ceil_div(n, 32);
  const unsigned gy = static_cast<unsigned>(ceil_div(m, NT * 32));
  static int per_cu[3] = {0, 0, 0};                 // per instantiation; the occupancy query is slow
  if constexpr (NT == 4 && PG == 16) {
    if (variant == 3) {
      auto kern = linear_wres3_kernel<T, NT, PG, ACT>;
      constexpr size_t lds = (static_cast<size_t>(NT) * 32 * LD + (kBlock3 / 64) * 16 * static_cast<size_t>(LDO)) * sizeof(T);
      if (!per_cu[2]) per_cu[2] = blocks_per_cu(kern, lds, kBlock3);
      LPG_REQUIRE(per_cu[2] > 0, "linear (W-resident): occupancy query failed");
      const int64_t gx = strip_grid_x(per_cu[2], gy, n_tiles, kBlock3 / 64);
      hipLaunchKernelGGL(kern, dim3(static_cast<unsigned>(gx), gy), dim3(kBlock3), lds, st, x, ldx, w, w_is_km, bias, y,
                         ldy, n, k, m, n_tiles, y_vec);
      return 1;
    }
  }
  if (variant == 2 || variant == 3 || PG == 20) {                   // PG = 20 (fp32, 128 < k <= 160) exists as variant 2 only
    auto kern = linear_wres2_kernel<T, NT, PG, ACT>;
    constexpr size_t lds = static_cast<size_t>(NT) * 32 * LD * sizeof(T);
    if (!per_cu[1]) per_cu[1] = blocks_per_cu(kern, lds);
    LPG_REQUIRE(per_cu[1] > 0, "linear (W-resident): occupancy query failed");
    const int64_t gx = strip_grid_x(per_cu[1], gy, n_tiles, kBlock / 64);
    hipLaunchKernelGGL(kern, dim3(static_cast<unsigned>(gx), gy), dim3(kBlock), lds, st, x, ldx, w, w_is_km, bias, y,
                       ldy, n, k, m, n_tiles);
  } else if constexpr (PG != 20) {
    auto kern = linear_wres_kernel<T, NT, PG, ACT>;
    constexpr size_t lds = (static_cast<size_t>(NT) * 32 * LD + 4 * 32 * static_cast<size_t>(LD > LDO ? LD : LDO)) * sizeof(T);
    if (!per_cu[0]) per_cu[0] = blocks_per_cu(kern, lds);
    LPG_REQUIRE(per_cu[0] > 0, "linear (W-resident): occupancy query failed");
    const int64_t gx = strip_grid_x(per_cu[0], gy, n_tiles, kBlock / 64);
    hipLaunchKernelGGL(kern, dim3(static_cast<unsigned>(gx), gy), dim3(kBlock), lds, st, x, ldx, w, w_is_km, bias, y,
                       ldy, n, k, m, n_tiles, y_vec);
  }
  return 1;
}

template <typename T, int NT, int PG>
int launch_act(const T* x, int64_t ldx, const T* w, int w_is_km, const float* bias, int act, T* y, int64_t ldy,
               int64_t n, int k, int m, int y_vec, int variant, hipStream_t st) {
  if (act == LPG_ACT_RELU) return launch_wres<T, NT, PG, LPG_ACT_RELU>(x, ldx, w, w_is_km, bias, y, ldy, n, k, m, y_vec, variant, st);
  if (act == LPG_ACT_TANH) return launch_wres<T, NT, PG, LPG_ACT_TANH>(x, ldx, w, w_is_km, bias, y, ldy, n, k, m, y_vec, variant, st);
  return launch_wres<T, NT, PG, LPG_ACT_NONE>(x, ldx, w, w_is_km, bias, y, ldy, n, k, m, y_vec, variant, st);
}

template <typename T, int PG>
int launch_nt(int nt, const T* x, int64_t ldx, const T* w, int w_is_km, const float* bias, int act, T* y, int64_t ldy,
              int64_t n, int k, int m, int y_vec, int variant, hipStream_t st) {
  if (nt == 1) return launch_act<T, 1, PG>(x, ldx, w, w_is_km, bias, act, y, ldy, n, k, m, y_vec, variant, st);
  if (nt == 2) return launch_act<T, 2, PG>(x, ldx, w, w_is_km, bias, act, y, ldy, n, k, m, y_vec, variant, st);
  return launch_act<T, 4, PG>(x, ldx, w, w_is_km, bias, act, y, ldy, n, k, m, y_vec, variant, st);
}

template <typename T>
int dispatch_wres(const T* x, int64_t ldx, const T* w, int w_is_km, const float* bias, int act, T* y, int64_t ldy,
                  int64_t n, int64_t k, int64_t m, hipStream_t st) {
  constexpr int EPB = Elt<T>::EPB, GK = Elt<T>::GK;
  int variant = static_cast<int>(options().linear_wres);   // -1 automatic; 0 off; 1 / 2 / 3 force that variant
  const bool forced = variant > 0;
  const int nt = m <= 32 ? 1 : (m <= 64 ? 2 : 4);
  const bool wide = sizeof(T) == 4 && k > 16 * GK && k <= 20 * GK && nt <= 2;   // SEAL's [onehot || 128-d] rows
  // persistent W-resident kernels pay off once every CU has several row tiles: from 16 K rows for narrow outputs, from
  // 48 K for fp32 128-wide ones (k = m = 128: 25 vs 17 us at n = 16 K, 26 vs 18 at 32 K, equal at 50 K;
  // profiles/r01/linear_small_n.log)
  const int64_t n_min = sizeof(T) == 4 && nt == 4 ? 49152 : 16384;
  if (variant == 0 || (n < n_min && !forced) || n < 1 || (k > 16 * GK && !wide) || m > 65535 * 32) return 0;
  auto al16 = [](const void* p) { return reinterpret_cast<uintptr_t>(p) % 16 == 0; };
  // X is read as whole 16-byte pieces of a row: rows must start 16-byte aligned and the last piece of a
  // row must lie inside the allocation (elements >= k are masked in registers, never used)
  if (ldx % EPB != 0 || !al16(x)) return 0;
  if (k % EPB != 0 && align_up(k, EPB) > ldx) return 0;    // overlapping row windows (ldx < k) with a ragged k
  const int pg = k <= 4 * GK ? 4 : (k <= 8 * GK ? 8 : (k <= 16 * GK ? 16 : 20));
  if (variant < 0) {
    // measured on MI355X (profiles/r01/linear_ab.log): the MFMA-bound fp32 shapes and k > 128 bf16 want the
    // staging-free variants 2 / 3 (3 = LDS-staged stores: fp32 128x128 1.60 -> 1.55 ms, bf16 256 -> 128 0.83 -> 0.74); the HBM-bound rest wants the coalesced variant 1; bf16 [k = 128] -> [m <= 32] is
    // the one shape where the tiled kernel is still ahead (0.28 vs 0.32 ms at n = 5 M)
    if (sizeof(T) == 4) variant = nt == 4 ? (pg == 16 ? 3 : 2) : 1;
    else variant = pg == 16 ? (nt == 4 ? 3 : 2) : 1;
    if (sizeof(T) == 2 && nt == 1 && pg == 8 && k % 8 == 0 && m % 8 == 0 && ldy % 8 == 0) return 0;
  }
  const int y_vec = ldy % EPB == 0 && al16(y);
  const int ki = static_cast<int>(k), mi = static_cast<int>(m);
  if (pg == 4) return launch_nt<T, 4>(nt, x, ldx, w, w_is_km, bias, act, y, ldy, n, ki, mi, y_vec, variant, st);
  if (pg == 8) return launch_nt<T, 8>(nt, x, ldx, w, w_is_km, bias, act, y, ldy, n, ki, mi, y_vec, variant, st);
  if (pg == 16) return launch_nt<T, 16>(nt, x, ldx, w, w_is_km, bias, act, y, ldy, n, ki, mi, y_vec, variant, st);
  if constexpr (sizeof(T) == 4) {
    if (nt == 1) return launch_act<T, 1, 20>(x, ldx, w, w_is_km, bias, act, y, ldy, n, ki, mi, y_vec, 2, st);
    return launch_act<T, 2, 20>(x, ldx, w, w_is_km, bias, act, y, ldy, n, ki, mi, y_vec, 2, st);
  }
  return 0;
}
}  // namespace

namespace lpg {
int linear_wres_f32(const float* x, int64_t ldx, const float* w, int w_is_km, const float* bias, int act, float* y,
                    int64_t ldy, int64_t n, int64_t k, int64_t m, hipStream_t st) {
  const int rc = dispatch_wres<float>(x, ldx, w, w_is_km, bias, act, y, ldy, n, k, m, st);
  if (rc == 1) ++options_mut().stat_linear_wres;
  return rc;
}
int linear_wres_bf16(const uint16_t* x, int64_t ldx, const uint16_t* w, int w_is_km, const float* bias, int act,
                     uint16_t* y, int64_t ldy, int64_t n, int64_t k, int64_t m, hipStream_t st) {
  const int rc = dispatch_wres<uint16_t>(x, ldx, w, w_is_km, bias, act, y, ldy, n, k, m, st);
  if (rc == 1) ++options_mut().stat_linear_wres;
  return rc;
}
}  // namespace lpg

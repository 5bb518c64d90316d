// Dense feature linear for fp32 storage on the bf16 matrix cores (K5, the MFMA-bound shapes):
//   Y[n,m] = act(X[n,k] @ W^T + bias),  X / W / Y float32.
// Reference anchor: GCNConv.lin of TwoWL/model/model.py:37 ([3P] PyG Linear, bias=False) and the dense layer of the
// SEAL head (SEAL/model/gnn.py:118-121); same contract as lpg_linear_f32 (linear.hip), which tries this path first.
//
// Why: v_mfma_f32_32x32x2_f32 peaks at 157 TF, so the fp32 linear at k = m = 128 is matrix-core bound (164 GF at
// n = 5 M: 1.04 ms at peak, 1.55-1.66 ms measured) although its 5.1 GB of traffic would take 0.9 ms.  The bf16
// instruction is 16x faster.  An fp32 number is EXACTLY the sum of three bf16 numbers (8 + 8 + 8 significant bits,
// split3 below), so  x w = sum over plane pairs;  the six pairs of weight >= 2^-16 (hi*hi, hi*mid, mid*hi, hi*lo,
// lo*hi, mid*mid) are kept, the three dropped pairs are <= 2^-24 |x| |w| each -- one fp32 rounding of the same
// product (|error| <= 2^-22 sum |x| |w| before accumulation, 40x inside the 1e-5 sum |x w| bound the tests hold).
// Sums run in the MFMA's fp32 accumulator as before.  6 MFMAs of 8 passes replace 8 of 16 passes: 2.7x less
// matrix-core time, and the kernel becomes HBM bound.
//
// Layout: a persistent workgroup of 8 waves keeps the three bf16 planes of a strip of W (NT x 32 output columns, all
// of k) in LDS: 3 x 128 x 128 x 2 B = 96 KB at k = m = 128.  A wave owns 32 rows of X at a time (two 16-row
// tiles of v_mfma_f32_16x16x32_bf16, which share every W fragment) and reads them straight from global memory in the
// B-fragment layout of that instruction: lane (c, g) takes 8 consecutive k = 32 bytes of row c, the four g of a row
// one whole 128-byte line, so a pair of load instructions consumes 16 full lines and nothing has to stay in the 32 KB
// L1 (the 32x32x16 layout -- 32 rows x 64 B per pair -- left 768 half-used lines in flight per CU and stopped at
// 4.0 TB/s).  Loads run four 32-wide k groups ahead of the multiply (16 KB in flight per wave), the split happens in
// registers, and the product is the TRANSPOSED one,  Y^T tile = W_strip X_tile^T:  the accumulator then holds, per
// lane, 4 consecutive output columns of ONE row, so y is written as 16-byte pieces, 64 contiguous bytes per row per
// instruction, without a trip through LDS (there is no LDS left for one).
// Strips (k m too large for one): workgroup v sits on XCD v % 8; the strips of one 256-row block run on the same XCD
// at the same time so that X is fetched from HBM once and re-read from that XCD's L2.
#include <algorithm>

#include "lpg_common.hpp"

namespace {
using namespace lpg;

using f32x4 = __attribute__((ext_vector_type(4))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) short;

constexpr int kBlockX3 = 512;
constexpr int kWavesX3 = kBlockX3 / 64;
constexpr size_t kLdsBudgetX3 = 148 * 1024;
constexpr int kMaxDevX3 = 16;
constexpr int kAheadX3 = 4;            // 32-wide k groups a wave's loads run ahead of its MFMAs (16 KB in flight per wave)

__device__ __forceinline__ void split3u(float v, uint32_t& hi, uint32_t& mid, uint32_t& lo) {   // the three terms as fp32 bit patterns with 16 zero low bits
  hi = __float_as_uint(v) & 0xFFFF0000u;
  const float r1 = v - __uint_as_float(hi);
  mid = __float_as_uint(r1) & 0xFFFF0000u;
  lo = __float_as_uint(r1 - __uint_as_float(mid));   // <= 8 significant bits: already a bf16 number
}

struct Frag3 {
  bf16x8 hi, mid, lo;
};

// 8 consecutive floats (two 16-byte pieces) -> the three bf16 fragments
__device__ __forceinline__ Frag3 split_frag(const float4& a, const float4& b) {
  const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
  union U {
    bf16x8 f;
    uint32_t u[4];
  } H, M, L;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    uint32_t h0, m0, l0, h1, m1, l1;
    split3u(v[2 * e], h0, m0, l0);
    split3u(v[2 * e + 1], h1, m1, l1);
    H.u[e] = (h0 >> 16) | h1;
    M.u[e] = (m0 >> 16) | m1;
    L.u[e] = (l0 >> 16) | (l1 & 0xFFFF0000u);
  }
  return {H.f, M.f, L.f};
}

template <int NT>
__global__ __launch_bounds__(kBlockX3) void linear_x3_kernel(const float* __restrict__ x, int64_t ldx,
                                                             const float* __restrict__ w, int w_is_km,
                                                             const float* __restrict__ bias, int act,
                                                             float* __restrict__ y, int64_t ldy, int64_t n, int k, int m,
                                                             int kpad, int strips, int64_t n_tiles, int w_vec) {
  extern __shared__ __attribute__((aligned(16))) unsigned char x3_smem[];
  // a plane is [k / 8][NT * 32 columns][8 k]: the fragment of lane (c, g) -- column c of a 16-wide tile, 8 k -- sits in
  // 16-byte slot c of its 256-byte bank row whatever g is, and each 16-lane group of a ds_read_b128 holds every c once
  // (conflict-free; [column][k] rows padded by 16 B put lanes 12 and 27 on one slot: 6e7 conflict cycles per launch)
  constexpr int NCOL = NT * 32;
  const int plane = kpad * NCOL;                          // bf16 elements per plane
  uint16_t* Wp = reinterpret_cast<uint16_t*>(x3_smem);    // [3][kpad / 8][NCOL][8]
  float* Bs = reinterpret_cast<float*>(Wp + static_cast<size_t>(3) * plane);   // [NCOL]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, g = lane >> 4;
  constexpr int NC = NT * 2;                              // 16-wide column tiles of the strip
  const int v = blockIdx.x, G = gridDim.x;
  int strip, q;
  const int Q = G / strips;                               // workgroups per strip (the host makes G a multiple of strips)
  if (G % (8 * strips) == 0) {                            // the strips of one row block on one XCD, side by side
    const int xcd = v & 7, slot = v >> 3, per = G / (8 * strips);
    strip = slot % strips;
    q = slot / strips + per * xcd;
  } else {
    strip = v % strips;
    q = v / strips;
  }
  const int j0 = strip * NT * 32;

  // ---- once per workgroup: the strip of W as three bf16 planes, zero beyond k / m
  // (16-byte reads of W: k % 4 == 0 and m % 4 == 0 hold here; w_vec = the pointer is 16-byte aligned as well)
  auto put3 = [&](int j, int kk, float val) {
    uint32_t ph, pm, pl;
    split3u(val, ph, pm, pl);
    const int at = ((kk >> 3) * NCOL + j) * 8 + (kk & 7);
    Wp[at] = static_cast<uint16_t>(ph >> 16);
    Wp[plane + at] = static_cast<uint16_t>(pm >> 16);
    Wp[2 * plane + at] = static_cast<uint16_t>(pl >> 16);
  };
  if (w_vec && !w_is_km) {                                // W [m][k]: 4 consecutive k of one column = 8 bytes of each plane
    const int k4 = kpad >> 2;
#pragma unroll 4
    for (int idx = tid; idx < NCOL * k4; idx += kBlockX3) {
      const int j = idx / k4, kk = (idx - j * k4) * 4;
      float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
      if (j0 + j < m && kk < k) val = *reinterpret_cast<const float4*>(w + static_cast<int64_t>(j0 + j) * k + kk);
      const float e[4] = {val.x, val.y, val.z, val.w};
      uint32_t ph[4], pm[4], pl[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) split3u(e[i], ph[i], pm[i], pl[i]);
      const int at = ((kk >> 3) * NCOL + j) * 8 + (kk & 7);
      *reinterpret_cast<uint2*>(Wp + at) = make_uint2((ph[0] >> 16) | ph[1], (ph[2] >> 16) | ph[3]);
      *reinterpret_cast<uint2*>(Wp + plane + at) = make_uint2((pm[0] >> 16) | pm[1], (pm[2] >> 16) | pm[3]);
      *reinterpret_cast<uint2*>(Wp + 2 * plane + at) =
          make_uint2((pl[0] >> 16) | (pl[1] & 0xFFFF0000u), (pl[2] >> 16) | (pl[3] & 0xFFFF0000u));
    }
  } else if (w_vec) {                                     // W [k][m]: 4 consecutive columns of one k
    constexpr int J4 = NCOL / 4;
#pragma unroll 4
    for (int idx = tid; idx < J4 * kpad; idx += kBlockX3) {
      const int kk = idx / J4, j = (idx - kk * J4) * 4;
      float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
      if (j0 + j < m && kk < k) val = *reinterpret_cast<const float4*>(w + static_cast<int64_t>(kk) * m + j0 + j);
      put3(j, kk, val.x);
      put3(j + 1, kk, val.y);
      put3(j + 2, kk, val.z);
      put3(j + 3, kk, val.w);
    }
  } else {
    for (int idx = tid; idx < NCOL * kpad; idx += kBlockX3) {
      int j, kk;
      if (w_is_km) {
        j = idx % NCOL;
        kk = idx / NCOL;
      } else {
        kk = idx % kpad;
        j = idx / kpad;
      }
      float val = 0.f;
      if (j0 + j < m && kk < k) val = w_is_km ? w[static_cast<int64_t>(kk) * m + j0 + j] : w[static_cast<int64_t>(j0 + j) * k + kk];
      put3(j, kk, val);
    }
  }
  for (int j = tid; j < NT * 32; j += kBlockX3) Bs[j] = (bias && j0 + j < m) ? bias[j0 + j] : 0.f;
  __syncthreads();

  const int groups = kpad >> 5;                           // 32 k values per group = one MFMA step
  const int64_t t_first = static_cast<int64_t>(q) * kWavesX3 + wave, t_stride = static_cast<int64_t>(Q) * kWavesX3;
  if (t_first >= n_tiles) return;                         // no barrier below
  const int64_t my_tiles = (n_tiles - 1 - t_first) / t_stride + 1;
  const int64_t total = my_tiles * groups;

  // the load cursor runs kAheadX3 groups ahead of the multiply cursor.  Lane (c, g) reads 32 B of row c: the four g of a
  // row cover one whole 128-byte line per pair of instructions, 16 full lines per wave instruction pair.
  int64_t l_tile = t_first;
  int l_g = 0;
  int64_t l_it = 0;
  auto load = [&](float4 (&b)[4]) {
    if (l_it < total) {                                   // wave-uniform
      const int col = l_g * 32 + g * 8;
      // unconditional loads (a select behind a load makes the compiler wait for it on the spot): pieces beyond k are
      // read from the row's start instead and zeroed when the group is multiplied
      const int c0 = col < k ? col : 0, c1 = col + 4 < k ? col + 4 : 0;
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) {
        const int64_t row = min(l_tile * 32 + rt * 16 + c, n - 1);
        const float* p = x + row * ldx;
        b[rt * 2] = *reinterpret_cast<const float4*>(p + c0);
        b[rt * 2 + 1] = *reinterpret_cast<const float4*>(p + c1);
      }
      if (++l_g == groups) {
        l_g = 0;
        l_tile += t_stride;
      }
    }
    ++l_it;
  };
  const bool ragged = (k & 31) != 0;                      // k % 4 == 0: a 16-byte piece is all in or all out

  f32x4 acc[2][NC];
  int64_t p_tile = t_first;
  int p_g = 0;
  auto process = [&](const float4 (&b)[4]) {
    if (p_g == 0) {
#pragma unroll
      for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int ct = 0; ct < NC; ++ct)
#pragma unroll
          for (int i = 0; i < 4; ++i) acc[rt][ct][i] = 0.f;
    }
    float4 q0 = b[0], q1 = b[1], q2 = b[2], q3 = b[3];
    if (ragged && p_g == groups - 1) {
      const int col = p_g * 32 + g * 8;
      const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
      if (col >= k) q0 = q2 = z;
      if (col + 4 >= k) q1 = q3 = z;
    }
    const Frag3 x0 = split_frag(q0, q1), x1 = split_frag(q2, q3);
    const uint16_t* wbase = Wp + ((p_g * 4 + g) * NCOL + c) * 8;
    // the W fragments of column tile ct + 1 are read while tile ct is multiplied (a wave has one partner on its SIMD
    // to hide LDS latency behind, not three)
    bf16x8 wh[2], wm[2], wl[2];
    wh[0] = *reinterpret_cast<const bf16x8*>(wbase);
    wm[0] = *reinterpret_cast<const bf16x8*>(wbase + plane);
    wl[0] = *reinterpret_cast<const bf16x8*>(wbase + 2 * plane);
#pragma unroll
    for (int ct = 0; ct < NC; ++ct) {
      const int cur = ct & 1, nxt = cur ^ 1;
      if (ct + 1 < NC) {
        const uint16_t* wrow = wbase + (ct + 1) * 16 * 8;
        wh[nxt] = *reinterpret_cast<const bf16x8*>(wrow);
        wm[nxt] = *reinterpret_cast<const bf16x8*>(wrow + plane);
        wl[nxt] = *reinterpret_cast<const bf16x8*>(wrow + 2 * plane);
      }
      __builtin_amdgcn_sched_barrier(0);                  // keep the three reads above ahead of this tile's MFMAs
      // small terms first; A = W (rows = output columns), B = X (columns = rows of X); the two row tiles alternate
      acc[0][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[cur], x0.hi, acc[0][ct], 0, 0, 0);
      acc[1][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[cur], x1.hi, acc[1][ct], 0, 0, 0);
      acc[0][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[cur], x0.lo, acc[0][ct], 0, 0, 0);
      acc[1][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[cur], x1.lo, acc[1][ct], 0, 0, 0);
      acc[0][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm[cur], x0.mid, acc[0][ct], 0, 0, 0);
      acc[1][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm[cur], x1.mid, acc[1][ct], 0, 0, 0);
      acc[0][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm[cur], x0.hi, acc[0][ct], 0, 0, 0);
      acc[1][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm[cur], x1.hi, acc[1][ct], 0, 0, 0);
      acc[0][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[cur], x0.mid, acc[0][ct], 0, 0, 0);
      acc[1][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[cur], x1.mid, acc[1][ct], 0, 0, 0);
      acc[0][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[cur], x0.hi, acc[0][ct], 0, 0, 0);
      acc[1][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[cur], x1.hi, acc[1][ct], 0, 0, 0);
    }
    if (++p_g == groups) {
      // lane (c, g): row c of a row tile; the 4 registers of column tile ct = output columns ct*16 + 4 g .. + 3, so
      // a store instruction writes 64 contiguous bytes of each of 16 rows
      auto put = [&](auto fn) {
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
          const int64_t row = p_tile * 32 + rt * 16 + c;
          if (row < n) {
            float* yr = y + row * ldy + j0 + 4 * g;
#pragma unroll
            for (int ct = 0; ct < NC; ++ct)
              if (j0 + ct * 16 + 4 * g < m) {             // m % 4 == 0: a piece is all in or all out
                const float4 bb = *reinterpret_cast<const float4*>(Bs + ct * 16 + 4 * g);
                float4 o;
                o.x = fn(acc[rt][ct][0] + bb.x);
                o.y = fn(acc[rt][ct][1] + bb.y);
                o.z = fn(acc[rt][ct][2] + bb.z);
                o.w = fn(acc[rt][ct][3] + bb.w);
                *reinterpret_cast<float4*>(yr + ct * 16) = o;
              }
          }
        }
      };
      if (act == LPG_ACT_NONE) put([](float a) { return a; });
      else if (act == LPG_ACT_RELU) put([](float a) { return fmaxf(a, 0.f); });
      else put([act](float a) { return apply_act(a, act); });
      p_g = 0;
      p_tile += t_stride;
    }
  };

  float4 buf[kAheadX3 + 1][4];
#pragma unroll
  for (int i = 0; i < kAheadX3; ++i) load(buf[i]);
  for (int64_t it = 0; it < total; it += kAheadX3 + 1) {
#pragma unroll
    for (int i = 0; i <= kAheadX3; ++i)                   // buffer i is multiplied while buffer i - 1 (mod) is refilled
      if (it + i < total) {
        load(buf[(i + kAheadX3) % (kAheadX3 + 1)]);
        process(buf[i]);
      }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// The tiled form (r03), for k % 32 == 0, m % 128 == 0 and W stored [m][k].
//
// What bounded the kernel above at 256 -> 256 (1.72 ms for 2 M rows = 0.30 of HBM; MFMA work at peak: 0.63 ms) is VALU
// ISSUE, not the matrix cores: a wave splits every X fragment it multiplies itself, W is cut into FOUR strips of 64
// columns at k = 256 (three planes of a wider strip do not fit the LDS), so every element of X is split four times
// and a 32-k group costs ~200 VALU instructions (ISA count: ~110 of split + 40 of moves around the ragged-k select +
// addressing) beside its 48 MFMAs of 16 cycles: 800 issue cycles against 768 matrix cycles, with two waves per SIMD.
//
// Here a 256-thread workgroup owns a 128 x 128 output tile and walks k in chunks of 32: the chunk of X (128 rows) and of W
// (128 columns) is loaded to registers one chunk ahead, split ONCE by the thread that loaded it (9 instructions per
// pair of values: v_perm packs the high halves, v_pk_add_f32 forms both remainders) and stored as three bf16 planes
// in LDS in the fragment order of v_mfma_f32_16x16x32_bf16 (48 KB per workgroup, three workgroups per CU, whose load /
// split / multiply phases interleave on the SIMDs); each of the four waves multiplies a 64 x 64 sub-tile: 24 LDS
// fragment reads and 96 MFMAs per chunk, and ~150 VALU instructions of split per thread per chunk against those 96
// MFMAs -- a quarter of the ratio above.  X is read from HBM once (the strips of a row tile run side by side on one
// XCD and share it through that XCD's L2), W (256 KB) stays in L2.
//
// Measured (profiles/r03/linear_x3_tiled.log, same box, 2 M rows 256 -> 256 / 5 M rows 128 -> 128): strip form 1.60 / 1.19 ms,
// this form 1.55 / 1.19 ms, and 0.018 against 0.029 ms at 50 K rows (no W staging per workgroup).  Three designs -- the
// strip form, this one, and this one on v_mfma_f32_32x32x16_bf16 (half the MFMA issues: 1.70 / 1.32 ms, slower: the chip
// holds a lower clock on that shape, MI355X guide 'DVFS give-back' (7)) -- all stop at ~1.0 PF of bf16 MFMA on random
// data: the six-product sum is bound by the matrix cores at the clock they sustain under load (the guide's best bf16
// GEMM body: 1.25 PF), i.e. 1.26 ms for this shape = 0.41 of the HBM time at best.  A persistent grid with cross-tile
// prefetch was slower (1.84 ms at two workgroups per CU; three do not fit without scratch).
// Counters of this form (profiles/r03/pmc_x3t_summary.txt): clock 1.96 GHz (GRBM_GUI_ACTIVE / 8 / wall), matrix cores busy
// 48 % of those cycles (SQ_VALU_MFMA_BUSY_CYCLES = 96 M MFMAs x 16), no LDS bank conflicts; with X confined to an
// 8 MB footprint (every row tile reading the same 64 tiles: a measurement build) 1.42 / 1.04 ms -- memory latency is
// worth 10-15 %, a second chunk of prefetch does not fit 168 registers.
// Also built and measured, not kept: a "ping-pong" form -- one persistent 512-thread workgroup per CU on 256 x 128 tiles,
// two chunk buffers in LDS (144 KB), the two waves of a SIMD splitting the next chunk and multiplying the current one in
// opposite order, one barrier per chunk, the fetch two chunks ahead (254 registers): 1.85 ms against this form's 1.55
// (matrix cores busy 35 %, vector ALU 32 %, i.e. the two jobs did NOT overlap, whichever pair of waves was given the
// opposite order); three independent workgroups per CU interleave better than two hand-paired waves.
constexpr int kTileR = 128, kTileC = 128;
constexpr int kBlockT = 256;
using f32x2 = __attribute__((ext_vector_type(2))) float;

// two values -> their three bf16 terms, packed (first value in the low half)
__device__ __forceinline__ void split_pair(float a, float b, uint32_t& H, uint32_t& M, uint32_t& L) {
  constexpr uint32_t kHiHalves = 0x07060302u;             // v_perm_b32: (bytes 3:2 of src0) : (bytes 3:2 of src1)
  const uint32_t a0 = __float_as_uint(a), b0 = __float_as_uint(b);
  H = __builtin_amdgcn_perm(b0, a0, kHiHalves);
  const f32x2 v = {a, b};
  const f32x2 t = {__uint_as_float(a0 & 0xFFFF0000u), __uint_as_float(b0 & 0xFFFF0000u)};
  const f32x2 r = v - t;
  const uint32_t a1 = __float_as_uint(r.x), b1 = __float_as_uint(r.y);
  M = __builtin_amdgcn_perm(b1, a1, kHiHalves);
  const f32x2 u = {__uint_as_float(a1 & 0xFFFF0000u), __uint_as_float(b1 & 0xFFFF0000u)};
  const f32x2 q = r - u;                                  // <= 8 significant bits each: already bf16 numbers
  L = __builtin_amdgcn_perm(__float_as_uint(q.y), __float_as_uint(q.x), kHiHalves);
}

__device__ __forceinline__ void split_octet(const float4& a, const float4& b, uint4& H, uint4& M, uint4& L) {
  split_pair(a.x, a.y, H.x, M.x, L.x);
  split_pair(a.z, a.w, H.y, M.y, L.y);
  split_pair(b.x, b.y, H.z, M.z, L.z);
  split_pair(b.z, b.w, H.w, M.w, L.w);
}

__global__ __launch_bounds__(kBlockT, 3) void linear_x3t_kernel(const float* __restrict__ x, int64_t ldx,
                                                                const float* __restrict__ w,
                                                                const float* __restrict__ bias, int act,
                                                                float* __restrict__ y, int64_t ldy, int64_t n, int k,
                                                                int strips, int64_t n_rtiles) {
  // a plane of a chunk is [4 k octets][128 rows or columns][8 k] bf16: lane (c, g) of a fragment read takes the 16-byte
  // slot (g, base + c); the 16 lanes of a group read 256 contiguous bytes
  __shared__ uint4 xs[3][4][kTileR];
  __shared__ uint4 ws[3][4][kTileC];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, g = lane >> 4;
  const int wr = wave >> 1, wc = wave & 1;                // the wave's 64 x 64 quarter of the tile
  // workgroup v sits on XCD v % 8: the strips of one row tile are neighbours in that XCD's dispatch order.  (A
  // persistent grid that prefetches the next tile's first chunk during the last chunk of this one measured SLOWER:
  // 1.84 ms against 1.71 at two workgroups per CU, and does not fit three per CU without scratch.)
  const int v = blockIdx.x, xcd = v & 7, slot = v >> 3;
  const int strip = slot % strips;
  const int64_t rtile = static_cast<int64_t>(slot / strips) * 8 + xcd;
  if (rtile >= n_rtiles) return;                          // whole workgroup: no barrier is skipped
  const int64_t r0 = rtile * kTileR;
  const int j0 = strip * kTileC;

  // loader role: thread (lr, lo) takes k octet lo of rows lr and lr + 64 of X and of columns lr and lr + 64 of W.  A wave
  // covers 16 rows: its four 16-lane groups are the four octets of the same rows, so one load instruction consumes 16
  // whole 128-byte lines, and each 16-lane group of a ds_write_b128 fills 16 CONSECUTIVE slots of one octet row
  // (lo = tid & 3 put the four octets of a row, 2 KB apart, on the same banks: SQ_LDS_BANK_CONFLICT 2.9e8 cycles per
  // launch, a third of the kernel)
  const int lr = (wave << 4) | (lane & 15), lo = lane >> 4;
  const float* px0 = x + min(r0 + lr, n - 1) * ldx + lo * 8;
  const float* px1 = x + min(r0 + lr + 64, n - 1) * ldx + lo * 8;
  const float* pw0 = w + static_cast<int64_t>(j0 + lr) * k + lo * 8;
  const float* pw1 = w + static_cast<int64_t>(j0 + lr + 64) * k + lo * 8;
  float4 bx[4], bw[4];
  auto fetch = [&](int kc) {
    const int o = kc * 32;
    bx[0] = *reinterpret_cast<const float4*>(px0 + o);
    bx[1] = *reinterpret_cast<const float4*>(px0 + o + 4);
    bx[2] = *reinterpret_cast<const float4*>(px1 + o);
    bx[3] = *reinterpret_cast<const float4*>(px1 + o + 4);
    bw[0] = *reinterpret_cast<const float4*>(pw0 + o);
    bw[1] = *reinterpret_cast<const float4*>(pw0 + o + 4);
    bw[2] = *reinterpret_cast<const float4*>(pw1 + o);
    bw[3] = *reinterpret_cast<const float4*>(pw1 + o + 4);
  };
  auto stage = [&]() {
    uint4 H, M, L;
    split_octet(bx[0], bx[1], H, M, L);
    xs[0][lo][lr] = H, xs[1][lo][lr] = M, xs[2][lo][lr] = L;
    split_octet(bx[2], bx[3], H, M, L);
    xs[0][lo][lr + 64] = H, xs[1][lo][lr + 64] = M, xs[2][lo][lr + 64] = L;
    split_octet(bw[0], bw[1], H, M, L);
    ws[0][lo][lr] = H, ws[1][lo][lr] = M, ws[2][lo][lr] = L;
    split_octet(bw[2], bw[3], H, M, L);
    ws[0][lo][lr + 64] = H, ws[1][lo][lr + 64] = M, ws[2][lo][lr + 64] = L;
  };

  const int chunks = k >> 5;
  f32x4 acc[4][4];                                        // [row tile of 16][column tile of 16]
#pragma unroll
  for (int rt = 0; rt < 4; ++rt)
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[rt][ct][i] = 0.f;

  fetch(0);
  for (int kc = 0; kc < chunks; ++kc) {
    if (kc) __syncthreads();                              // every wave is done with the previous chunk's planes
    stage();
    __syncthreads();
    if (kc + 1 < chunks) fetch(kc + 1);                   // in flight while this chunk is multiplied
    bf16x8 wf[3][4];
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) wf[p][ct] = __builtin_bit_cast(bf16x8, ws[p][g][wc * 64 + ct * 16 + c]);
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) {
      const bf16x8 xh = __builtin_bit_cast(bf16x8, xs[0][g][wr * 64 + rt * 16 + c]);
      const bf16x8 xm = __builtin_bit_cast(bf16x8, xs[1][g][wr * 64 + rt * 16 + c]);
      const bf16x8 xl = __builtin_bit_cast(bf16x8, xs[2][g][wr * 64 + rt * 16 + c]);
      // small terms first; A = W (rows = output columns), B = X (columns = rows of X); consecutive MFMAs go to
      // different accumulators
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[2][ct], xh, acc[rt][ct], 0, 0, 0);
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[0][ct], xl, acc[rt][ct], 0, 0, 0);
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[1][ct], xm, acc[rt][ct], 0, 0, 0);
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[1][ct], xh, acc[rt][ct], 0, 0, 0);
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[0][ct], xm, acc[rt][ct], 0, 0, 0);
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[0][ct], xh, acc[rt][ct], 0, 0, 0);
    }
  }

  // lane (c, g): row c of a row tile; the 4 registers of column tile ct = output columns ct * 16 + 4 g .. + 3
  auto put = [&](auto fn) {
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) {
      const int64_t row = r0 + wr * 64 + rt * 16 + c;
      if (row < n) {
        const int col = j0 + wc * 64 + 4 * g;
        float* yr = y + row * ldy + col;
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
          float4 bb = make_float4(0.f, 0.f, 0.f, 0.f);
          if (bias) bb = *reinterpret_cast<const float4*>(bias + col + ct * 16);
          float4 o;
          o.x = fn(acc[rt][ct][0] + bb.x);
          o.y = fn(acc[rt][ct][1] + bb.y);
          o.z = fn(acc[rt][ct][2] + bb.z);
          o.w = fn(acc[rt][ct][3] + bb.w);
          *reinterpret_cast<float4*>(yr + ct * 16) = o;
        }
      }
    }
  };
  if (act == LPG_ACT_NONE) put([](float a) { return a; });
  else if (act == LPG_ACT_RELU) put([](float a) { return fmaxf(a, 0.f); });
  else put([act](float a) { return apply_act(a, act); });
}

int launch_x3t(const float* x, int64_t ldx, const float* w, const float* bias, int act, float* y, int64_t ldy, int64_t n, int k,
               int m, int cus, hipStream_t st) {
  const int strips = m / kTileC;
  const int64_t n_rtiles = ceil_div(n, kTileR);
  (void)cus;
  const int64_t g = align_up(n_rtiles, 8) * strips;       // row tile = 8 * (slot / strips) + XCD: whole groups of 8
  if (g > 0x7fffffff) return 0;
  hipLaunchKernelGGL(linear_x3t_kernel, dim3(static_cast<unsigned>(g)), dim3(kBlockT), 0, st, x, ldx, w, bias, act, y, ldy, n,
                     k, strips, n_rtiles);
  return 1;
}

template <int NT>
int launch_x3(const float* x, int64_t ldx, const float* w, int w_is_km, const float* bias, int act, float* y, int64_t ldy,
              int64_t n, int k, int m, int kpad, int cus, int dev, hipStream_t st) {
  const size_t lds = static_cast<size_t>(3) * NT * 32 * kpad * 2 + NT * 32 * 4;
  auto kern = linear_x3_kernel<NT>;
  static bool attr_set[kMaxDevX3] = {};                   // per instantiation and device (a code object is loaded per device)
  if (!attr_set[dev]) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                            160 * 1024) != hipSuccess) {
      set_error("linear (bf16 planes): cannot raise the dynamic LDS limit");
      return LPG_E_LAUNCH;
    }
    attr_set[dev] = true;
  }
  const int strips = static_cast<int>(ceil_div(m, NT * 32));
  const int64_t n_tiles = ceil_div(n, 32);
  // one workgroup per CU (LDS bound); a multiple of 8 * strips keeps the strips of a row block on one XCD
  int64_t per_strip = std::max<int64_t>(1, cus / strips);
  per_strip = std::min<int64_t>(per_strip, ceil_div(n_tiles, kWavesX3));
  int64_t g = per_strip * strips;
  if (g >= 8 * strips) g -= g % (8 * strips);
  hipLaunchKernelGGL(kern, dim3(static_cast<unsigned>(g)), dim3(kBlockX3), lds, st, x, ldx, w, w_is_km, bias, act, y, ldy,
                     n, k, m, kpad, strips, n_tiles, reinterpret_cast<uintptr_t>(w) % 16 == 0 ? 1 : 0);
  return 1;
}
}  // namespace

namespace lpg {
// return 1 = launched, 0 = shape does not qualify (the caller goes on to the fp32-MFMA kernels), < 0 = error
int linear_x3_f32(const float* x, int64_t ldx, const float* w, int w_is_km, const float* bias, int act, float* y,
                  int64_t ldy, int64_t n, int64_t k, int64_t m, hipStream_t st) {
  const int64_t mode = options().linear_x3;               // -1 automatic; 0 off; 1 force the strip form wherever the layout allows; 2 the tiled form (or nothing)
  if (mode == 0 || n < 1) return 0;
  auto al16 = [](const void* p) { return reinterpret_cast<uintptr_t>(p) % 16 == 0; };
  if (k % 4 != 0 || m % 4 != 0 || ldx % 4 != 0 || ldy % 4 != 0 || !al16(x) || !al16(y) || k > 768 || ldx < k) return 0;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevX3) return 0;
  static int cus_of[kMaxDevX3] = {};
  if (!cus_of[dev] &&
      (hipDeviceGetAttribute(&cus_of[dev], hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus_of[dev] <= 0))
    cus_of[dev] = 256;
  const int cus = cus_of[dev];
  // the tiled form: k % 32 == 0, m % 128 == 0, W [m][k] 16-byte aligned, bias (if any) 16-byte aligned
  const bool tiled_ok = !w_is_km && k % 32 == 0 && m % kTileC == 0 && al16(w) && (!bias || al16(bias)) && k >= 32;
  // automatic: from 16 K rows; at k m <= 128 x 128 only up to 512 K rows (beyond, the strip form with all of W resident
  // is 3-5 % ahead: 5 M x 128 -> 128 1.24 against 1.30 ms, profiles/r03/linear_x3_tiled.log)
  const bool tiled_auto = mode < 0 && options().linear_wres <= 0 && n >= 16384 && (k * m > 128 * 128 || n <= (int64_t(1) << 19));
  if (tiled_ok && (mode == 2 || tiled_auto)) {
    const int rc = launch_x3t(x, ldx, w, bias, act, y, ldy, n, static_cast<int>(k), static_cast<int>(m), cus, st);
    if (rc == 1) ++options_mut().stat_linear_x3;
    if (rc != 0) return rc;
  }
  if (mode == 2) return 0;
  const int kpad = static_cast<int>(align_up(k, 32));
  int nt = m <= 32 ? 1 : (m <= 64 ? 2 : 4);
  auto lds_of = [&](int t) { return static_cast<size_t>(3) * t * 32 * kpad * 2 + t * 32 * 4; };
  while (nt > 1 && lds_of(nt) > kLdsBudgetX3) nt >>= 1;
  if (lds_of(nt) > kLdsBudgetX3) return 0;
  if (ceil_div(m, nt * 32) > 32) return 0;                // very wide outputs: the tiled kernels
  if (mode < 0) {
    if (options().linear_wres > 0) return 0;              // a W-resident variant is forced: that is what runs
    // the shapes the fp32 matrix cores bound (k m / (2 (k + m)) = flop per byte / 1: 128 x 128 -> 32, 128 x 64 -> 21;
    // at 64 x 64 and 128 x 32 both kernels run at the HBM rate), with enough rows to pay for staging the planes in
    // every workgroup: measured crossover 50-65 K rows at 128 x 128, below 16 K for larger k m
    // (profiles/r02/linear_x3.log)
    if ((k * m) / (2 * (k + m)) < 20) return 0;
    if (n < (k * m > 128 * 128 ? 16384 : 57344)) return 0;
  }
  const int ki = static_cast<int>(k), mi = static_cast<int>(m);
  int rc;
  if (nt == 1) rc = launch_x3<1>(x, ldx, w, w_is_km, bias, act, y, ldy, n, ki, mi, kpad, cus, dev, st);
  else if (nt == 2) rc = launch_x3<2>(x, ldx, w, w_is_km, bias, act, y, ldy, n, ki, mi, kpad, cus, dev, st);
  else rc = launch_x3<4>(x, ldx, w, w_is_km, bias, act, y, ldy, n, ki, mi, kpad, cus, dev, st);
  if (rc == 1) ++options_mut().stat_linear_x3;
  return rc;
}
}  // namespace lpg

// SEAL graph convolution, layers 1..4 (SEAL/model/gnn.py:46-63), second formulation of the dense-block stack: every
// matrix product of the stack on the bf16 matrix cores (v_mfma_f32_32x32x16_bf16), in two numeric classes
//
//   NP = 3  fp32 storage, fp32-class arithmetic.  An fp32 value is three bf16 terms (hi | mid | lo: 8 + 8 + 8 mantissa
//           bits, exact -- csrc/seal_conv.hip split3).  The aggregation (A + I) H has exact 0 / 1 / 2 entries on one side,
//           so its three plane products are exact and summed in the MFMA's fp32 accumulator.  The two dense products
//           of a layer, H' = Z W and a16 += Z k1, keep the six plane pairs whose weight is >= 2^-16 of the leading one
//           (hi*hi, hi*mid, mid*hi, hi*lo, lo*hi, mid*mid); the three dropped pairs are <= 2^-24 |z| |w| each, the
//           size of one fp32 rounding of the same product.  12 MFMAs of 8 passes per 32 x 32 x 32 product instead of
//           16 v_mfma_f32_32x32x2_f32 of 16 passes each (the r01 kernel, kept as option seal_conv_variant = 1).
//   NP = 1  bf16 storage (BASELINE configs[2]: "d = 128, bf16"): H and Z are rounded to bf16 (nearest even) wherever
//           the generic bf16 path stores them, weights are bf16, one plane everywhere: 2 MFMAs per aggregation block,
//           2 per dense product; the planes take 17 KB of LDS instead of 51.
//
// Layout of the products is the transposed one of seal_conv_t_kernel (lane = vertex, registers = features; the A
// operand of the aggregation is a fragment of H^T, the B operand 8 bits of the lane's own adjacency row through a byte
// table): see that kernel's header.  What is new besides the plane products:
//   * the H_1 planes are filled from COALESCED reads -- a lane takes two consecutive rows x four features (two 16-byte
//     loads of 64-byte row segments: 16 cache lines per wave instruction) and writes row-pair dwords into the planes
//     conflict-free; the r01 kernel walked one feature across rows (64 cache lines per instruction, 32 instructions
//     per thread: the texture pipe, not the matrix pipe, bounded its set-up phase);
//   * W_2, W_3 and the head's conv1d #1 slices are staged in LDS as ready-made bf16 A-operand fragments per plane.
#include <stdlib.h>

#include "lpg_common.hpp"

namespace {
using namespace lpg;

using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) short;

constexpr int MAXN = LPG_SEAL_MAXN;   // 128
constexpr int WORDS = MAXN / 32;
constexpr int WIN = 128;              // a workgroup owns the subgraphs that start in a 128-row window
constexpr int RMAX = WIN + MAXN;      // rows it can end up with (<= 255)
constexpr int kConvBlock = 256;       // 4 waves, up to two 32-row tiles each
constexpr int HS = 256 + 8;           // row stride of a transposed bf16 plane (528 B: 16-byte aligned; 132 dwords = 4 mod 16)

// tanh(v) = 1 - 2 / (exp(2 v) + 1): v_exp_f32 and v_rcp_f32 (1 ulp each; __frcp_rn expands to the ten-instruction IEEE division
// sequence, 16 times per tile and layer): |error| < 4e-7 absolute on an output in [-1, 1]; saturates cleanly
__device__ __forceinline__ float tanh_fast_div(float v) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(__expf(2.0f * v) + 1.0f); }
// the same with the row scale folded into the exponent: tanh(acc * s) = 1 - 2 / (2^(acc * s2) + 1), s2 = s * 2 log2(e)
__device__ __forceinline__ float tanh_scaled(float acc, float s2) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(acc * s2) + 1.0f); }
__device__ __forceinline__ int acc_row(int i, int h) { return (i & 3) + 8 * (i >> 2) + 4 * h; }

__device__ __forceinline__ float to_f32(float v) { return v; }
__device__ __forceinline__ float to_f32(uint16_t v) { return bf16_to_f32(v); }

// bf16 planes of a value: NP = 3 exact truncation split, NP = 1 round to nearest even
template <int NP>
__device__ __forceinline__ void planes_of(float v, uint32_t (&p)[NP]) {
  if constexpr (NP == 1) {
    p[0] = f32_to_bf16(v);
  } else {
    const uint32_t uh = __float_as_uint(v) & 0xFFFF0000u;
    const float r1 = v - __uint_as_float(uh);
    const uint32_t um = __float_as_uint(r1) & 0xFFFF0000u;
    const float r2 = r1 - __uint_as_float(um);
    p[0] = uh >> 16; p[1] = um >> 16; p[2] = __float_as_uint(r2) >> 16;
  }
}

// the planes of TWO values, packed (a in the low half of every dword): NP = 1 one v_cvt_pk_bf16_f32; NP = 3 the exact
// truncation split with each plane's two upper halves merged by ONE v_perm_b32 -- (and, sub, and, sub) per value + three
// perms per pair = 11 instructions, against 14 + 3 with a shift per plane and value and a shift-or per plane
template <int NP>
__device__ __forceinline__ void planes_pair(float a, float b, uint32_t (&pk)[NP]) {
  if constexpr (NP == 1) {
    pk[0] = pack_bf16x2(a, b);
  } else {
    const uint32_t ua = __float_as_uint(a), ub = __float_as_uint(b);
    const float r1a = a - __uint_as_float(ua & 0xFFFF0000u), r1b = b - __uint_as_float(ub & 0xFFFF0000u);
    const uint32_t ma = __float_as_uint(r1a), mb = __float_as_uint(r1b);
    const float r2a = r1a - __uint_as_float(ma & 0xFFFF0000u), r2b = r1b - __uint_as_float(mb & 0xFFFF0000u);
    pk[0] = __builtin_amdgcn_perm(ub, ua, 0x07060302u);              // (hi16(b) << 16) | hi16(a)
    pk[1] = __builtin_amdgcn_perm(mb, ma, 0x07060302u);
    pk[2] = __builtin_amdgcn_perm(__float_as_uint(r2b), __float_as_uint(r2a), 0x07060302u);
  }
}

// plane pairs (value plane, weight plane) of a dense product, smallest terms first
template <int NP> struct Terms;
template <> struct Terms<1> {
  static constexpr int N = 1;
  static constexpr int zp[1] = {0};
  static constexpr int wp[1] = {0};
};
template <> struct Terms<3> {
  static constexpr int N = 6;
  static constexpr int zp[6] = {1, 2, 0, 1, 0, 0};
  static constexpr int wp[6] = {1, 0, 2, 0, 1, 0};
};
// against a weight that always has three planes (the head's fp32 conv1d #1 kernel)
template <int NP> struct KTerms;
template <> struct KTerms<1> {
  static constexpr int N = 3;
  static constexpr int zp[3] = {0, 0, 0};
  static constexpr int wp[3] = {2, 1, 0};
};
template <> struct KTerms<3> : Terms<3> {};

// the two fusions at the ends of the stack (either may be off):
//  * tables mode (h1 == nullptr): H_1 = W_1[tag] + P[node] is gathered in the prologue -- what lpg_seal_h1_tables_* writes
//    to HBM and this kernel would read back (187 MB each way per 100 K candidates, one launch);
//  * pooled != nullptr (with head_k1): SortPooling (SEAL/model/gnn.py:82-91) of the head's 16-wide rows runs in the
//    epilogue -- a workgroup owns whole subgraphs, the rows and sort keys are in its registers -- and the kernel writes
//    pooled [B, pool_k, 16] instead of a16 [total, 16] and key [total] (lpg_seal_sortpool_f32: another launch).
struct ConvFuse {
  const int32_t* nodes;      // [B, 128] subgraph vertices' graph node ids
  const int32_t* tags;       // [B, 128] raw DRNL tags
  const int32_t* tag_map;    // raw tag -> column of the one-hot block (may be null: tags are used as they are)
  int64_t tag_map_len, T;
  const float* w1_tag;       // [T, 32] the tag rows of W_1
  const float* p_table;      // [n_graph, 32] emb @ W_1[T:]
  int64_t n_graph;
  float* pooled;             // [B, pool_k, 16]
  int pool_k;
  int32_t* err_flag;
};

struct RowCtxT {
  int i, start;        // vertex index inside its subgraph (-100000: no row), the subgraph's first local row
  float one;           // D^-1 of the row times 2 log2(e) (0: no row)
  uint32_t dadd;       // what "+ 1.0" is on the diagonal's bf16: 0x0080 when it already holds 1.0 (self loop), else 0x3F80
  uint32_t b0, b1, b2, b3;
  int j_lo, j_hi;      // column tiles the tile's subgraphs overlap
};

// K1: the head's conv1d #1 is fused (k1 != nullptr).  A template parameter only for the LDS budget: the concat-writing
// fp32 variant without the head (what a caller who wants concat(Z1, Z2, Z3) as SEAL/model/gnn.py:63 produces it gets)
// needs its 9 KB transposition slice but not the 9.8 KB of k1 fragments -- with both it was 89 KB, ONE workgroup per CU
// instead of two ("desired occupancy was 2, final occupancy is 1"; 1.13 ms per 100 K candidates against 0.58 fused).
// -DLPG_TUNING builds only (profiles/seal_conv_phases.py): where a workgroup's time per 128-row window goes, in ticks of
// the 100 MHz wall clock summed over the launch -- [1] the layers, [2] SortPooling epilogue, [3] windows, and the prologue
// in steps: [4] window bounds + subgraph offsets, [5] row table + tags / nodes, [7] table gathers + plane split.  Stamps sit
// right behind barriers the kernel has anyway.
#ifdef LPG_TUNING
__device__ unsigned long long g_conv_phase[12];
#define CONV_PHASE(k)                                                \
  do {                                                               \
    if (threadIdx.x == 0) {                                          \
      const unsigned long long now_ = __builtin_amdgcn_s_memrealtime(); \
      cph_acc[k] += now_ - cph_t0;                                   \
      cph_t0 = now_;                                                 \
    }                                                                \
  } while (0)
#else
#define CONV_PHASE(k) do { } while (0)
#endif

template <int NP, typename TIN, bool CAT, bool K1>
// (fp32 with BOTH the concat and the fused head -- a combination no caller in the package uses -- needs 260 registers: one
// workgroup per CU is declared for it instead of letting the compiler miss a bound of two)
__global__ __launch_bounds__(kConvBlock, (NP == 1 && !CAT) ? 3 : ((NP == 3 && CAT && K1) ? 1 : 2)) void seal_conv_t2_kernel(
    const int64_t* __restrict__ node_off, const int32_t* __restrict__ first, const uint32_t* __restrict__ adj_bits,
    const TIN* __restrict__ h1, int64_t ldh, const TIN* __restrict__ w2, const TIN* __restrict__ w3,
    const TIN* __restrict__ w4, TIN* __restrict__ cat, int64_t ldc, float* __restrict__ key,
    const float* __restrict__ k1, float* __restrict__ a16, const ConvFuse fx, int64_t n_windows,
    const int64_t* __restrict__ total_ptr) {
  // Windows are walked with a grid stride: n_windows of them, or -- total_ptr given (the batch's vertex total, still on
  // the device) -- ceil(*total_ptr / 128), so that the host never has to read the total to size this launch.  The weight
  // fragments are staged once per workgroup, not once per window.
  __shared__ __attribute__((aligned(16))) uint16_t Hp[NP][32][HS];     // H^T as NP bf16 planes
  __shared__ __attribute__((aligned(16))) uint4 lut[256];             // byte -> 8 x bf16 {0, 1}
  __shared__ __attribute__((aligned(16))) uint4 Wb[2][NP][2][64];     // W_2, W_3: A fragments [layer][plane][k step][lane]
  __shared__ __attribute__((aligned(16))) uint4 Kb[K1 ? 3 : 1][K1 ? 3 : 1][2][2][K1 ? 17 : 1];   // k1 slices (+ w4 in row 16 of layer 3): [layer][plane][k step][h][row]
  __shared__ __attribute__((aligned(16))) uint4 W4b[NP][2][2];        // w4 alone (no head fusion): [plane][k step][h], row 0
  __shared__ uint32_t rinfo[RMAX];
  __shared__ int soff[WIN + 2];
  __shared__ int32_t rnode[RMAX];                   // tables mode: the row's graph node (-1: invalid) ...
  __shared__ uint16_t rtag[RMAX];                   // ... and its column in the one-hot block
  __shared__ __attribute__((aligned(16))) float zt[CAT ? kConvBlock / 64 : 1][CAT ? 16 : 1][36];
  // ---- once per workgroup: the weight operands
  {
  const int tid = threadIdx.x;
  // W_l[f][o] -> the A fragment of lane (o = ln & 31, hh = ln >> 5) for k step s2: f = acc_row(8 s2 + j, hh), j = 0..7
  {
    const int l = tid >> 7, s2 = (tid >> 6) & 1, ln = tid & 63;       // 2 x 2 x 64 = 256 fragments: one per thread
    const TIN* w = l == 0 ? w2 : w3;
    uint32_t pk[NP][4];
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      uint32_t pr[NP];
      planes_pair<NP>(to_f32(w[acc_row(8 * s2 + 2 * jj, ln >> 5) * 32 + (ln & 31)]),
                      to_f32(w[acc_row(8 * s2 + 2 * jj + 1, ln >> 5) * 32 + (ln & 31)]), pr);
#pragma unroll
      for (int p = 0; p < NP; ++p) pk[p][jj] = pr[p];
    }
#pragma unroll
    for (int p = 0; p < NP; ++p) Wb[l][p][s2][ln] = make_uint4(pk[p][0], pk[p][1], pk[p][2], pk[p][3]);
  }
  if (k1) {
    // k1[32 l + f][c] -> row c of the A operand (rows 0..15); row 16 of layer 3 = w4 (32 -> 1 rides along)
    for (int idx = tid; idx < 3 * 2 * 2 * 17; idx += kConvBlock) {
      const int row = idx % 17, hh = (idx / 17) & 1, s2 = (idx / 34) & 1, l = idx / 68;
      uint32_t pk[3][4];
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        float va = 0.f, vb = 0.f;
        const int fa = acc_row(8 * s2 + 2 * jj, hh), fb = acc_row(8 * s2 + 2 * jj + 1, hh);
        if (row < 16) { va = k1[(l * 32 + fa) * 16 + row]; vb = k1[(l * 32 + fb) * 16 + row]; }
        else if (l == 2) { va = to_f32(w4[fa]); vb = to_f32(w4[fb]); }
        uint32_t pr[3];
        planes_pair<3>(va, vb, pr);
#pragma unroll
        for (int p = 0; p < 3; ++p) pk[p][jj] = pr[p];
      }
#pragma unroll
      for (int p = 0; p < 3; ++p) Kb[l][p][s2][hh][row] = make_uint4(pk[p][0], pk[p][1], pk[p][2], pk[p][3]);
    }
  } else if (tid < 4) {
    const int s2 = tid >> 1, hh = tid & 1;
    uint32_t pk[NP][4];
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      uint32_t pr[NP];
      planes_pair<NP>(to_f32(w4[acc_row(8 * s2 + 2 * jj, hh)]), to_f32(w4[acc_row(8 * s2 + 2 * jj + 1, hh)]), pr);
#pragma unroll
      for (int p = 0; p < NP; ++p) pk[p][jj] = pr[p];
    }
#pragma unroll
    for (int p = 0; p < NP; ++p) W4b[p][s2][hh] = make_uint4(pk[p][0], pk[p][1], pk[p][2], pk[p][3]);
  }
  }
  const int64_t nwin = total_ptr ? (total_ptr[0] + WIN - 1) / WIN : n_windows;
#ifdef LPG_TUNING
  unsigned long long cph_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, cph_t0 = 0;
#endif
  for (int64_t win_id = blockIdx.x; win_id < nwin; win_id += gridDim.x) {
  __syncthreads();                                  // the previous window's last readers of the planes / tables are done
#ifdef LPG_TUNING
  if (threadIdx.x == 0) cph_t0 = __builtin_amdgcn_s_memrealtime();
#endif
  // the thread's coordinates are re-derived per window behind an opaque zero: with loop-invariant lane arithmetic the
  // compiler hoisted every LDS / global address of the body out of the window loop and kept ~70 more VGPRs live
  // (186 -> 250, the bf16 kernel lost its third wave per SIMD)
  int opaque0;
  asm volatile("v_mov_b32 %0, 0" : "=v"(opaque0));
  const int tid = static_cast<int>(threadIdx.x) + opaque0, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int bf = first[win_id], bl = first[win_id + 1];
  if (bl <= bf) continue;                           // a window inside one big subgraph owns nothing (block-uniform)
  const int nsub = bl - bf;
  const int64_t r_lo = node_off[bf];
  const int nrows = static_cast<int>(node_off[bl] - r_lo);
  const int ntiles = (nrows + 31) >> 5;             // <= 8

  // ---- H_1 planes: a wave-sized group of items covers 16 row pairs x 4 feature quads (half a 32-row tile, half the
  // features): lane (i = l & 15, q = l >> 4) takes the 16-byte pieces [4 kq, 4 kq + 4) of rows 2 rp and 2 rp + 1 and
  // writes 12 (or 4) row-pair dwords; dword index = feature * 132 + rp, so the 64 lanes of a write hit 64 banks.
  // `piece(row, kq, v)` yields the four values: a 16-byte read of h1, or (tables mode, after the row table below) the
  // sum of two 16-byte gathers.
  auto fill_planes = [&](auto piece) {
    // U items per thread and step: every gather of the step is issued before the first plane is split (the fp32 kernel
    // has the registers for two: a 128-row window is then ONE round trip to the tables instead of two)
    constexpr int U = NP == 3 ? 2 : 1;
    for (int idx0 = tid; idx0 < ntiles * 128; idx0 += U * kConvBlock) {
      float v0[U][4], v1[U][4];
      int rows[U], kqs[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int idx = idx0 + u * kConvBlock;
        const int g = idx >> 6, l = idx & 63;
        const int rp = (g >> 1) * 16 + (l & 15);
        kqs[u] = (g & 1) * 4 + (l >> 4);
        rows[u] = idx < ntiles * 128 ? 2 * rp : nrows;                    // beyond the window: nothing to read or write
#pragma unroll
        for (int j = 0; j < 4; ++j) v0[u][j] = v1[u][j] = 0.f;             // rows past the end are zero (0 x NaN is not)
        if (rows[u] < nrows) piece(rows[u], kqs[u], v0[u]);
        if (rows[u] + 1 < nrows) piece(rows[u] + 1, kqs[u], v1[u]);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (idx0 + u * kConvBlock >= ntiles * 128) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          uint32_t pr[NP];
          planes_pair<NP>(v0[u][j], v1[u][j], pr);
#pragma unroll
          for (int p = 0; p < NP; ++p) *reinterpret_cast<uint32_t*>(&Hp[p][4 * kqs[u] + j][rows[u]]) = pr[p];
        }
      }
    }
  };
  if (h1)
    fill_planes([&](int row, int kq, float (&v)[4]) {
      if constexpr (sizeof(TIN) == 4) {
        const float4 q = *reinterpret_cast<const float4*>(h1 + (r_lo + row) * ldh + 4 * kq);
        v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
      } else {
        const uint2 q = *reinterpret_cast<const uint2*>(h1 + (r_lo + row) * ldh + 4 * kq);
        v[0] = bf16_to_f32(q.x & 0xFFFFu); v[1] = bf16_to_f32(q.x >> 16); v[2] = bf16_to_f32(q.y & 0xFFFFu); v[3] = bf16_to_f32(q.y >> 16);
      }
    });
  {                                                 // byte -> 8 bf16 (bit j set -> 1.0 = 0x3F80 in element j)
    uint32_t d[4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
      d[q] = (((tid >> (2 * q)) & 1) ? 0x3F80u : 0u) | (((tid >> (2 * q + 1)) & 1) ? 0x3F800000u : 0u);
    lut[tid] = make_uint4(d[0], d[1], d[2], d[3]);
  }
  // ---- per-subgraph offsets, per-row (subgraph, vertex)
  for (int q = tid; q <= nsub; q += kConvBlock) soff[q] = static_cast<int>(node_off[bf + q] - r_lo);
  __syncthreads();
  CONV_PHASE(4);
  // a thread per ROW finds its subgraph by bisection over the <= 129 offsets (a thread per SUBGRAPH writing its rows one
  // after the other was 2.1 of the prologue's 6.3 us on the reference's candidate mix: 54 dependent LDS writes)
  for (int lr = tid; lr < nrows; lr += kConvBlock) {
    int lo = 0, hi = nsub;                           // the last q with soff[q] <= lr (an empty subgraph shares its offset
    while (hi - lo > 1) {                            // with its successor and is skipped: the successor is the last)
      const int mid = (lo + hi) >> 1;
      if (soff[mid] <= lr) lo = mid; else hi = mid;
    }
    const uint32_t ri = (static_cast<uint32_t>(lo) << 8) | static_cast<uint32_t>(lr - soff[lo]);
    rinfo[lr] = ri;
    if (!h1) {
      // tables mode: the row's (node, tag column) in the same pass -- no barrier between the row table and its first use
      const int64_t slot = static_cast<int64_t>(bf + (ri >> 8)) * MAXN + (ri & 255u);
      int32_t tg = fx.tags[slot];
      const int32_t nd = fx.nodes[slot];
      bool ok = true;
      if (fx.tag_map) {
        ok = tg >= 0 && tg < fx.tag_map_len;
        tg = fx.tag_map[ok ? tg : 0];
      }
      ok = ok && tg >= 0 && tg < fx.T && nd >= 0 && nd < fx.n_graph;
      if (!ok) flag_error(fx.err_flag, LPG_F_INDEX_RANGE);
      rnode[lr] = ok ? nd : -1;
      rtag[lr] = static_cast<uint16_t>(ok ? tg : 0);
    }
  }
  __syncthreads();
  CONV_PHASE(5);
  if (!h1) {
    // tables mode: per row its (node, tag column), then the planes from W_1[tag] + P[node] (fp32 sum; NP = 1 rounds it
    // once to bf16 -- bit for bit what lpg_seal_h1_tables_f32 / _bf16 followed by the h1 read above gives)
    fill_planes([&](int row, int kq, float (&v)[4]) {
      const int32_t nd = rnode[row];
      if (nd < 0) return;                              // an invalid row stays zero, as the tables kernel leaves it
      const float4 a = *reinterpret_cast<const float4*>(fx.w1_tag + static_cast<int64_t>(rtag[row]) * 32 + 4 * kq);
      const float4 q = *reinterpret_cast<const float4*>(fx.p_table + static_cast<int64_t>(nd) * 32 + 4 * kq);
      v[0] = a.x + q.x; v[1] = a.y + q.y; v[2] = a.z + q.z; v[3] = a.w + q.w;
    });
    __syncthreads();
  }
  CONV_PHASE(7);

  auto row_ctx = [&](int I) {
    RowCtxT c{-100000, 0, 0.f, 0x3F80u, 0u, 0u, 0u, 0u, 0, -1};
    if (I >= ntiles) return c;
    const int lr0 = 32 * I, lr1 = min(32 * I + 31, nrows - 1);
    c.j_lo = soff[rinfo[lr0] >> 8] >> 5;
    c.j_hi = (soff[(rinfo[lr1] >> 8) + 1] - 1) >> 5;
    const int lr = 32 * I + r;
    if (lr < nrows) {
      const uint32_t ri = rinfo[lr];
      const int q = static_cast<int>(ri >> 8);
      c.i = static_cast<int>(ri & 255u);
      c.start = soff[q];
      const uint4 q4 = *reinterpret_cast<const uint4*>(adj_bits + (static_cast<int64_t>(bf + q) * MAXN + c.i) * WORDS);
      c.b0 = q4.x; c.b1 = q4.y; c.b2 = q4.z; c.b3 = q4.w;
      const int pc = __popc(c.b0) + __popc(c.b1) + __popc(c.b2) + __popc(c.b3);
      // D^-1 = 1 / rowsum(A + I) (fp32 division of small integers is exact to rounding), times 2 log2(e) for tanh_scaled
      c.one = (1.0f / static_cast<float>(pc + 1)) * 2.885390081777927f;
      const uint32_t wi = c.i < 32 ? c.b0 : (c.i < 64 ? c.b1 : (c.i < 96 ? c.b2 : c.b3));
      c.dadd = ((wi >> (c.i & 31)) & 1u) ? 0x0080u : 0x3F80u;
    }
    return c;
  };
  const RowCtxT ctx0 = row_ctx(wave), ctx1 = row_ctx(wave + kConvBlock / 64);

  // Z^T of a row tile: tanh(D^-1 sum_J H_J^T (A + I)_{IJ}^T); lane = vertex, register i = feature acc_row(i, h)
  f32x16 zero16;                                     // one zero accumulator for the first MFMA of every chain (D != C is allowed)
#pragma unroll
  for (int i = 0; i < 16; ++i) zero16[i] = 0.f;
  auto aggregate = [&](const RowCtxT& cx) {
    f32x16 acc = zero16;
    for (int J = cx.j_lo; J <= cx.j_hi; ++J) {
      const int p0 = 32 * J - cx.start;
      const int w = p0 >> 5, sh = p0 & 31;           // arithmetic shift: p0 may be negative
      auto word = [&](int k) { return k == 0 ? cx.b0 : (k == 1 ? cx.b1 : (k == 2 ? cx.b2 : (k == 3 ? cx.b3 : 0u))); };
      const uint32_t lo = word(w), hi = word(w + 1);
      const uint32_t win = sh ? ((lo >> sh) | (hi << (32 - sh))) : lo;
      const int cd = cx.i - p0;                      // the diagonal's column in this tile (out of range: none)
      bf16x8 av[2];
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        uint4 t = lut[(win >> (16 * s2 + 8 * h)) & 0xFFu];
        const int dj = cd - (16 * s2 + 8 * h);
        const uint32_t add = (dj >= 0 && dj < 8) ? (cx.dadd << (16 * (dj & 1))) : 0u;
        const int sel = dj >> 1;
        t.x += sel == 0 ? add : 0u;
        t.y += sel == 1 ? add : 0u;
        t.z += sel == 2 ? add : 0u;
        t.w += sel == 3 ? add : 0u;
        av[s2] = *reinterpret_cast<bf16x8*>(&t);
      }
#pragma unroll
      for (int p = NP - 1; p >= 0; --p)              // smallest terms first
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          const bf16x8 hv = *reinterpret_cast<const bf16x8*>(&Hp[p][r][32 * J + 16 * s2 + 8 * h]);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(hv, av[s2], acc, 0, 0, 0);
        }
    }
    // tanh(acc * s) = 1 - 2 / (2^(acc * s2) + 1), two values at a time: the multiply, the add and the final fma are
    // packed (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32), only exp2 and rcp stay per value -- 3.5 instead of 5
    // instructions per value, sixteen values per tile and layer.  cx.one carries D^-1 * 2 log2(e).
    using f32x2 = __attribute__((ext_vector_type(2))) float;
    const f32x2 s2v = {cx.one, cx.one}, one2 = {1.0f, 1.0f}, m2 = {-2.0f, -2.0f};
#pragma unroll
    for (int i = 0; i < 16; i += 2) {
      f32x2 v = {acc[i], acc[i + 1]};
      v = v * s2v;
      v = f32x2{__builtin_amdgcn_exp2f(v.x), __builtin_amdgcn_exp2f(v.y)} + one2;
      v = f32x2{__builtin_amdgcn_rcpf(v.x), __builtin_amdgcn_rcpf(v.y)};
      v = __builtin_elementwise_fma(m2, v, one2);
      acc[i] = v.x;
      acc[i + 1] = v.y;
    }
    return acc;
  };

  f32x16 aa0, aa1;                                   // a16^T tiles (head fusion)
#pragma unroll
  for (int i = 0; i < 16; ++i) aa0[i] = aa1[i] = 0.f;
  float kreg0 = 0.f, kreg1 = 0.f;                    // the tiles' sort keys (layer 4, feature 0), kept for the pooling epilogue
#pragma unroll 1
  for (int layer = 1; layer <= 4; ++layer) {
    f32x16 hn0, hn1;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int I = wave + t * (kConvBlock / 64);
      if (I >= ntiles) continue;                     // wave-uniform
      const f32x16 z = aggregate(t == 0 ? ctx0 : ctx1);
      const int lr = 32 * I + r;
      if (layer == 4) {
        if (t == 0) kreg0 = z[0]; else kreg1 = z[0];               // feature 0
        if (key && h == 0 && lr < nrows) key[r_lo + lr] = z[0];
        continue;
      }
      if constexpr (CAT) {
        if (cat) {
          // half a tile at a time through a wave-private LDS slice: the rows leave as whole lines
#pragma unroll
          for (int half = 0; half < 2; ++half) {
            if ((r >> 4) == half) {
#pragma unroll
              for (int g = 0; g < 4; ++g)                // features 8 g + 4 h + {0..3}
                *reinterpret_cast<float4*>(&zt[wave][r & 15][8 * g + 4 * h]) = make_float4(z[4 * g], z[4 * g + 1], z[4 * g + 2], z[4 * g + 3]);
            }
            __builtin_amdgcn_s_waitcnt(0xC07F);          // lgkmcnt(0): the slice is wave-private
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int k = 0; k < 2; ++k) {
              const int row = (lane >> 3) + 8 * k, quad = lane & 7;
              const float4 v = *reinterpret_cast<const float4*>(&zt[wave][row][4 * quad]);
              const int lrr = 32 * I + 16 * half + row;
              if (lrr < nrows) {
                TIN* dst = cat + (r_lo + lrr) * ldc + (layer - 1) * 32 + 4 * quad;
                if constexpr (sizeof(TIN) == 4) {
                  *reinterpret_cast<float4*>(dst) = v;
                } else {
                  *reinterpret_cast<uint2*>(dst) = make_uint2(pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w));
                }
              }
            }
            __builtin_amdgcn_s_waitcnt(0xC07F);
            __builtin_amdgcn_wave_barrier();
          }
        }
      }
      // Z^T as bf16 B fragments: k step s2 takes the lane's registers 8 s2 .. 8 s2 + 7 (features acc_row(8 s2 + j, h))
      bf16x8 zb[NP][2];
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        uint32_t pk[NP][4];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          uint32_t pr[NP];
          planes_pair<NP>(z[8 * s2 + 2 * jj], z[8 * s2 + 2 * jj + 1], pr);
#pragma unroll
          for (int p = 0; p < NP; ++p) pk[p][jj] = pr[p];
        }
#pragma unroll
        for (int p = 0; p < NP; ++p) {
          uint4 q = make_uint4(pk[p][0], pk[p][1], pk[p][2], pk[p][3]);
          zb[p][s2] = *reinterpret_cast<bf16x8*>(&q);
        }
      }
      // H'^T = W^T Z^T
      f32x16 hn = zero16;
      const bool merged = k1 && layer == 3;          // 32 -> 1 (w4) rides in row 16 of the conv1d #1 product
      if (layer < 3) {
#pragma unroll
        for (int q = 0; q < Terms<NP>::N; ++q)
#pragma unroll
          for (int s2 = 0; s2 < 2; ++s2) {
            const uint4 wv = Wb[layer - 1][Terms<NP>::wp[q]][s2][lane];
            hn = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(&wv), zb[Terms<NP>::zp[q]][s2], hn, 0, 0, 0);
          }
      } else if (!merged) {                          // 32 -> 1: only output row o = 0 is non-zero
#pragma unroll
        for (int q = 0; q < Terms<NP>::N; ++q)
#pragma unroll
          for (int s2 = 0; s2 < 2; ++s2) {
            uint4 wv = W4b[Terms<NP>::wp[q]][s2][h];
            if (r != 0) wv = make_uint4(0u, 0u, 0u, 0u);
            hn = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(&wv), zb[Terms<NP>::zp[q]][s2], hn, 0, 0, 0);
          }
      }
      if (t == 0) hn0 = hn; else hn1 = hn;
      if (k1) {                                      // wave-uniform
        f32x16 aa = t == 0 ? aa0 : aa1;
#pragma unroll
        for (int q = 0; q < KTerms<NP>::N; ++q)
#pragma unroll
          for (int s2 = 0; s2 < 2; ++s2) {
            uint4 kv = Kb[layer - 1][KTerms<NP>::wp[q]][s2][h][r <= 16 ? r : 16];
            if (r > 16 || (r == 16 && !merged)) kv = make_uint4(0u, 0u, 0u, 0u);
            aa = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(&kv), zb[KTerms<NP>::zp[q]][s2], aa, 0, 0, 0);
          }
        if (t == 0) aa0 = aa; else aa1 = aa;
        if (layer == 3 && a16 && lr < nrows) {       // outputs c = acc_row(i, h) < 16  <=>  i < 8
          *reinterpret_cast<float4*>(a16 + (r_lo + lr) * 16 + 4 * h) = make_float4(aa[0], aa[1], aa[2], aa[3]);
          *reinterpret_cast<float4*>(a16 + (r_lo + lr) * 16 + 8 + 4 * h) = make_float4(aa[4], aa[5], aa[6], aa[7]);
        }
      }
    }
    if (layer == 4) break;
    __syncthreads();                                 // every aggregation of this layer has read H
    CONV_PHASE(0);
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int I = wave + t * (kConvBlock / 64);
      if (I >= ntiles) continue;
      if (k1 && layer == 3) {
        // layer 4 reads feature 0 only (the sort key): it sits in row 16 of the merged product
        if (h == 0) {
          uint32_t pl[NP];
          planes_of<NP>(t == 0 ? aa0[8] : aa1[8], pl);
#pragma unroll
          for (int p = 0; p < NP; ++p) Hp[p][0][32 * I + r] = static_cast<uint16_t>(pl[p]);
        }
        continue;
      }
#pragma unroll
      for (int i = 0; i < 16; i += 2) {              // lane = vertex 32 I + r, registers i, i + 1 = features acc_row(i, h), + 1
        uint32_t pr[NP];
        planes_pair<NP>(t == 0 ? hn0[i] : hn1[i], t == 0 ? hn0[i + 1] : hn1[i + 1], pr);
        const int f = acc_row(i, h);                 // acc_row(i + 1, h) == f + 1 for even i
#pragma unroll
        for (int p = 0; p < NP; ++p) {
          Hp[p][f][32 * I + r] = static_cast<uint16_t>(pr[p]);
          Hp[p][f + 1][32 * I + r] = static_cast<uint16_t>(pr[p] >> 16);     // ds_write_b16_d16_hi: no shift
        }
      }
    }
    __syncthreads();
    CONV_PHASE(6);
  }
  CONV_PHASE(1);
  if (fx.pooled && k1) {
    // ---- SortPooling in the epilogue (SEAL/model/gnn.py:82-91): per subgraph the pool_k rows with the largest keys,
    // key descending, ties by vertex index, NaN last (the total order of seal_sortpool_kernel); missing rows are zero.
    // The 16-wide rows and the keys move from the accumulators into LDS on top of the planes and the bit table, which
    // nobody reads any more.
    float* As = reinterpret_cast<float*>(&Hp[0][0][0]);               // [nrows][16]: <= 16 320 B <= one plane
    // [nrows] sort composites: (order-preserving integer image of the key) << 8 | (127 - vertex index) -- "a comes before b"
    // is ONE unsigned 64-bit compare (the float form with its NaN and tie rules cost ~10 VALU instructions per key in a
    // kernel that is VALU-bound: +20 %).  NaN -> 0 (last), -0 -> +0 (a tie with +0, as `==` has it).
    unsigned long long* comp = reinterpret_cast<unsigned long long*>(&lut[0]);
    __syncthreads();                                                  // every wave is past its last read of Hp / lut
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int I = wave + t * (kConvBlock / 64);
      const int lr = 32 * I + r;
      if (I < ntiles && lr < nrows) {
        const f32x16& aa = t == 0 ? aa0 : aa1;
        if (h == 0) {
          const float kf = (t == 0 ? kreg0 : kreg1) + 0.0f;
          uint32_t u = __float_as_uint(kf);
          u = (kf != kf) ? 0u : (u ^ ((u >> 31) ? 0xFFFFFFFFu : 0x80000000u));
          comp[lr] = (static_cast<unsigned long long>(u) << 8) | static_cast<unsigned long long>(127 - static_cast<int>(rinfo[lr] & 255u));
        }
        *reinterpret_cast<float4*>(As + lr * 16 + 4 * h) = make_float4(aa[0], aa[1], aa[2], aa[3]);
        *reinterpret_cast<float4*>(As + lr * 16 + 8 + 4 * h) = make_float4(aa[4], aa[5], aa[6], aa[7]);
      }
    }
    __syncthreads();
    CONV_PHASE(8);
    const int pk = fx.pool_k;
    // a PAIR of neighbouring lanes per vertex, each counting over half of the subgraph (a window has 128 rows on average:
    // with a thread per vertex half of the workgroup idled through the kernel's one O(n^2) loop, 2.3 of 25 us per window)
    for (int it = tid; it < 2 * ((nrows + 31) & ~31); it += kConvBlock) {   // whole waves: the shuffle below needs both lanes
      const int lr = min(it >> 1, nrows - 1), half = it & 1;
      const uint32_t ri = rinfo[lr];
      const int q = static_cast<int>(ri >> 8), s0 = soff[q], n = soff[q + 1] - s0;
      const unsigned long long mine = comp[lr];
      const int j0 = half ? (n >> 1) : 0, j1 = half ? n : (n >> 1);
      int rank = 0;
#pragma unroll 4
      for (int j = j0; j < j1; ++j) rank += comp[s0 + j] > mine ? 1 : 0;
      rank += __shfl_xor(rank, 1);
      if (half == 0 && (it >> 1) < nrows) rtag[lr] = static_cast<uint16_t>(rank);   // (the tag columns are not needed any more)
    }
    __syncthreads();
    CONV_PHASE(9);
    // the kept rows leave as whole 64-byte pieces: four lanes per row, 16 rows per wave instruction (a thread per row
    // writing its four 16-byte pieces one after the other touched 64 different lines per instruction)
    for (int idx = tid; idx < nrows * 4; idx += kConvBlock) {
      const int lr = idx >> 2, rank = rtag[lr];
      if (rank < pk) {
        const int q = static_cast<int>(rinfo[lr] >> 8);
        reinterpret_cast<float4*>(fx.pooled + (static_cast<int64_t>(bf + q) * pk + rank) * 16)[idx & 3] =
            reinterpret_cast<const float4*>(As + lr * 16)[idx & 3];
      }
    }
    // subgraphs with fewer than pool_k vertices (most of them): zero rows, one 16-byte piece per thread and step (a thread
    // per subgraph writing its ~10 rows in series was as slow as the separate SortPooling launch)
    for (int idx = tid; idx < nsub * pk * 4; idx += kConvBlock) {
      const int q = idx / (pk * 4), rem = idx - q * (pk * 4), rr = rem >> 2;
      if (rr >= soff[q + 1] - soff[q])
        reinterpret_cast<float4*>(fx.pooled + (static_cast<int64_t>(bf + q) * pk + rr) * 16)[rem & 3] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
#ifdef LPG_TUNING
  CONV_PHASE(2);
  if (threadIdx.x == 0) cph_acc[3] += 1ull;
#endif
  }   // windows
#ifdef LPG_TUNING
  if (threadIdx.x == 0 && cph_acc[3])
    for (int k = 0; k < 12; ++k) atomicAdd(&g_conv_phase[k], cph_acc[k]);
#endif
}

// workgroups of `kernel` the chip holds at once (occupancy x CUs); 512 if the runtime cannot say
inline int64_t resident_slots(const void* kernel) {
  int per_cu = 0, dev = 0;
  hipDeviceProp_t prop;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, kConvBlock, 0) != hipSuccess || per_cu < 1) per_cu = 2;
  if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 512;
  return static_cast<int64_t>(per_cu) * prop.multiProcessorCount;
}

template <int NP, typename TIN>
int launch_conv2(const int64_t* node_off, const int32_t* first, int64_t n_windows, const uint32_t* adj_bits, const TIN* h1,
                 int64_t ldh, const TIN* w2, const TIN* w3, const TIN* w4, TIN* cat, int64_t ldc, float* key,
                 const float* head_k1, float* head_a16, const ConvFuse& fx, hipStream_t st, int64_t n_subgraphs = 0) {
  // n_windows >= 0: known on the host; < 0: ceil(node_off[n_subgraphs] / 128), read on the device (no host sync between
  // the extraction and this launch).  Either way a bounded grid walks the windows: the weight fragments are staged once
  // per workgroup instead of once per window.
  const int64_t upper = n_windows >= 0 ? n_windows : n_subgraphs + 1;          // a subgraph has <= 128 rows
  // eight workgroups per resident slot: the hardware dispatcher still balances windows of different cost (a grid of exactly
  // the resident slots waits for its unluckiest member; 1.33 x the slots runs a second, mostly empty round: bf16 0.44 -> 0.64 ms;
  // 4 x / 8 x / 16 x the slots: fused bf16 stack 0.470 / 0.458 / 0.475 ms, fp32 0.700 / 0.681 / 0.696)
  static const int64_t slots_v[4] = {
      resident_slots(reinterpret_cast<const void*>(&seal_conv_t2_kernel<NP, TIN, false, false>)),
      resident_slots(reinterpret_cast<const void*>(&seal_conv_t2_kernel<NP, TIN, false, true>)),
      resident_slots(reinterpret_cast<const void*>(&seal_conv_t2_kernel<NP, TIN, true, false>)),
      resident_slots(reinterpret_cast<const void*>(&seal_conv_t2_kernel<NP, TIN, true, true>))};
  const int64_t cap = 8 * slots_v[(cat ? 2 : 0) + (head_k1 ? 1 : 0)];
  const int64_t g = upper < cap ? (upper < 1 ? 1 : upper) : cap;
  const dim3 grid(static_cast<unsigned>(g)), blk(kConvBlock);
  const int64_t* total_ptr = n_windows >= 0 ? nullptr : node_off + n_subgraphs;
  if (cat && head_k1)
    hipLaunchKernelGGL((seal_conv_t2_kernel<NP, TIN, true, true>), grid, blk, 0, st, node_off, first, adj_bits, h1, ldh, w2, w3, w4,
                       cat, ldc, key, head_k1, head_a16, fx, n_windows, total_ptr);
  else if (cat)
    hipLaunchKernelGGL((seal_conv_t2_kernel<NP, TIN, true, false>), grid, blk, 0, st, node_off, first, adj_bits, h1, ldh, w2, w3, w4,
                       cat, ldc, key, head_k1, head_a16, fx, n_windows, total_ptr);
  else if (head_k1)
    hipLaunchKernelGGL((seal_conv_t2_kernel<NP, TIN, false, true>), grid, blk, 0, st, node_off, first, adj_bits, h1, ldh, w2, w3, w4,
                       cat, ldc, key, head_k1, head_a16, fx, n_windows, total_ptr);
  else
    hipLaunchKernelGGL((seal_conv_t2_kernel<NP, TIN, false, false>), grid, blk, 0, st, node_off, first, adj_bits, h1, ldh, w2, w3, w4,
                       cat, ldc, key, head_k1, head_a16, fx, n_windows, total_ptr);
  return check_launch("seal_conv_stack");
}

// argument checks shared by the fused entry points; fills fx
int make_fuse(ConvFuse* fx, bool have_h1, const int32_t* nodes, const int32_t* tags, const int32_t* tag_map, int64_t tag_map_len,
              int64_t T, const float* w1_tag, const float* p_table, int64_t n_graph, const float* head_k1, float* pooled,
              int64_t pool_k, float* key, float* head_a16, void* cat, int32_t* err_flag) {
  auto al16 = [](const void* p) { return reinterpret_cast<uintptr_t>(p) % 16 == 0; };
  if (!have_h1) {
    LPG_REQUIRE(nodes && tags && w1_tag && p_table && T >= 1 && T < 65536 && n_graph >= 1 && tag_map_len >= 0 && (tag_map || tag_map_len == 0),
                "seal_score_stack: without h1 the stack needs nodes, tags, w1_tag, p_table (1 <= T < 65536)");
    LPG_REQUIRE(al16(w1_tag) && al16(p_table), "seal_score_stack: tables must be 16-byte aligned");
  }
  if (pooled) {
    LPG_REQUIRE(head_k1 && pool_k >= 1 && pool_k <= MAXN && al16(pooled), "seal_score_stack: pooled needs head_k1, 1 <= pool_k <= %d, 16-byte alignment", MAXN);
  } else {
    LPG_REQUIRE(key && (cat || head_k1) && (!head_k1 || head_a16), "seal_score_stack: without pooled the stack needs key and cat or head_a16");
  }
  *fx = ConvFuse{nodes, tags, tag_map, tag_map_len, T, w1_tag, p_table, n_graph, pooled, static_cast<int>(pool_k), err_flag};
  return LPG_OK;
}
}  // namespace

namespace lpg {
int seal_conv2_f32(const int64_t* node_off, const int32_t* first, int64_t n_windows, const uint32_t* adj_bits,
                   const float* h1, int64_t ldh, const float* w2, const float* w3, const float* w4, float* cat, int64_t ldc,
                   float* key, const float* head_k1, float* head_a16, hipStream_t st) {
  return launch_conv2<3, float>(node_off, first, n_windows, adj_bits, h1, ldh, w2, w3, w4, cat, ldc, key, head_k1, head_a16,
                                ConvFuse{}, st);
}
}  // namespace lpg

// ---- the scoring stack with both ends fused: tags / nodes in, pooled rows out
LPG_EXPORT int lpg_seal_score_stack_f32(const int64_t* node_off, const int32_t* first, int64_t n_windows,
                                        const uint32_t* adj_bits, const float* h1, int64_t ldh, const int32_t* nodes,
                                        const int32_t* tags, const int32_t* tag_map, int64_t tag_map_len, int64_t T,
                                        const float* w1_tag, const float* p_table, int64_t n_graph, const float* w2,
                                        const float* w3, const float* w4, const float* head_k1, float* key, float* head_a16,
                                        float* pooled, int64_t pool_k, int64_t n_subgraphs, int32_t* err_flag, void* stream) {
  LPG_REQUIRE(n_windows >= -1 && n_windows < (int64_t(1) << 31) && n_subgraphs >= 0 && n_subgraphs < (int64_t(1) << 31) &&
              (!h1 || (ldh >= 32 && ldh % 4 == 0)), "seal_score_stack: bad sizes");
  if (n_windows == 0 || (n_windows < 0 && n_subgraphs == 0)) return LPG_OK;
  LPG_REQUIRE(node_off && first && adj_bits && w2 && w3 && w4, "seal_score_stack: null pointer");
  LPG_REQUIRE((!h1 || reinterpret_cast<uintptr_t>(h1) % 16 == 0) && (!head_a16 || reinterpret_cast<uintptr_t>(head_a16) % 16 == 0),
              "seal_score_stack: h1 / head_a16 must be 16-byte aligned");
  ConvFuse fx;
  if (int rc = make_fuse(&fx, h1 != nullptr, nodes, tags, tag_map, tag_map_len, T, w1_tag, p_table, n_graph, head_k1, pooled, pool_k,
                         key, head_a16, nullptr, err_flag)) return rc;
  LPG_REQUIRE(head_k1, "seal_score_stack: head_k1 is required (the concat form is lpg_seal_conv_stack_f32)");
  return launch_conv2<3, float>(node_off, first, n_windows, adj_bits, h1, ldh, w2, w3, w4, nullptr, 96, key, head_k1, head_a16, fx,
                                as_stream(stream), n_subgraphs);
}

LPG_EXPORT int lpg_seal_score_stack_bf16(const int64_t* node_off, const int32_t* first, int64_t n_windows,
                                         const uint32_t* adj_bits, const uint16_t* h1, int64_t ldh, const int32_t* nodes,
                                         const int32_t* tags, const int32_t* tag_map, int64_t tag_map_len, int64_t T,
                                         const float* w1_tag, const float* p_table, int64_t n_graph, const uint16_t* w2,
                                         const uint16_t* w3, const uint16_t* w4, const float* head_k1, float* key,
                                         float* head_a16, float* pooled, int64_t pool_k, int64_t n_subgraphs, int32_t* err_flag,
                                         void* stream) {
  LPG_REQUIRE(n_windows >= -1 && n_windows < (int64_t(1) << 31) && n_subgraphs >= 0 && n_subgraphs < (int64_t(1) << 31) &&
              (!h1 || (ldh >= 32 && ldh % 4 == 0)), "seal_score_stack_bf16: bad sizes");
  if (n_windows == 0 || (n_windows < 0 && n_subgraphs == 0)) return LPG_OK;
  LPG_REQUIRE(node_off && first && adj_bits && w2 && w3 && w4, "seal_score_stack_bf16: null pointer");
  LPG_REQUIRE((!h1 || reinterpret_cast<uintptr_t>(h1) % 8 == 0) && (!head_a16 || reinterpret_cast<uintptr_t>(head_a16) % 16 == 0),
              "seal_score_stack_bf16: h1 must be 8-byte, head_a16 16-byte aligned");
  ConvFuse fx;
  if (int rc = make_fuse(&fx, h1 != nullptr, nodes, tags, tag_map, tag_map_len, T, w1_tag, p_table, n_graph, head_k1, pooled, pool_k,
                         key, head_a16, nullptr, err_flag)) return rc;
  LPG_REQUIRE(head_k1, "seal_score_stack_bf16: head_k1 is required (the concat form is lpg_seal_conv_stack_bf16)");
  return launch_conv2<1, uint16_t>(node_off, first, n_windows, adj_bits, h1, ldh, w2, w3, w4, nullptr, 96, key, head_k1, head_a16, fx,
                                   as_stream(stream), n_subgraphs);
}

LPG_EXPORT int lpg_seal_conv_stack_bf16(const int64_t* node_off, const int32_t* first, int64_t n_windows,
                                        const uint32_t* adj_bits, const uint16_t* h1, int64_t ldh, const uint16_t* w2,
                                        const uint16_t* w3, const uint16_t* w4, uint16_t* cat, int64_t ldc, float* key,
                                        const float* head_k1, float* head_a16, void* stream) {
  LPG_REQUIRE(n_windows >= 0 && n_windows < (int64_t(1) << 31) && ldh >= 32 && ldh % 4 == 0 && (!cat || (ldc >= 96 && ldc % 4 == 0)),
              "seal_conv_stack_bf16: bad sizes (ldh, ldc multiples of 4)");
  if (n_windows == 0) return LPG_OK;
  LPG_REQUIRE(node_off && first && adj_bits && h1 && w2 && w3 && w4 && key, "seal_conv_stack_bf16: null pointer");
  LPG_REQUIRE((cat || head_k1) && (!head_k1 == !head_a16), "seal_conv_stack_bf16: needs cat, or head_k1 together with head_a16");
  auto al = [](const void* p, int a) { return reinterpret_cast<uintptr_t>(p) % a == 0; };
  LPG_REQUIRE(al(h1, 8) && (!cat || al(cat, 8)) && (!head_a16 || al(head_a16, 16)),
              "seal_conv_stack_bf16: h1 / cat must be 8-byte aligned, head_a16 16-byte aligned");
  return launch_conv2<1, uint16_t>(node_off, first, n_windows, adj_bits, h1, ldh, w2, w3, w4, cat, ldc, key, head_k1, head_a16,
                                   ConvFuse{}, as_stream(stream));
}

LPG_EXPORT int lpg_seal_conv_phase_ticks(uint64_t* out8_host, int reset) {
#ifdef LPG_TUNING
  LPG_REQUIRE(out8_host, "seal_conv_phase_ticks: null pointer");
  LPG_HIP(hipDeviceSynchronize());
  LPG_HIP(hipMemcpyFromSymbol(out8_host, HIP_SYMBOL(g_conv_phase), 12 * sizeof(unsigned long long)));
  if (reset) {
    const unsigned long long zero[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    LPG_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_conv_phase), zero, sizeof(zero)));
  }
  return LPG_OK;
#else
  (void)out8_host; (void)reset;
  set_error("seal_conv_phase_ticks: only in a -DLPG_TUNING build of the library");
  return LPG_E_INVALID;
#endif
}

// Dense feature linear, bf16 storage (K5 for the bf16 configurations):
//   Y[n,m] = act(X[n,k] @ W^T + bias),  X/W/Y bfloat16, fp32 accumulate on v_mfma_f32_32x32x16_bf16.
// Reference anchor: same op as linear.hip (GCNConv.lin, SEAL's tf.matmul(X, W)); the bf16 storage is
// BASELINE.json configs[2] ("d = 128, bf16"), not something the fp32-only reference has.
//
// A wave owns a 32x32 output tile.  Fragment layout of the instruction: lane l (r = l & 31,
// h = l >> 5) holds A[row r][k = 8h .. 8h+7] and B[k = 8h .. 8h+7][col r] per 16-deep step, i.e.
// 8 CONTIGUOUS bf16 of a row-major [rows][k] tile -- so X and W ([m,k] layout) are staged row-major
// in LDS (rows padded by 16 B) and a fragment is one ds_read_b128.
#include "lpg_common.hpp"

namespace {
using namespace lpg;

using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) short;

constexpr int KT = 64;        // k values staged per step (4 MFMA steps of 16)
constexpr int LDP = KT + 8;   // padded LDS row: 144 B, keeps 16-B alignment

// NT accumulator tiles per wave: a wave owns 32 rows and NT*32 output columns, so X is read once.
// FAST (everything 16-byte aligned: k % 8 == 0, m % 8 == 0): register prefetch of the next K step's
// X / W pieces during the MFMAs, and the output tile staged through LDS so that y is written as whole
// 16-byte pieces of a row (2-byte scattered stores were the slowest part of the first version).
template <int NT, bool FAST>
__global__ __launch_bounds__(kBlock, 2) void linear_mfma_bf16_kernel(const uint16_t* __restrict__ x, int64_t ldx,
                                                                     const uint16_t* __restrict__ w, int w_is_km,
                                                                     const float* __restrict__ bias, int act,
                                                                     uint16_t* __restrict__ y, int64_t ldy, int64_t n,
                                                                     int64_t k, int64_t m) {
  constexpr int XS = 4 * 32 * LDP, WS = NT * 32 * LDP;
  constexpr int TW = NT * 32 + 8;                        // output tile row (bf16), keeps 16-B alignment
  constexpr int SM = (XS + WS) > 4 * 32 * TW ? (XS + WS) : 4 * 32 * TW;
  __shared__ __attribute__((aligned(16))) uint16_t smem[SM];
  uint16_t(*Xs)[32][LDP] = reinterpret_cast<uint16_t(*)[32][LDP]>(smem);
  uint16_t(*Ws)[LDP] = reinterpret_cast<uint16_t(*)[LDP]>(smem + XS);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t row0 = static_cast<int64_t>(blockIdx.x) * 128 + wave * 32;
  const int64_t j0 = static_cast<int64_t>(blockIdx.y) * (NT * 32);
  const int fr = lane & 31, fk = lane >> 5;
  f32x16 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;

  auto mfma_step = [&]() {
#pragma unroll
    for (int s = 0; s < KT / 16; ++s) {
      const bf16x8 a = *reinterpret_cast<const bf16x8*>(&Xs[wave][fr][s * 16 + 8 * fk]);
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const bf16x8 b = *reinterpret_cast<const bf16x8*>(&Ws[t * 32 + fr][s * 16 + 8 * fk]);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[t], 0, 0, 0);
      }
    }
  };
  auto stage_w_km = [&](int64_t k0) {                    // W[k][m]: transpose while staging (2-byte writes)
    for (int it = 0; it < (NT * 32 * KT) / kBlock; ++it) {
      const int idx = it * kBlock + tid, kk = idx / (NT * 32), j = idx % (NT * 32);
      Ws[j][kk] = (k0 + kk < k && j0 + j < m) ? w[(k0 + kk) * m + j0 + j] : uint16_t(0);
    }
  };

  if constexpr (FAST) {
    uint4 xr[4], wr[NT];      // X: 32 rows x 128 B = 256 pieces, 4 per lane; W [NT*32][64]: NT per thread
    auto load_stage = [&](int64_t k0) {
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int piece = it * 64 + lane, r = piece >> 3, seg = piece & 7;
        const int64_t row = row0 + r, kk = k0 + seg * 8;
        const bool ok = row < n && kk < k;
        xr[it] = *reinterpret_cast<const uint4*>(x + (ok ? row * ldx + kk : 0));
      }
      if (!w_is_km) {
#pragma unroll
        for (int it = 0; it < NT; ++it) {
          const int piece = it * kBlock + tid, j = piece >> 3, seg = piece & 7;
          const bool ok = j0 + j < m && k0 + seg * 8 < k;
          wr[it] = *reinterpret_cast<const uint4*>(w + (ok ? (j0 + j) * k + k0 + seg * 8 : 0));
        }
      }
    };
    auto store_stage = [&](int64_t k0) {
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int piece = it * 64 + lane, r = piece >> 3, seg = piece & 7;
        const bool ok = row0 + r < n && k0 + seg * 8 < k;
        *reinterpret_cast<uint4*>(&Xs[wave][r][seg * 8]) = ok ? xr[it] : make_uint4(0, 0, 0, 0);
      }
      if (!w_is_km) {
#pragma unroll
        for (int it = 0; it < NT; ++it) {
          const int piece = it * kBlock + tid, j = piece >> 3, seg = piece & 7;
          const bool ok = j0 + j < m && k0 + seg * 8 < k;
          *reinterpret_cast<uint4*>(&Ws[j][seg * 8]) = ok ? wr[it] : make_uint4(0, 0, 0, 0);
        }
      } else {
        stage_w_km(k0);
      }
    };
    load_stage(0);
    for (int64_t k0 = 0; k0 < k; k0 += KT) {
      asm volatile("" ::: "memory");
      store_stage(k0);
      __syncthreads();
      if (k0 + KT < k) load_stage(k0 + KT);              // in flight while this step's MFMAs run
      mfma_step();
      __syncthreads();
    }
    uint16_t* tile = smem + wave * 32 * TW;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int c = t * 32 + (lane & 31);
      const float b = (bias && j0 + c < m) ? bias[j0 + c] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        tile[row * TW + c] = f32_to_bf16(apply_act(acc[t][r] + b, act));
      }
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);                  // lgkmcnt(0); the tile is wave-private
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int it = 0; it < NT * 2; ++it) {                // 32 rows x NT*4 pieces (8 bf16) = NT*128, NT*2 per lane
      const int piece = it * 64 + lane, r = piece / (NT * 4), c = (piece % (NT * 4)) * 8;
      const int64_t row = row0 + r, col = j0 + c;
      if (row < n && col < m)                            // m % 8 == 0: a piece is all-in or all-out
        *reinterpret_cast<uint4*>(y + row * ldy + col) = *reinterpret_cast<const uint4*>(&tile[r * TW + c]);
    }
  } else {
    for (int64_t k0 = 0; k0 < k; k0 += KT) {
#pragma unroll 4
      for (int it = 0; it < 32; ++it) {
        const int64_t r = row0 + it, kk = k0 + lane;
        Xs[wave][it][lane] = (r < n && kk < k) ? x[r * ldx + kk] : uint16_t(0);
      }
      if (!w_is_km) {
        for (int it = 0; it < (NT * 32 * KT) / kBlock; ++it) {
          const int idx = it * kBlock + tid, j = idx >> 6, kk = idx & 63;
          Ws[j][kk] = (k0 + kk < k && j0 + j < m) ? w[(j0 + j) * k + k0 + kk] : uint16_t(0);
        }
      } else {
        stage_w_km(k0);
      }
      __syncthreads();
      mfma_step();
      __syncthreads();
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int64_t col = j0 + t * 32 + (lane & 31);
      if (col < m) {
        const float b = bias ? bias[col] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int64_t row = row0 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          if (row < n) y[row * ldy + col] = f32_to_bf16(apply_act(acc[t][r] + b, act));
        }
      }
    }
  }
}

template <int NT>
void launch_bf16(const uint16_t* x, int64_t ldx, const uint16_t* w, int w_is_km, const float* bias, int act,
                 uint16_t* y, int64_t ldy, int64_t n, int64_t k, int64_t m, int fast, hipStream_t st) {
  const dim3 grid(static_cast<unsigned>(ceil_div(n, 128)), static_cast<unsigned>(ceil_div(m, NT * 32)));
  if (fast)
    hipLaunchKernelGGL((linear_mfma_bf16_kernel<NT, true>), grid, dim3(kBlock), 0, st, x, ldx, w, w_is_km, bias, act, y,
                       ldy, n, k, m);
  else
    hipLaunchKernelGGL((linear_mfma_bf16_kernel<NT, false>), grid, dim3(kBlock), 0, st, x, ldx, w, w_is_km, bias, act,
                       y, ldy, n, k, m);
}
}  // namespace

LPG_EXPORT int lpg_linear_bf16(const uint16_t* x, int64_t ldx, const uint16_t* w, int w_is_km, const float* bias,
                               int act, uint16_t* y, int64_t ldy, int64_t n, int64_t k, int64_t m, void* stream) {
  LPG_REQUIRE(act >= LPG_ACT_NONE && act <= LPG_ACT_TANH, "linear_bf16: unknown activation");
  LPG_REQUIRE(n >= 0 && k >= 1 && m >= 1 && ldx >= 1 && ldy >= m, "linear_bf16: bad sizes");
  if (n == 0) return LPG_OK;
  LPG_REQUIRE(x && w && y, "linear_bf16: null pointer");
  LPG_REQUIRE(ceil_div(n, 128) < (int64_t(1) << 31) && ceil_div(m, 32) < 65536, "linear_bf16: grid too large");
  const int vec_ok = k % 8 == 0 && ldx % 8 == 0 && m % 8 == 0 && ldy % 8 == 0 &&
                     reinterpret_cast<uintptr_t>(x) % 16 == 0 && reinterpret_cast<uintptr_t>(w) % 16 == 0 &&
                     reinterpret_cast<uintptr_t>(y) % 16 == 0;
  hipStream_t st = as_stream(stream);
  {
    const int r = linear_wres_bf16(x, ldx, w, w_is_km, bias, act, y, ldy, n, k, m, st);
    if (r < 0) return r;
    if (r > 0) return check_launch("linear_bf16 (W-resident)");
  }
  if (m <= 32) launch_bf16<1>(x, ldx, w, w_is_km, bias, act, y, ldy, n, k, m, vec_ok, st);
  else if (m <= 64) launch_bf16<2>(x, ldx, w, w_is_km, bias, act, y, ldy, n, k, m, vec_ok, st);
  else launch_bf16<4>(x, ldx, w, w_is_km, bias, act, y, ldy, n, k, m, vec_ok, st);
  return check_launch("linear_bf16");
}

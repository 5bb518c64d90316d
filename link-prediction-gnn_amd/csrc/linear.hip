// Dense feature linear (K5):  Y[n,m] = X[n,k] @ W^T (+ bias)   on the fp32-input matrix cores.
// Reference: GCNConv.lin (no bias, weight [out,in]) inside TwoWL/model/model.py:37 [3P PyG];
// SEAL's tf.matmul(X, W) with W [in,out] (SEAL/model/gnn.py:46,50,54,58) -> w_is_km.
//
// v_mfma_f32_32x32x2_f32: exact fp32 products, fp32 accumulate (bitwise a k-ordered fmaf chain),
// which keeps the 1e-5 parity bar without any reduced-precision path.  A wave owns a 32x32
// output tile; X and W tiles are staged through LDS with coalesced global reads and a +1 padded
// row so the per-lane fragment reads (row = lane&31, k = 2*step + lane>>5) are conflict-free.
#include "lpg_common.hpp"

namespace {
using namespace lpg;

using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int KT = 64;       // k values staged per step
constexpr int LDP = KT + 1;  // padded LDS row

// SPLITK = false: the 4 waves of a workgroup own 4 different 32-row tiles and each walks all of K
//                 (k <= 64: one staging step, nothing to split).
// SPLITK = true : the 4 waves share ONE 32-row tile and take every 4th K-slice; the partial
//                 accumulators are folded through LDS in wave order (fixed order => reproducible).
//                 4x the workgroups and a quarter of the dependent staging steps per wave: at the
//                 reference's sizes ([620,256]x[256,32]) the op is latency bound, not flop bound.
template <bool SPLITK>
__global__ __launch_bounds__(kBlock) void linear_mfma_f32_kernel(const float* __restrict__ x, int64_t ldx,
                                                                 const float* __restrict__ w, int w_is_km,
                                                                 const float* __restrict__ bias, int act,
                                                                 float* __restrict__ y, int64_t ldy, int64_t n,
                                                                 int64_t k, int64_t m) {
  __shared__ float Xs[4][32][LDP];
  __shared__ float Ws[SPLITK ? 4 : 1][32][LDP];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t row0 = SPLITK ? static_cast<int64_t>(blockIdx.x) * 32 : static_cast<int64_t>(blockIdx.x) * 128 + wave * 32;
  const int64_t j0 = static_cast<int64_t>(blockIdx.y) * 32;
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  const int fr = lane & 31, fk = lane >> 5;

  if constexpr (SPLITK) {
    for (int64_t k0 = static_cast<int64_t>(wave) * KT; k0 < k; k0 += 4 * KT) {
      // every wave stages its own X slice and its own W slice: no workgroup barrier in the loop.
      // Two phases with a compiler barrier between them: all 64 loads are issued from clamped
      // addresses first, the selects and LDS writes follow (written as `cond ? load : 0`, hipcc
      // branches around every load and waits vmcnt(0) 64 times -- 64 dependent round trips).
      float xr[32], wr[32];
#pragma unroll
      for (int it = 0; it < 32; ++it) {
        const int64_t r = row0 + it, kk = k0 + lane;
        xr[it] = x[(r < n ? r : n - 1) * ldx + (kk < k ? kk : k - 1)];
      }
#pragma unroll
      for (int it = 0; it < 32; ++it) {
        if (w_is_km) {            // W[k][m]: lanes 0-31 walk m for k-row 2*it + fk
          const bool ok = k0 + 2 * it + fk < k && j0 + fr < m;
          wr[it] = w[(ok ? k0 + 2 * it + fk : 0) * m + (ok ? j0 + fr : 0)];
        } else {                  // W[m][k]: lanes walk k for output column `it`
          const bool ok = k0 + lane < k && j0 + it < m;
          wr[it] = w[(ok ? j0 + it : 0) * k + (ok ? k0 + lane : 0)];
        }
      }
      asm volatile("" ::: "memory");
#pragma unroll
      for (int it = 0; it < 32; ++it) {
        const int64_t r = row0 + it, kk = k0 + lane;
        Xs[wave][it][lane] = (r < n && kk < k) ? xr[it] : 0.f;
        if (w_is_km)
          Ws[wave][fr][2 * it + fk] = (k0 + 2 * it + fk < k && j0 + fr < m) ? wr[it] : 0.f;
        else
          Ws[wave][it][lane] = (k0 + lane < k && j0 + it < m) ? wr[it] : 0.f;
      }
      __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0): this wave's LDS writes are done (wave-private slices)
      __builtin_amdgcn_wave_barrier();
#pragma unroll 8
      for (int kk = 0; kk < KT; kk += 2)
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(Xs[wave][fr][kk + fk], Ws[wave][fr][kk + fk], acc, 0, 0, 0);
      __builtin_amdgcn_wave_barrier();
    }
    // fold the 4 partial tiles in wave order through LDS (reuse Xs as [4][16][64] floats)
    float* red = &Xs[0][0][0];
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) red[(wave * 16 + r) * 64 + lane] = acc[r];
    __syncthreads();
    if (wave == 0) {
#pragma unroll
      for (int r = 0; r < 16; ++r)
        acc[r] = ((red[(0 * 16 + r) * 64 + lane] + red[(1 * 16 + r) * 64 + lane]) + red[(2 * 16 + r) * 64 + lane]) +
                 red[(3 * 16 + r) * 64 + lane];
    } else {
      return;
    }
  } else {
    for (int64_t k0 = 0; k0 < k; k0 += KT) {
      // X tile: each wave stages its own 32 rows (one 256-byte row segment per instruction);
      // W tile (32 output columns x KT) is shared by the 4 waves.  Loads first, selects after.
      float xr[32], wr[(32 * KT) / kBlock];
#pragma unroll
      for (int it = 0; it < 32; ++it) {
        const int64_t r = row0 + it, kk = k0 + lane;
        xr[it] = x[(r < n ? r : n - 1) * ldx + (kk < k ? kk : k - 1)];
      }
#pragma unroll
      for (int it = 0; it < (32 * KT) / kBlock; ++it) {
        const int idx = it * kBlock + tid;
        const int kk = w_is_km ? idx >> 5 : idx & 63, j = w_is_km ? idx & 31 : idx >> 6;
        const bool ok = k0 + kk < k && j0 + j < m;
        const int64_t off = w_is_km ? (k0 + kk) * m + j0 + j : (j0 + j) * k + k0 + kk;
        wr[it] = w[ok ? off : 0];
      }
      asm volatile("" ::: "memory");
#pragma unroll
      for (int it = 0; it < 32; ++it) {
        const int64_t r = row0 + it, kk = k0 + lane;
        Xs[wave][it][lane] = (r < n && kk < k) ? xr[it] : 0.f;
      }
#pragma unroll
      for (int it = 0; it < (32 * KT) / kBlock; ++it) {
        const int idx = it * kBlock + tid;
        const int kk = w_is_km ? idx >> 5 : idx & 63, j = w_is_km ? idx & 31 : idx >> 6;
        Ws[0][j][kk] = (k0 + kk < k && j0 + j < m) ? wr[it] : 0.f;
      }
      __syncthreads();
#pragma unroll 8
      for (int kk = 0; kk < KT; kk += 2)
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(Xs[wave][fr][kk + fk], Ws[0][fr][kk + fk], acc, 0, 0, 0);
      __syncthreads();
    }
  }
  // C/D layout: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
  const int64_t col = j0 + (lane & 31);
  if (col < m) {
    const float b = bias ? bias[col] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int64_t row = row0 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      if (row < n) y[row * ldy + col] = apply_act(acc[r] + b, act);
    }
  }
}
}  // namespace

LPG_EXPORT int lpg_linear_f32(const float* x, int64_t ldx, const float* w, int w_is_km, const float* bias, int act,
                              float* y, int64_t ldy, int64_t n, int64_t k, int64_t m, void* stream) {
  LPG_REQUIRE(act >= LPG_ACT_NONE && act <= LPG_ACT_TANH, "linear: unknown activation");
  LPG_REQUIRE(n >= 0 && k >= 1 && m >= 1 && ldx >= 1 && ldy >= m, "linear: bad sizes n=%lld k=%lld m=%lld",
              (long long)n, (long long)k, (long long)m);
  if (n == 0) return LPG_OK;
  LPG_REQUIRE(x && w && y, "linear: null pointer");
  // split K across the 4 waves when there is more than one K-slice and the problem is small enough
  // to be latency bound (few row tiles); big problems keep 128 rows per workgroup.
  const bool splitk = k > KT && n <= (int64_t(1) << 16);
  const int64_t gx = ceil_div(n, splitk ? 32 : 128), gy = ceil_div(m, 32);
  LPG_REQUIRE(gx < (int64_t(1) << 31) && gy < 65536, "linear: grid too large");
  const dim3 grid(static_cast<unsigned>(gx), static_cast<unsigned>(gy));
  if (splitk)
    hipLaunchKernelGGL(linear_mfma_f32_kernel<true>, grid, dim3(kBlock), 0, as_stream(stream), x, ldx, w, w_is_km, bias,
                       act, y, ldy, n, k, m);
  else
    hipLaunchKernelGGL(linear_mfma_f32_kernel<false>, grid, dim3(kBlock), 0, as_stream(stream), x, ldx, w, w_is_km, bias,
                       act, y, ldy, n, k, m);
  return check_launch("linear");
}

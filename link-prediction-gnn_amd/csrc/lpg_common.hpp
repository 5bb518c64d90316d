// Shared helpers for the gfx950 kernels of liblpg_hip.so (see include/lpg_hip.h for the ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/lpg_hip.h"

#define LPG_EXPORT extern "C" __attribute__((visibility("default")))

namespace lpg {

constexpr int kWave = 64;   // CDNA4 wavefront
constexpr int kBlock = 256; // 4 waves per workgroup, one per SIMD

void set_error(const char* fmt, ...);

// Process-wide kernel-selection options (csrc/options.hip; lpg_set_option / lpg_get_option / lpg_reset_options in
// the ABI).  Plain loads at launch time: no getenv on any launch path.  Every option selects between code paths that
// all return CORRECT results (tests pin each variant to the oracle); the measurement-only variants that do not
// (spmm_mode 2: gather-only upper bound) exist only in a -DLPG_TUNING build of the library, which additionally
// seeds the table once, at load, from LPG_* environment variables for the sweep scripts under profiles/.
struct Options {
  int64_t spmm_chunk = 0;        // edges per SpMM chunk (multiple of 16); 0 = automatic (chunk_for)
  int64_t spmm_mode = 0;         // 0 default; 1 index-prefetch variant; 2 gather-only (LPG_TUNING builds only)
  int64_t spmm_lds = -1;         // LDS bytes reserved per workgroup (occupancy limiter); -1 = automatic
  int64_t spmm_block = 0;        // threads per SpMM workgroup: 0 automatic, or 64 / 128 / 256
  int64_t spmm_off32 = 1;        // 32-bit gather offsets when X spans < 4 GiB
  int64_t spmm_rows_max = 32;    // longest row for which the row-parallel one-launch SpMM is taken (0 disables)
  int64_t spmm_rows_ecap = int64_t(1) << 21;
  int64_t spmm_fill_overlap = 1; // 0: the empty-row fill of big graphs runs after the fix-up on the caller's stream instead of beside the main kernel
  int64_t spmm_no_edge = 0;      // 1: narrow features (d <= 16) stay on the main kernel
  int64_t seal_conv_fp32 = 0;    // 1: the fp32-MFMA formulation of the SEAL conv stack (no head fusion)
  int64_t seal_conv_variant = 0; // 0: every product on the bf16 matrix cores (csrc/seal_conv2.hip); 1: the r01 kernel (fp32 MFMA dense products)
  int64_t gn_small = 1;          // 0: GraphNorm keeps the three-launch path for small n
  int64_t linear_percu_delta = 0;
  int64_t linear_wres = -1;      // -1 automatic; 0 off; 1 / 2 / 3 force that W-resident variant (also below n_min)
  int64_t spmm_phase_timing = 0; // 1: record events around the SpMM's kernels (lpg_csr_spmm_phase_ms)
  int64_t seal_pair_small = 1;   // 0: every candidate pair takes the 128-thread extraction workgroup
  int64_t linear_x3 = -1;        // -1 automatic; 0 off; 1 force the strip form, 2 the tiled form: fp32 linear as exact bf16 plane products on the bf16 matrix cores (linear_x3.hip)
  int64_t stat_linear_x3 = 0;    // statistic: launches of that kernel so far
  int64_t stat_linear_wres = 0;  // statistic: W-resident linear launches so far (tests assert the path taken; set 0 to reset)
};
const Options& options();
Options& options_mut();

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return LPG_E_LAUNCH;
  }
  return LPG_OK;
}

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }
inline int64_t align_up(int64_t a, int64_t b) { return ceil_div(a, b) * b; }

// grid for a grid-stride elementwise kernel: enough workgroups to fill 256 CUs x 8, no more
inline int grid_for(int64_t work_items, int block = kBlock, int64_t cap = 256 * 8) {
  int64_t g = ceil_div(work_items, block);
  if (g < 1) g = 1;
  if (g > cap) g = cap;
  return static_cast<int>(g);
}

#define LPG_REQUIRE(cond, ...)            \
  do {                                    \
    if (!(cond)) {                        \
      ::lpg::set_error(__VA_ARGS__);      \
      return LPG_E_INVALID;               \
    }                                     \
  } while (0)

#define LPG_HIP(call)                                                        \
  do {                                                                       \
    hipError_t e_ = (call);                                                  \
    if (e_ != hipSuccess) {                                                  \
      ::lpg::set_error("%s failed: %s", #call, hipGetErrorString(e_));       \
      return LPG_E_LAUNCH;                                                   \
    }                                                                        \
  } while (0)

// linear_wres.hip: persistent W-resident kernels for big n; return 1 = launched, 0 = shape does not
// qualify (the caller falls through to the tiled kernels), < 0 = error
int linear_wres_f32(const float* x, int64_t ldx, const float* w, int w_is_km, const float* bias, int act, float* y,
                    int64_t ldy, int64_t n, int64_t k, int64_t m, hipStream_t st);
int linear_wres_bf16(const uint16_t* x, int64_t ldx, const uint16_t* w, int w_is_km, const float* bias, int act,
                     uint16_t* y, int64_t ldy, int64_t n, int64_t k, int64_t m, hipStream_t st);

// linear_x3.hip: fp32 linear on the bf16 matrix cores through the exact three-plane split; same return convention
int linear_x3_f32(const float* x, int64_t ldx, const float* w, int w_is_km, const float* bias, int act, float* y,
                  int64_t ldy, int64_t n, int64_t k, int64_t m, hipStream_t st);

// seal_conv2.hip: the SEAL conv stack with every product on the bf16 matrix cores (fp32-class plane arithmetic)
int seal_conv2_f32(const int64_t* node_off, const int32_t* first, int64_t n_windows, const uint32_t* adj_bits,
                   const float* h1, int64_t ldh, const float* w2, const float* w3, const float* w4, float* cat, int64_t ldc,
                   float* key, const float* head_k1, float* head_a16, hipStream_t st);

// streaming 16-byte store (global_store_dwordx4 ... nt): for outputs much larger than the caches a pure write stream
// runs at 5.3 TB/s instead of 4.1 (embedding, n = 5 M x 128 floats); the line still stays in the XCD's L2 for a
// consumer that comes right after (MI355X guide, "stores of each flavour")
__device__ __forceinline__ void store_stream_u4(void* p, uint4 v) {
  using u4 = __attribute__((ext_vector_type(4))) uint32_t;
  const u4 q = {v.x, v.y, v.z, v.w};
  __builtin_nontemporal_store(q, reinterpret_cast<u4*>(p));
}
__device__ __forceinline__ void store_stream(float* p, float4 v) {
  using f4 = __attribute__((ext_vector_type(4))) float;
  f4 q = {v.x, v.y, v.z, v.w};
  __builtin_nontemporal_store(q, reinterpret_cast<f4*>(p));
}

__device__ __forceinline__ void flag_error(int32_t* err_flag, int32_t bit) {
  if (err_flag) atomicOr(err_flag, bit);
}

// tanh through one v_exp_f32: tanh(v) = 1 - 2 / (exp(2v) + 1).  libm's tanhf inlines ~60
// instructions at every call site (the SpMM flushes rows at 8 unrolled sites); this form is ~6 and
// accurate to ~2e-7 absolute on |tanh| <= 1 (parity bar 1e-5).  Saturates cleanly: exp -> inf gives 1,
// exp -> 0 gives -1.
__device__ __forceinline__ float fast_tanh(float v) {
  const float v2 = v * v;
  if (v2 < 0.0625f)   // |v| < 0.25: odd series keeps RELATIVE accuracy where 1 - 2/(e+1) cancels
    return v * (1.0f + v2 * (-0.33333334f + v2 * (0.13333334f + v2 * (-0.053968254f + v2 * 0.021869488f))));
  const float e = __expf(2.0f * v);
  return 1.0f - 2.0f / (e + 1.0f);
}

__device__ __forceinline__ float apply_act(float v, int act) {
  if (act == LPG_ACT_RELU) return v > 0.f ? v : 0.f;
  if (act == LPG_ACT_TANH) return fast_tanh(v);
  return v;
}

// bf16 <-> f32 (round to nearest even; plain casts keep NaN a NaN, see MI355X guide)
__device__ __forceinline__ float bf16_to_f32(uint16_t b) { return __uint_as_float(static_cast<uint32_t>(b) << 16); }
// round to nearest even through gfx950's own conversion (v_cvt_pk_bf16_f32: ONE instruction; the bit-twiddled form --
// NaN test, + 0x7fff + lsb, shift -- was six, sixteen times per tile and layer in the VALU-bound SEAL stack).  Same
// result for every non-NaN input; NaN stays a (quiet) NaN.
__device__ __forceinline__ uint16_t f32_to_bf16(float f) { return __builtin_bit_cast(uint16_t, static_cast<__bf16>(f)); }
// two values -> one dword (a in the low half): one v_cvt_pk_bf16_f32 for the pair
__device__ __forceinline__ uint32_t pack_bf16x2(float a, float b) {
  using f32x2 = __attribute__((ext_vector_type(2))) float;
  using bf16x2 = __attribute__((ext_vector_type(2))) __bf16;
  const f32x2 v = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
}

}  // namespace lpg

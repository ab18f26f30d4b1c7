// Device helpers of the "f16mx" operand format (include/wsovod_hip.h, WSOVOD_F16MX; round 6): a row is groups of 32 values =
// 128 bytes  [32 x fp16 hi | 32 x OCP e4m3 q | 32 x e4m3 ql],  q = e4m3(x 2^-s),  ql = e4m3((x - hi) 2^-(s - 11)).
// Activations are written with the UNIT scale s = 0 (no scale array, no row maximum in the producing epilogue: e4m3's own
// exponent spans 2^-9 .. 448, values beyond saturate and then cost the cross terms' accuracy, not the product's); weights
// carry one E8M0 byte per row (gemm8mx.hip: wsovod_f16mx_encode).
#pragma once
#include "common.h"

namespace wsovod_mx {

typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef int i32x2 __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

// byte offset of value k's group inside an f16mx row; the three planes of the group sit at +2w, +64 + w, +96 + w (w = k & 31)
__device__ __forceinline__ long long mx_group(long long k) { return (k >> 5) << 7; }

// (the builtins exist in the device pass only; the host pass sees the same declarations with inert bodies)
// mx_sat: v_cvt_pk_fp8_f32 does NOT saturate on gfx950 -- probed: 449 -> 0x7e, 480 and everything above -> 0x7f (NaN) -- so the
// value is clamped to the format's +-448 first
__device__ __forceinline__ float mx_sat(float v) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_fmed3f(v, -448.0f, 448.0f);
#else
  return v;
#endif
}

// four values -> their hi halves and the two packed e4m3 words, scales 2^-sq / 2^-sl given as multipliers
__device__ __forceinline__ void mx_enc4(const f32x4 v, float inv_q, float inv_l, f16x4& hi, int& q, int& ql) {
  float lo[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    hi[j] = (_Float16)v[j];
    const float h = (float)hi[j];
    lo[j] = __builtin_isinf(h) ? 0.f : v[j] - h;  // (beyond the fp16 range the value stays infinite; inf - inf would be NaN)
  }
#if defined(__HIP_DEVICE_COMPILE__)
  q = __builtin_amdgcn_cvt_pk_fp8_f32(mx_sat(v[0] * inv_q), mx_sat(v[1] * inv_q), 0, false);
  q = __builtin_amdgcn_cvt_pk_fp8_f32(mx_sat(v[2] * inv_q), mx_sat(v[3] * inv_q), q, true);
  ql = __builtin_amdgcn_cvt_pk_fp8_f32(mx_sat(lo[0] * inv_l), mx_sat(lo[1] * inv_l), 0, false);
  ql = __builtin_amdgcn_cvt_pk_fp8_f32(mx_sat(lo[2] * inv_l), mx_sat(lo[3] * inv_l), ql, true);
#else
  q = ql = 0;
  (void)inv_q; (void)inv_l; (void)lo;
#endif
}
__device__ __forceinline__ void mx_enc4_unit(const f32x4 v, f16x4& hi, int& q, int& ql) { mx_enc4(v, 1.0f, 2048.0f, hi, q, ql); }

// the value a unit-scale f16mx element stands for: hi + ql 2^-11 (what a residual / a decoder reads)
__device__ __forceinline__ f32x4 mx_dec4_unit(const f16x4 hi, const int ql) {
#if !defined(__HIP_DEVICE_COMPILE__)
  (void)ql;
  return f32x4{(float)hi[0], (float)hi[1], (float)hi[2], (float)hi[3]};
#else
  return f32x4{(float)hi[0] + __builtin_amdgcn_cvt_f32_fp8(ql, 0) * (1.0f / 2048.0f),
               (float)hi[1] + __builtin_amdgcn_cvt_f32_fp8(ql, 1) * (1.0f / 2048.0f),
               (float)hi[2] + __builtin_amdgcn_cvt_f32_fp8(ql, 2) * (1.0f / 2048.0f),
               (float)hi[3] + __builtin_amdgcn_cvt_f32_fp8(ql, 3) * (1.0f / 2048.0f)};
#endif
}
// four consecutive values k .. k + 3 (k a multiple of 4) of the unit-scale f16mx row at `row`
__device__ __forceinline__ f32x4 mx_load4_unit(const char* row, int k) {
  const char* g = row + mx_group(k);
  const int w = k & 31;
  return mx_dec4_unit(*(const f16x4*)(g + 2 * w), *(const int*)(g + 96 + w));
}
__device__ __forceinline__ void mx_store4_unit(char* row, int k, const f32x4 v) {
  f16x4 hi;
  int q, ql;
  mx_enc4_unit(v, hi, q, ql);
  char* g = row + mx_group(k);
  const int w = k & 31;
  *(f16x4*)(g + 2 * w) = hi;
  *(int*)(g + 64 + w) = q;
  *(int*)(g + 96 + w) = ql;
}

}  // namespace wsovod_mx

// Shared helpers for the wsovod_hip C-ABI library (gfx950 / CDNA4 only).
//
// Error convention mirrors the reference's native layer (AT_ASSERTM / AT_ERROR ->
// RuntimeError, wsovod/layers/ROILoopPool/ROILoopPool_cuda.cu:258-265,311): every
// entry point returns 0 on success or a non-zero code, and the text of the last
// error on the calling thread is available through wsovod_last_error().
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/wsovod_hip.h"

namespace wsovod {

void set_error(const char* fmt, ...);

// Per-kernel profiling table (hipEvent bracketed launches; see runtime.hip).
int prof_slot(const char* name);  // stable slot id for a kernel family name
struct ProfScope {
  int id;
  hipStream_t stream;
  hipEvent_t e0, e1;
  bool on;
  ProfScope(int kernel_id, hipStream_t s, double flops, double bytes);
  ~ProfScope();
};

// bf16x2 forms of elementwise entry points (bf16x2.hip), reached through the dtype argument of the public functions
int x2_maxpool2x2(const void* in, int N, int H, int W, int C, int Ho, int Wo, int stride, int zero_pad, void* out, hipStream_t s);
int x2_add_group_rows(const void* x, long long ldx, const int* row_group, const float* add, long long ld_add, int M, int N,
                      void* out, long long ldo, hipStream_t s);

}  // namespace wsovod

#define WS_CHECK_ARG(cond, ...)                 \
  do {                                          \
    if (!(cond)) {                              \
      wsovod::set_error(__VA_ARGS__);           \
      return WSOVOD_ERR_INVALID_ARGUMENT;       \
    }                                           \
  } while (0)

#define WS_CHECK_LAUNCH(name)                                                     \
  do {                                                                            \
    hipError_t e_ = hipGetLastError();                                            \
    if (e_ != hipSuccess) {                                                       \
      wsovod::set_error("%s: launch failed: %s", name, hipGetErrorString(e_));    \
      return WSOVOD_ERR_HIP;                                                      \
    }                                                                             \
  } while (0)

// A HIP runtime call whose status must not be lost (hipFuncSetAttribute for > 64 KiB of LDS, ...): a refusal is reported
// through the library's error convention at the call site instead of surfacing later as an unexplained launch failure.
#define WS_CHECK_HIP(expr, what)                                                             \
  do {                                                                                       \
    hipError_t e_ = (expr);                                                                  \
    if (e_ != hipSuccess) {                                                                  \
      (void)hipGetLastError();                                                               \
      wsovod::set_error("%s: %s failed: %s", what, #expr, hipGetErrorString(e_));            \
      return WSOVOD_ERR_HIP;                                                                 \
    }                                                                                        \
  } while (0)

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__device__ __forceinline__ float to_f32(float v) { return v; }
__device__ __forceinline__ float to_f32(bf16_t v) { return (float)v; }
template <typename T>
__device__ __forceinline__ T from_f32(float v);
template <>
__device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <>
__device__ __forceinline__ bf16_t from_f32<bf16_t>(float v) { return (bf16_t)v; }

__device__ __forceinline__ float wave_reduce_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_reduce_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

__host__ __device__ static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
static inline long long ceil_div_ll(long long a, long long b) { return (a + b - 1) / b; }

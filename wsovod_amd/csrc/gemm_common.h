// Pieces shared by the GEMM / implicit-GEMM kernels (gemm.hip: generic tiles; gemm8.hip: the 8-phase 256x256 tile).
#pragma once
#include "common.h"

namespace wsovod_gemm {

struct GemmArgs {
  const char* A;
  const char* B;
  long long lda, ldb;  // elements
  int M, N, K;
  void* C;
  long long ldc;
  int dtype_c;
  void* Ct;
  long long ldct;
  int dtype_ct;
  float alpha;
  const float* row_scale;
  const float* bias;
  const void* residual;
  long long ldr;
  int dtype_r;
  int relu;
  float dropout_p;
  unsigned long long seed;
  const int* row_group;
  const float* group_add;
  long long ld_ga;
  const void* mask_src;
  long long ldm;
  int dtype_m;
  float mask_scale;
  int accumulate;
  // implicit-GEMM convolution geometry
  int H, W, Cin, Ho, Wo, KH, KW, stride, pad, dil;
  int pool;  // conv3x3_c64 only: 2 = MaxPool2d(2, 2) fused into the epilogue, C is the pooled map
  // split-K (gemm8.hip, plain GEMMs with few output tiles): the grid is ksplit copies of the tile grid, copy z reduces
  // K-steps [z*slice_steps, (z+1)*slice_steps) and stores its raw fp32 sums to partial[z][M][partial_ld]; a finalize
  // kernel adds the copies and applies the epilogue
  int ksplit, slice_steps;
  float* partial;
  long long partial_ld;
  long long a_bytes;  // conv: byte size of the NHWC input (must be < 2^31)
  int tiles_m, tiles_n;
  int group_m;  // tile-order group height (see the XCD remap in the kernel)
  // conv: optional second input contracted 1x1 / stride 1 after the KH*KW*Cin main K range (fused projection shortcut)
  const char* A2;
  int Cin2;
  long long a2_bytes;
};

template <typename T>
struct Traits;
template <>
struct Traits<float> {
  static constexpr int EPC = 4;   // elements per 16-byte chunk
  static constexpr int BKE = 32;  // elements per K-step (128 B)
};
template <>
struct Traits<bf16_t> {
  static constexpr int EPC = 8;
  static constexpr int BKE = 64;
};

__device__ __forceinline__ int lds_off(int row, int chunk) {
  return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4);
}

__device__ __forceinline__ float load_as_f32(const void* p, long long idx, int dtype) {
  return dtype == WSOVOD_BF16 ? (float)((const bf16_t*)p)[idx] : ((const float*)p)[idx];
}
__device__ __forceinline__ void store_from_f32(void* p, long long idx, int dtype, float v) {
  if (dtype == WSOVOD_BF16)
    ((bf16_t*)p)[idx] = (bf16_t)v;
  else
    ((float*)p)[idx] = v;
}

// splitmix64 finaliser: counter-based, stateless dropout mask on (seed, m, n)
__device__ __forceinline__ float uniform01(unsigned long long seed, unsigned long long ctr) {
  unsigned long long z = seed + 0x9E3779B97F4A7C15ull * (ctr + 1);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z = z ^ (z >> 31);
  return (float)(z >> 40) * (1.0f / 16777216.0f);
}


// The epilogue chain of the GEMM kernels for 4 consecutive columns nb..nb+3 of row m (v = raw sums), stores included:
// used by the split-K finalize kernels (the tile kernels carry vectorised copies of the same chain).
__device__ __forceinline__ void epilogue_store4(const GemmArgs& p, int m, int nb, const float (&v)[4]) {
  const float keep_scale = p.dropout_p > 0.f ? 1.0f / (1.0f - p.dropout_p) : 1.0f;
  const float rs = p.row_scale ? p.row_scale[m] : 1.f;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int n = nb + r;
    if (n >= p.N) continue;
    float x = v[r] * p.alpha;
    if (p.row_scale) x *= rs;
    if (p.bias) x += p.bias[n];
    if (p.residual) x += load_as_f32(p.residual, (long long)m * p.ldr + n, p.dtype_r);
    if (p.relu) x = fmaxf(x, 0.f);
    if (p.dropout_p > 0.f) {
      const float u = uniform01(p.seed, (unsigned long long)m * (unsigned long long)p.N + n);
      x = u >= p.dropout_p ? x * keep_scale : 0.f;
    }
    if (p.group_add) x += p.group_add[(long long)p.row_group[m] * p.ld_ga + n];
    if (p.mask_src) x = load_as_f32(p.mask_src, (long long)m * p.ldm + n, p.dtype_m) > 0.f ? x * p.mask_scale : 0.f;
    if (p.C && p.accumulate) x += ((float*)p.C)[(long long)m * p.ldc + n];
    if (p.C) store_from_f32(p.C, (long long)m * p.ldc + n, p.dtype_c, x);
    if (p.Ct) store_from_f32(p.Ct, (long long)n * p.ldct + m, p.dtype_ct, x);
  }
}

// gemm8.hip: bf16 256x256 tile, 8 wavefronts in two staggered groups (see the file header).
int launch_gemm256_8ph(const GemmArgs& a, bool conv, hipStream_t s, double flops, double bytes, bool allow_split = false);

}  // namespace wsovod_gemm

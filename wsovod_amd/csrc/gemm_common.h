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
  const unsigned long long* seed_add;  // optional device scalar added to `seed` (a step counter kept on the device)
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
  long long a_plane;  // plain GEMM on a PLANAR bf16x2 A (WSOVOD_BF16X2P): bytes from the hi plane to the lo plane, 0 = interleaved
  int tiles_m, tiles_n;
  int m_base;  // gemm_nt_kernel: first output row of this launch (a launch may cover rows [m_base, M) only: tail launches)
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

// ---- bf16x2 (include/wsovod_hip.h: WSOVOD_BF16X2): value k of a row sits at bf16 index x2_pos(k) (hi) and
// x2_pos(k) + 32 (lo) of the row's 2 * ld bf16 slots
__device__ __forceinline__ long long x2_pos(int k) { return ((long long)(k >> 5) << 6) | (k & 31); }
__device__ __forceinline__ bf16_t x2_lo(float v, bf16_t hi) {
  const float h = (float)hi;
  return (bf16_t)(__builtin_isinf(h) ? 0.f : v - h);  // an infinite value stays infinite (inf - inf would store a NaN)
}

// element (m, n) of a row-major matrix with leading dimension ld (values) in any of the three dtypes
__device__ __forceinline__ float load_as_f32(const void* p, long long m, long long ld, int n, int dtype) {
  if (dtype == WSOVOD_BF16) return (float)((const bf16_t*)p)[m * ld + n];
  if (dtype == WSOVOD_BF16X2) {
    const bf16_t* q = (const bf16_t*)p + 2 * m * ld + x2_pos(n);
    return (float)q[0] + (float)q[32];
  }
  return ((const float*)p)[m * ld + n];
}
__device__ __forceinline__ void store_from_f32(void* p, long long m, long long ld, int n, int dtype, float v) {
  if (dtype == WSOVOD_BF16) {
    ((bf16_t*)p)[m * ld + n] = (bf16_t)v;
  } else if (dtype == WSOVOD_BF16X2) {
    bf16_t* q = (bf16_t*)p + 2 * m * ld + x2_pos(n);
    const bf16_t hi = (bf16_t)v;
    q[0] = hi;
    q[32] = x2_lo(v, hi);
  } else {
    ((float*)p)[m * ld + n] = v;
  }
}
// 4 consecutive values n .. n+3 (n a multiple of 4; rows 8-byte (bf16, bf16x2) / 16-byte (fp32) aligned)
__device__ __forceinline__ f32x4 load4_as_f32(const void* p, long long m, long long ld, int n, int dtype) {
  if (dtype == WSOVOD_BF16) {
    const bf16x4 r = *(const bf16x4*)((const bf16_t*)p + m * ld + n);
    return f32x4{(float)r[0], (float)r[1], (float)r[2], (float)r[3]};
  }
  if (dtype == WSOVOD_BF16X2) {
    const bf16_t* q = (const bf16_t*)p + 2 * m * ld + x2_pos(n);
    const bf16x4 h = *(const bf16x4*)q, l = *(const bf16x4*)(q + 32);
    return f32x4{(float)h[0] + (float)l[0], (float)h[1] + (float)l[1], (float)h[2] + (float)l[2], (float)h[3] + (float)l[3]};
  }
  return *(const f32x4*)((const float*)p + m * ld + n);
}
__device__ __forceinline__ void store4_from_f32(void* p, long long m, long long ld, int n, int dtype, const f32x4 v) {
  if (dtype == WSOVOD_BF16) {
    *(bf16x4*)((bf16_t*)p + m * ld + n) = bf16x4{(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
  } else if (dtype == WSOVOD_BF16X2) {
    bf16_t* q = (bf16_t*)p + 2 * m * ld + x2_pos(n);
    const bf16x4 h = bf16x4{(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
    *(bf16x4*)q = h;
    *(bf16x4*)(q + 32) = bf16x4{x2_lo(v[0], h[0]), x2_lo(v[1], h[1]), x2_lo(v[2], h[2]), x2_lo(v[3], h[3])};
  } else {
    *(f32x4*)((float*)p + m * ld + n) = v;
  }
}
// true when 4-value groups of rows of `p` can move as the vector accesses above
__device__ __forceinline__ bool vec4_ok(const void* p, long long ld, int dtype) {
  if (dtype == WSOVOD_F32) return (ld & 3) == 0 && ((uintptr_t)p & 15) == 0;
  if (dtype == WSOVOD_BF16) return (ld & 3) == 0 && ((uintptr_t)p & 7) == 0;
  return (ld & 31) == 0 && ((uintptr_t)p & 15) == 0;  // bf16x2: whole 32-value groups per row
}

// the dropout seed of this launch: the host part plus, when given, the device-resident step term (a captured HIP graph
// replays with the same kernel arguments: the part of the seed that changes per step lives in memory)
#define WS_DROPOUT_SEED(p) ((p).seed_add ? (p).seed + *(p).seed_add : (p).seed)

// Counter-based, stateless dropout mask on (seed, m, n): ONE splitmix64 value per quad of consecutive columns
// (n & ~3 .. n | 3 of row m), 16 of its bits per element: keep iff bits >= p * 2^16 (round 5: a 64-bit hash per element was
// ~40 VALU instructions each, ~0.18 ms of the step in the FC epilogues; p is honoured to 2^-16, 0.5 exactly)
__device__ __forceinline__ unsigned long long dropout_quad(unsigned long long seed, long long m, int N, int nb) {
  const unsigned long long ctr = (unsigned long long)m * (unsigned long long)((N + 3) >> 2) + (unsigned long long)(nb >> 2);
  unsigned long long z = seed + 0x9E3779B97F4A7C15ull * (ctr + 1);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
__device__ __forceinline__ unsigned dropout_threshold(float p) { return (unsigned)(p * 65536.0f + 0.5f); }
__device__ __forceinline__ bool dropout_keep(unsigned long long z, int r, unsigned thr) {
  return (unsigned)((z >> (16 * r)) & 0xFFFFull) >= thr;
}

// The epilogue chain of the GEMM kernels for 4 consecutive columns nb..nb+3 of row m (v = raw sums), stores included:
// used by the split-K finalize kernels (the tile kernels carry vectorised copies of the same chain).
__device__ __forceinline__ void epilogue_store4(const GemmArgs& p, int m, int nb, const float (&v)[4]) {
  const float keep_scale = p.dropout_p > 0.f ? 1.0f / (1.0f - p.dropout_p) : 1.0f;
  const float rs = p.row_scale ? p.row_scale[m] : 1.f;
  const unsigned long long dz = p.dropout_p > 0.f ? dropout_quad(WS_DROPOUT_SEED(p), m, p.N, nb) : 0ull;
  const unsigned dthr = dropout_threshold(p.dropout_p);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int n = nb + r;
    if (n >= p.N) continue;
    float x = v[r] * p.alpha;
    if (p.row_scale) x *= rs;
    if (p.bias) x += p.bias[n];
    if (p.residual) x += load_as_f32(p.residual, m, p.ldr, n, p.dtype_r);
    if (p.relu) x = fmaxf(x, 0.f);
    if (p.dropout_p > 0.f) x = dropout_keep(dz, r, dthr) ? x * keep_scale : 0.f;
    if (p.group_add) x += p.group_add[(long long)p.row_group[m] * p.ld_ga + n];
    if (p.mask_src) x = load_as_f32(p.mask_src, m, p.ldm, n, p.dtype_m) > 0.f ? x * p.mask_scale : 0.f;
    if (p.C && p.accumulate) x += ((float*)p.C)[(long long)m * p.ldc + n];
    if (p.C) store_from_f32(p.C, m, p.ldc, n, p.dtype_c, x);
    if (p.Ct) store_from_f32(p.Ct, n, p.ldct, m, p.dtype_ct, x);
  }
}

// gemm8.hip: bf16 256x256 tile, 8 wavefronts in two staggered groups (see the file header).
// x3: bf16x2 operands (three-MFMA products); merged: the two-phase form of the K-step (tile_hint 2256256)
int launch_gemm256_8ph(const GemmArgs& a, bool conv, hipStream_t s, double flops, double bytes, bool allow_split = false,
                       bool x3 = false, bool merged = false);

}  // namespace wsovod_gemm

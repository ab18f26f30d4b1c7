// Elementwise kernels of the bf16x2 activation format (include/wsovod_hip.h: WSOVOD_BF16X2; MODEL.HIP.PRECISION =
// "parity"): every value is a (hi, lo) pair of bf16 numbers, hi = bf16(x), lo = bf16(x - hi), stored in groups of 32
// values as [32 hi | 32 lo] = one 128-byte line.  All kernels here are HBM streams: 16-byte accesses per lane, whole
// lines per 8 lanes.
//   wsovod_bf16x2_encode / _decode : fp32 <-> bf16x2 (weights once per optimizer step; tests)
//   maxpool2x2 on bf16x2 NHWC maps (resnet_wsl.py:85-92,408), called from wsovod_maxpool2x2_nhwc
//   add_group_rows on bf16x2 rows (roi_heads.py:762-763), called from wsovod_add_group_rows
#include "common.h"

namespace {

__device__ __forceinline__ bf16_t lo_of(float v, bf16_t hi) {
  const float h = (float)hi;
  return (bf16_t)(__builtin_isinf(h) ? 0.f : v - h);
}

// one thread = 8 consecutive values of one row (a quarter of a 32-group)
__global__ __launch_bounds__(256) void x2_encode_kernel(const float* __restrict__ src, long long ld_src, int rows, int cols,
                                                        bf16_t* __restrict__ dst, long long ld_dst) {
  const int cg = cols >> 3;
  const long long total = (long long)rows * cg;
  const bool al = ((ld_src & 3) == 0) && (((uintptr_t)src & 15) == 0);
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int r = (int)(i / cg), c = (int)(i - (long long)r * cg) * 8;
    const float* s = src + (long long)r * ld_src + c;
    float v[8];
    if (al) {
      const f32x4 a = *(const f32x4*)s, b = *(const f32x4*)(s + 4);
      v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3]; v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = s[j];
    }
    bf16x8 hi, lo;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      hi[j] = (bf16_t)v[j];
      lo[j] = lo_of(v[j], hi[j]);
    }
    bf16_t* d = dst + 2 * (long long)r * ld_dst + ((c >> 5) << 6) + (c & 31);
    *(bf16x8*)d = hi;
    *(bf16x8*)(d + 32) = lo;
  }
}

__global__ __launch_bounds__(256) void x2_decode_kernel(const bf16_t* __restrict__ src, long long ld_src, int rows, int cols,
                                                        float* __restrict__ dst, long long ld_dst) {
  const int cg = cols >> 3;
  const long long total = (long long)rows * cg;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int r = (int)(i / cg), c = (int)(i - (long long)r * cg) * 8;
    const bf16_t* s = src + 2 * (long long)r * ld_src + ((c >> 5) << 6) + (c & 31);
    const bf16x8 hi = *(const bf16x8*)s, lo = *(const bf16x8*)(s + 32);
    float* d = dst + (long long)r * ld_dst + c;
#pragma unroll
    for (int j = 0; j < 8; ++j) d[j] = (float)hi[j] + (float)lo[j];
  }
}

// 2x2 max pool, NHWC, C values per pixel (C % 32 == 0): the (hi, lo) pair of the largest hi + lo is copied
__global__ void x2_maxpool2x2_kernel(const bf16_t* __restrict__ in, int N, int H, int W, int C, int Ho, int Wo, int stride,
                                     int zero_pad, bf16_t* __restrict__ out) {
  const int cv = C >> 3;
  const long long total = (long long)N * Ho * Wo * cv;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % cv) * 8;
    const int wo = (int)((i / cv) % Wo);
    const int ho = (int)((i / ((long long)cv * Wo)) % Ho);
    const int n = (int)(i / ((long long)cv * Wo * Ho));
    const int slot = ((c >> 5) << 6) + (c & 31);
    bf16x8 bh, bl;
    float best[8];
    bool any = false;
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
      for (int dx = 0; dx < 2; ++dx) {
        const int h = ho * stride + dy, w = wo * stride + dx;
        bf16x8 vh, vl;
        if (h < H && w < W) {
          const bf16_t* p = in + (((long long)n * H + h) * W + w) * 2 * C + slot;
          vh = *(const bf16x8*)p;
          vl = *(const bf16x8*)(p + 32);
        } else if (zero_pad) {
#pragma unroll
          for (int j = 0; j < 8; ++j) vh[j] = vl[j] = (bf16_t)0.f;
        } else {
          continue;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float v = (float)vh[j] + (float)vl[j];
          if (!any || v > best[j]) {
            best[j] = v;
            bh[j] = vh[j];
            bl[j] = vl[j];
          }
        }
        any = true;
      }
    bf16_t* o = out + (((long long)n * Ho + ho) * Wo + wo) * 2 * C + slot;
    *(bf16x8*)o = bh;
    *(bf16x8*)(o + 32) = bl;
  }
}

// out[m] = x[m] + add[row_group[m]] on bf16x2 rows (N % 32 == 0): RB rows x 8 values per lane in flight
__global__ __launch_bounds__(256) void x2_add_group_rows_kernel(const bf16_t* __restrict__ x, long long ldx,
                                                                const int* __restrict__ row_group,
                                                                const float* __restrict__ add, long long lda, int M, int N,
                                                                bf16_t* __restrict__ out, long long ldo) {
  constexpr int RB = 8;
  for (int m0 = blockIdx.x * RB; m0 < M; m0 += gridDim.x * RB) {
    for (int n = threadIdx.x * 8; n < N; n += 256 * 8) {
      const int slot = ((n >> 5) << 6) + (n & 31);
      bf16x8 vh[RB], vl[RB];
#pragma unroll
      for (int r = 0; r < RB; ++r)
        if (m0 + r < M) {
          const bf16_t* p = x + 2 * (long long)(m0 + r) * ldx + slot;
          vh[r] = *(const bf16x8*)p;
          vl[r] = *(const bf16x8*)(p + 32);
        }
#pragma unroll
      for (int r = 0; r < RB; ++r) {
        if (m0 + r >= M) continue;
        const float* a = add + (long long)row_group[m0 + r] * lda + n;
        bf16x8 oh, ol;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float v = ((float)vh[r][j] + (float)vl[r][j]) + a[j];
          oh[j] = (bf16_t)v;
          ol[j] = lo_of(v, oh[j]);
        }
        bf16_t* o = out + 2 * (long long)(m0 + r) * ldo + slot;
        *(bf16x8*)o = oh;
        *(bf16x8*)(o + 32) = ol;
      }
    }
  }
}

int grid_for(long long total, int block) { return (int)std::min<long long>((total + block - 1) / block, 256 * 64); }

}  // namespace

namespace wsovod {

int x2_maxpool2x2(const void* in, int N, int H, int W, int C, int Ho, int Wo, int stride, int zero_pad, void* out,
                  hipStream_t s) {
  const long long total = (long long)N * Ho * Wo * (C / 8);
  hipLaunchKernelGGL(x2_maxpool2x2_kernel, dim3(grid_for(total, 256)), dim3(256), 0, s, (const bf16_t*)in, N, H, W, C, Ho, Wo,
                     stride, zero_pad, (bf16_t*)out);
  return 0;
}

int x2_add_group_rows(const void* x, long long ldx, const int* row_group, const float* add, long long ld_add, int M, int N,
                      void* out, long long ldo, hipStream_t s) {
  const int grid = std::min(ceil_div(M, 8), 1 << 20);
  hipLaunchKernelGGL(x2_add_group_rows_kernel, dim3(grid), dim3(256), 0, s, (const bf16_t*)x, ldx, row_group, add, ld_add, M,
                     N, (bf16_t*)out, ldo);
  return 0;
}

}  // namespace wsovod

extern "C" {

int wsovod_bf16x2_encode(const float* src, long long ld_src, int rows, int cols, void* dst, long long ld_dst,
                         wsovod_stream_t stream) {
  WS_CHECK_ARG(rows >= 0 && cols >= 0 && cols % 32 == 0, "wsovod_bf16x2_encode: cols=%d must be a multiple of 32", cols);
  if (rows == 0 || cols == 0) return WSOVOD_OK;
  WS_CHECK_ARG(src && dst && ld_src >= cols && ld_dst >= cols && ld_dst % 4 == 0 && ((uintptr_t)dst & 15) == 0,
               "wsovod_bf16x2_encode: bad pointer / leading dimension");
  static int slot = wsovod::prof_slot("bf16x2_encode");
  hipStream_t s = (hipStream_t)stream;
  wsovod::ProfScope prof(slot, s, 0.0, (double)rows * cols * 8.0);
  hipLaunchKernelGGL(x2_encode_kernel, dim3(grid_for((long long)rows * (cols / 8), 256)), dim3(256), 0, s, src, ld_src, rows,
                     cols, (bf16_t*)dst, ld_dst);
  WS_CHECK_LAUNCH("wsovod_bf16x2_encode");
  return WSOVOD_OK;
}

int wsovod_bf16x2_decode(const void* src, long long ld_src, int rows, int cols, float* dst, long long ld_dst,
                         wsovod_stream_t stream) {
  WS_CHECK_ARG(rows >= 0 && cols >= 0 && cols % 32 == 0, "wsovod_bf16x2_decode: cols=%d must be a multiple of 32", cols);
  if (rows == 0 || cols == 0) return WSOVOD_OK;
  WS_CHECK_ARG(src && dst && ld_src >= cols && ld_dst >= cols && ld_src % 4 == 0 && ((uintptr_t)src & 15) == 0,
               "wsovod_bf16x2_decode: bad pointer / leading dimension");
  static int slot = wsovod::prof_slot("bf16x2_decode");
  hipStream_t s = (hipStream_t)stream;
  wsovod::ProfScope prof(slot, s, 0.0, (double)rows * cols * 8.0);
  hipLaunchKernelGGL(x2_decode_kernel, dim3(grid_for((long long)rows * (cols / 8), 256)), dim3(256), 0, s, (const bf16_t*)src,
                     ld_src, rows, cols, dst, ld_dst);
  WS_CHECK_LAUNCH("wsovod_bf16x2_decode");
  return WSOVOD_OK;
}

}  // extern "C"

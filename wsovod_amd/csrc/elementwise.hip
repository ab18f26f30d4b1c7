// HBM-bound helper kernels of the hot path: image preprocessing + stem im2col, NHWC max pool,
// global average pool, transpose/cast (weight shadows, X^T for dW), row L2 norm (+backward),
// segmented column sums (bias / data-aware grads), device-scalar scaling, fused SGD-momentum.
// All are coalesced 16-B-per-lane streaming kernels; none is reshaped into a GEMM.
#include "common.h"
#include "f16mx.h"

namespace {

constexpr int kMaxGrid = 256 * 16;  // 256 CUs x 16 workgroups, grid-stride beyond that

inline int grid_for(long long work_items, int per_block) {
  return (int)std::max<long long>(1, std::min<long long>(ceil_div_ll(work_items, per_block), kMaxGrid));
}

// ---------------------------------------------------------------------------------
// (a1) preprocess_image: (x - mean) / std on uint8 CHW BGR, zero outside each image's own
// size (ImageList.from_tensors pads AFTER normalisation).  rcnn_wsovod.py:321-328
// ---------------------------------------------------------------------------------
__global__ void preprocess_nchw_kernel(const uint8_t* __restrict__ img, const int* __restrict__ sizes,
                                       float m0, float m1, float m2, float s0, float s1, float s2, int N, int Hp,
                                       int Wp, float* __restrict__ out) {
  const long long total = (long long)N * 3 * Hp * Wp;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int w = (int)(i % Wp);
    const int h = (int)((i / Wp) % Hp);
    const int c = (int)((i / ((long long)Wp * Hp)) % 3);
    const int n = (int)(i / ((long long)Wp * Hp * 3));
    const bool inside = h < sizes[2 * n] && w < sizes[2 * n + 1];
    const float mean = c == 0 ? m0 : (c == 1 ? m1 : m2);
    const float sd = c == 0 ? s0 : (c == 1 ? s1 : s2);
    out[i] = inside ? ((float)img[i] - mean) / sd : 0.f;
  }
}

// Stem conv1 (3x3, stride 2, pad 1, Cin=3; resnet_wsl.py:375-383) as a GEMM operand: row m =
// output pixel, 32 columns: k = (r*3+q)*3 + c for k < 27 (matching the [Cout][kh][kw][Cin]
// weight order), zeros above.  Normalisation is fused; padding is zero AFTER normalisation.
template <typename T>
__global__ void stem_im2col_kernel(const uint8_t* __restrict__ img, const int* __restrict__ sizes, float m0,
                                   float m1, float m2, float s0, float s1, float s2, int N, int Hp, int Wp, int Ho,
                                   int Wo, T* __restrict__ out) {
  const long long total = (long long)N * Ho * Wo * 32;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int k = (int)(i & 31);
    const long long m = i >> 5;
    const int wo = (int)(m % Wo);
    const int ho = (int)((m / Wo) % Ho);
    const int n = (int)(m / ((long long)Wo * Ho));
    float v = 0.f;
    if (k < 27) {
      const int c = k % 3, tap = k / 3, r = tap / 3, q = tap % 3;
      const int h = ho * 2 - 1 + r, w = wo * 2 - 1 + q;
      if (h >= 0 && w >= 0 && h < sizes[2 * n] && w < sizes[2 * n + 1]) {
        const float mean = c == 0 ? m0 : (c == 1 ? m1 : m2);
        const float sd = c == 0 ? s0 : (c == 1 ? s1 : s2);
        v = ((float)img[(((long long)n * 3 + c) * Hp + h) * Wp + w] - mean) / sd;
      }
    }
    out[i] = from_f32<T>(v);
  }
}

// ---------------------------------------------------------------------------------
// 2x2 max pool over NHWC (stem pool s2; res2 tail s2; res3 tail ZeroPad2d((0,1,0,1)) + s1).
// resnet_wsl.py:85-92,408.  8 channels (bf16) / 4 channels (fp32) per lane = 16 B.
// ---------------------------------------------------------------------------------
template <typename T>
__global__ void maxpool2x2_nhwc_kernel(const T* __restrict__ in, int N, int H, int W, int C, int Ho, int Wo,
                                       int stride, int zero_pad, T* __restrict__ out) {
  constexpr int V = 16 / sizeof(T);
  const int cv = C / V;
  const long long total = (long long)N * Ho * Wo * cv;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % cv) * V;
    const int wo = (int)((i / cv) % Wo);
    const int ho = (int)((i / ((long long)cv * Wo)) % Ho);
    const int n = (int)(i / ((long long)cv * Wo * Ho));
    float best[V];
    bool any = false;
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
      for (int dx = 0; dx < 2; ++dx) {
        const int h = ho * stride + dy, w = wo * stride + dx;
        float v[V];
        if (h < H && w < W) {
          const uint4 raw = *(const uint4*)(in + (((long long)n * H + h) * W + w) * C + c);
          const T* e = (const T*)&raw;
#pragma unroll
          for (int j = 0; j < V; ++j) v[j] = to_f32(e[j]);
        } else if (zero_pad) {
#pragma unroll
          for (int j = 0; j < V; ++j) v[j] = 0.f;
        } else {
          continue;
        }
#pragma unroll
        for (int j = 0; j < V; ++j) best[j] = any ? fmaxf(best[j], v[j]) : v[j];
        any = true;
      }
    uint4 o;
    T* oe = (T*)&o;
#pragma unroll
    for (int j = 0; j < V; ++j) oe[j] = from_f32<T>(best[j]);
    *(uint4*)(out + (((long long)n * Ho + ho) * Wo + wo) * C + c) = o;
  }
}

// ---------------------------------------------------------------------------------
// 2x2 max pool backward (round 6; trainable res2 / res3 stages, resnet_wsl.py:85-92 under autograd): gather form -- every
// input position adds the output gradients of the windows whose FIRST maximum (scan order (0,0), (0,1), (1,0), (1,1), strict
// '>': torch's max_pool2d index rule; the zero cells of ZeroPad2d((0,1,0,1)) take part and swallow their share) it is.
// X2: `in` is a bf16x2 map (value = hi + lo, as the forward pool compares them).  Gradients fp32.
// ---------------------------------------------------------------------------------
template <typename T, bool X2>
__global__ void maxpool2x2_bwd_kernel(const T* __restrict__ in, int N, int H, int W, int C, int Ho, int Wo, int stride,
                                      int zero_pad, const float* __restrict__ dout, float* __restrict__ din) {
  constexpr int V = 4;
  const int cv = C / V;
  const long long total = (long long)N * H * W * cv;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % cv) * V;
    const int w = (int)((i / cv) % W);
    const int h = (int)((i / ((long long)cv * W)) % H);
    const int n = (int)(i / ((long long)cv * W * H));
    auto cell = [&](int hh, int ww, float (&v)[V]) -> bool {  // false: the cell does not exist (no padding)
      if (hh < H && ww < W) {
        const long long pix = ((long long)n * H + hh) * W + ww;
        if constexpr (X2) {
          const bf16_t* q = (const bf16_t*)in + pix * 2 * C + ((c >> 5) << 6) + (c & 31);
#pragma unroll
          for (int j = 0; j < V; ++j) v[j] = (float)q[j] + (float)q[32 + j];
        } else {
#pragma unroll
          for (int j = 0; j < V; ++j) v[j] = to_f32(in[pix * C + c + j]);
        }
        return true;
      }
#pragma unroll
      for (int j = 0; j < V; ++j) v[j] = 0.f;
      return zero_pad != 0;
    };
    float acc[V] = {0.f, 0.f, 0.f, 0.f};
    const int ho_lo = stride == 2 ? (h >> 1) : max(h - 1, 0), ho_hi = stride == 2 ? (h >> 1) : h;
    const int wo_lo = stride == 2 ? (w >> 1) : max(w - 1, 0), wo_hi = stride == 2 ? (w >> 1) : w;
    for (int ho = ho_lo; ho <= ho_hi; ++ho)
      for (int wo = wo_lo; wo <= wo_hi; ++wo) {
        if (ho >= Ho || wo >= Wo) continue;
        const int me = (h - ho * stride) * 2 + (w - wo * stride);  // this position's index in the window's scan
        float best[V];
        int arg[V] = {-1, -1, -1, -1};
        bool any = false;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          float v[V];
          if (!cell(ho * stride + (k >> 1), wo * stride + (k & 1), v)) continue;
#pragma unroll
          for (int j = 0; j < V; ++j)
            if (!any || v[j] > best[j]) { best[j] = v[j]; arg[j] = k; }
          any = true;
        }
        const float* g = dout + (((long long)n * Ho + ho) * Wo + wo) * C + c;
#pragma unroll
        for (int j = 0; j < V; ++j)
          if (arg[j] == me) acc[j] += g[j];
      }
    float* d = din + (((long long)n * H + h) * W + w) * C + c;
#pragma unroll
    for (int j = 0; j < V; ++j) d[j] = acc[j];
  }
}

// ---------------------------------------------------------------------------------
// transpose + cast: dst[c][r] = src[r][c]  (64x64 LDS tile; 8 elements = 16/32 B per lane on both
// the load and the store side), and plain cast.
// ---------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ void load8(const T* p, bool vec, int nvalid, float (&v)[8]) {
  if (vec) {
    if constexpr (sizeof(T) == 2) {
      const uint4 raw = *(const uint4*)p;
      const T* e = (const T*)&raw;
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = to_f32(e[j]);
    } else {
      const float4 a = *(const float4*)p, b = *(const float4*)(p + 4);
      v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    }
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = j < nvalid ? to_f32(p[j]) : 0.f;
  }
}
template <typename T>
__device__ __forceinline__ void store8(T* p, bool vec, int nvalid, const float (&v)[8]) {
  if (vec) {
    if constexpr (sizeof(T) == 2) {
      uint4 raw;
      T* e = (T*)&raw;
#pragma unroll
      for (int j = 0; j < 8; ++j) e[j] = from_f32<T>(v[j]);
      *(uint4*)p = raw;
    } else {
      *(float4*)p = make_float4(v[0], v[1], v[2], v[3]);
      *(float4*)(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
    }
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (j < nvalid) p[j] = from_f32<T>(v[j]);
  }
}

template <typename TS, typename TD>
__global__ __launch_bounds__(256) void transpose_cast_kernel(const TS* __restrict__ src, long long lds_, int R, int C,
                                                             TD* __restrict__ dst, long long ldd) {
  __shared__ float tile[64][65];
  const int tid = threadIdx.x;
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  const bool src_al = ((lds_ * sizeof(TS)) % 16 == 0) && (((uintptr_t)src & 15) == 0);
  const bool dst_al = ((ldd * sizeof(TD)) % 16 == 0) && (((uintptr_t)dst & 15) == 0);
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int row = (tid >> 3) + 32 * it, cg = (tid & 7) * 8;
    const int r = r0 + row, c = c0 + cg;
    float v[8];
    if (r < R && c < C) load8(src + (long long)r * lds_ + c, src_al && c + 8 <= C, C - c, v);
    else {
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = 0.f;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) tile[row][cg + j] = v[j];
  }
  __syncthreads();
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int col = (tid >> 3) + 32 * it, rg = (tid & 7) * 8;
    const int c = c0 + col, r = r0 + rg;
    if (c < C && r < R) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = tile[rg + j][col];
      store8(dst + (long long)c * ldd + r, dst_al && r + 8 <= R, R - r, v);
    }
  }
}

template <typename TS, typename TD>
__global__ void cast_kernel(const TS* __restrict__ src, TD* __restrict__ dst, long long n) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    dst[i] = from_f32<TD>(to_f32(src[i]));
}

// ---------------------------------------------------------------------------------
// bf16x3 operand split (MODEL.HIP.PRECISION "bf16x3": fp32-grade products on the bf16 MFMA path).
// x = hi + lo + O(2^-17 |x|) with hi = bf16(x), lo = bf16(x - hi).  A contraction sum_k a_k b_k is evaluated as
// sum_k (ah*bh + ah*bl + al*bh) by handing the UNCHANGED bf16 kernels operands three times as long along the
// reduction index:  A side [hi | hi | lo],  B side [hi | lo | hi]  (the al*bl term, 2^-16 relative to a product, is
// dropped; accumulation stays fp32 inside the MFMA chain).  One pass: 4 B read, 6 B written per element.
//   side 0 = A pattern (hi, hi, lo), side 1 = B pattern (hi, lo, hi);
//   block b of row r lands at dst[b * block_stride + r * ld_dst + c]  (block_stride = cols: the blocks sit side by
//   side along K -- NT GEMMs, NHWC conv channels; block_stride = rows_padded * ld_dst: stacked along the rows -- the
//   operands of the transposed-read dW kernel, whose reduction index is the row).
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void split3_bf16_kernel(const float* __restrict__ src, long long ld_src, int rows,
                                                          int cols, bf16_t* __restrict__ dst, long long ld_dst,
                                                          long long block_stride, int side) {
  const int cg = (cols + 7) >> 3;  // 8-element groups per row
  const long long total = (long long)rows * cg;
  const bool al = ((ld_src & 3) == 0) && ((((uintptr_t)src) & 15) == 0) && ((ld_dst & 7) == 0) &&
                  ((block_stride & 7) == 0) && ((((uintptr_t)dst) & 15) == 0);
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int r = (int)(i / cg), c = (int)(i - (long long)r * cg) * 8;
    const float* s = src + (long long)r * ld_src + c;
    bf16_t* d = dst + (long long)r * ld_dst + c;
    float v[8];
    const int n = min(8, cols - c);
    if (al && n == 8) {
      const f32x4 a = *(const f32x4*)s, b = *(const f32x4*)(s + 4);
      v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3];
      v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = j < n ? s[j] : 0.f;
    }
    bf16x8 hi, lo;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      hi[j] = (bf16_t)v[j];
      lo[j] = (bf16_t)(v[j] - (float)hi[j]);
    }
    const bf16x8 b1 = side ? lo : hi, b2 = side ? hi : lo;
    if (al && n == 8) {
      *(bf16x8*)d = hi;
      *(bf16x8*)(d + block_stride) = b1;
      *(bf16x8*)(d + 2 * block_stride) = b2;
    } else {
      for (int j = 0; j < n; ++j) {
        d[j] = hi[j];
        d[block_stride + j] = b1[j];
        d[2 * block_stride + j] = b2[j];
      }
    }
  }
}

// ---------------------------------------------------------------------------------
// Row L2 norm for the cosine-similarity head: row_scale[m] = T / max(||x_m||, eps)
// (F.normalize eps=1e-12, open_vocabulary_classifier.py:91-92).  Wavefront per row.
// ---------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void row_l2norm_kernel(const T* __restrict__ x, long long ld, int M, int D,
                                                         float temp, float eps, float* __restrict__ row_scale) {
  const int lane = threadIdx.x & 63;
  const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (m >= M) return;
  float s = 0.f;
  for (int d = lane; d < D; d += 64) {
    const float v = to_f32(x[(long long)m * ld + d]);
    s += v * v;
  }
  s = wave_reduce_sum(s);
  if (lane == 0) row_scale[m] = temp / fmaxf(sqrtf(s), eps);
}

// backward of zn = T*z/max(||z||,eps) followed by the ReLU that produced z:
//   dz = s*(u - z*(z.u)/||z||^2)   (||z|| > eps)     | dz = s*u (clamped branch);  dz *= (z > 0)
template <typename T>
__global__ __launch_bounds__(256) void row_l2norm_bwd_kernel(const T* __restrict__ z, long long ldz,
                                                             const float* __restrict__ u, long long ldu, int M, int D,
                                                             float temp, float eps, int relu_mask,
                                                             float* __restrict__ dz, long long lddz) {
  const int lane = threadIdx.x & 63;
  const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (m >= M) return;
  float nn = 0.f, zu = 0.f;
  for (int d = lane; d < D; d += 64) {
    const float zv = to_f32(z[(long long)m * ldz + d]);
    nn += zv * zv;
    zu += zv * u[(long long)m * ldu + d];
  }
  nn = wave_reduce_sum(nn);
  zu = wave_reduce_sum(zu);
  const float nrm = sqrtf(nn);
  const bool clamped = nrm <= eps;
  const float s = temp / fmaxf(nrm, eps);
  const float coef = clamped ? 0.f : zu / nn;
  for (int d = lane; d < D; d += 64) {
    const float zv = to_f32(z[(long long)m * ldz + d]);
    float g = s * (u[(long long)m * ldu + d] - zv * coef);
    if (relu_mask && !(zv > 0.f)) g = 0.f;
    dz[(long long)m * lddz + d] = g;
  }
}

// ---------------------------------------------------------------------------------
// Segmented column sum: out[g][n] (+)= scale * sum_{m in segment g} x[m][n]  (bias grads with one segment; per-image
// data-aware-feature grads; global average pool with uniform segments).  Deterministic AND unquantised: a fixed-order
// two-stage reduction (round 3; it replaces the 64-bit fixed-point atomics of round 2, whose 2^-30 absolute quantum was
// coarser than fp32 for small gradient partials, flushed |partials| < 4.7e-10 to zero and wrapped past 8.6e9).
//   stage 1: grid = (column groups of 64 lanes x V columns) x (chunks of 128 rows).  A workgroup walks the segments
//            that intersect its chunk in order; per piece its 4 wavefronts stride the rows (lane = V columns, one 16-byte
//            load per row), combine through LDS in a fixed order and store ONE fp32 partial per column into the chunk's
//            slot: `first` (the first segment that intersects the chunk), `last` (the last one, if different), or -- for a
//            segment that lies wholly inside the chunk -- the segment's own `direct` row.  No atomics.
//   stage 2: thread per (segment, column) adds the slots of the chunks the segment spans in chunk order (fp64
//            accumulator), applies the scale and stores / accumulates.
// Same input -> same bits, whatever the launch timing; NaN / Inf propagate as in any fp32 sum.
// Workspace (caller-owned, fp32): [2 * chunks + G][N]  (first | last | direct).
// ---------------------------------------------------------------------------------
constexpr int CS_RC = 128;  // rows per chunk

struct SegView {
  const int* seg;
  int G, uniform_rows, M;
  __device__ __forceinline__ int at(int i) const { return seg ? seg[i] : min(i * uniform_rows, M); }
  // segment that contains row x (seg[0] <= x < seg[G]): skips empty segments
  __device__ __forceinline__ int containing(int x) const {
    if (!seg) return x / uniform_rows;
    int lo = 0, hi = G;  // invariant: seg[lo] <= x < seg[hi]
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (seg[mid] <= x) lo = mid; else hi = mid;
    }
    return lo;
  }
  // first / last segment intersecting rows [r0, r1); false if none
  __device__ __forceinline__ bool span(int r0, int r1, int& gf, int& gl) const {
    const int a = max(r0, at(0)), b = min(r1, at(G));
    if (a >= b) return false;
    gf = containing(a);
    gl = containing(b - 1);
    return true;
  }
};

template <typename T>
__global__ __launch_bounds__(256) void segment_colsum_kernel(const T* __restrict__ x, long long ld, const SegView sv, int N,
                                                             int chunks, float* __restrict__ ws) {
  constexpr int V = 16 / sizeof(T);  // columns per lane: one 16-B load per row
  __shared__ float part[3][64][V + 1];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n0 = (blockIdx.x * 64 + lane) * V;
  const int c = blockIdx.y;
  const int r0 = c * CS_RC, r1 = min(r0 + CS_RC, sv.M);
  const bool col_ok = n0 < N;
  const bool vec = (n0 + V <= N) && ((ld * sizeof(T)) % 16 == 0) && (((uintptr_t)x & 15) == 0);
  int gf, gl;
  if (!sv.span(r0, r1, gf, gl)) return;
  for (int g = gf; g <= gl; ++g) {
    const int a = max(r0, sv.at(g)), b = min(r1, sv.at(g + 1));
    if (a >= b) continue;  // empty segment
    float acc[V];
#pragma unroll
    for (int j = 0; j < V; ++j) acc[j] = 0.f;
    if (col_ok)
      for (int m = a + wave; m < b; m += 4) {
        const T* row = x + (long long)m * ld + n0;
        if (vec) {
          typedef unsigned int u4 __attribute__((ext_vector_type(4)));
          const u4 raw = __builtin_nontemporal_load((const u4*)row);  // read once: streaming (no L2 allocation)
          const T* e = (const T*)&raw;
#pragma unroll
          for (int j = 0; j < V; ++j) acc[j] += to_f32(e[j]);
        } else {
#pragma unroll
          for (int j = 0; j < V; ++j)
            if (n0 + j < N) acc[j] += to_f32(row[j]);
        }
      }
    if (wave > 0) {
#pragma unroll
      for (int j = 0; j < V; ++j) part[wave - 1][lane][j] = acc[j];
    }
    __syncthreads();
    if (wave == 0 && col_ok) {
      float* dst = g == gf ? ws + (long long)c * N : g == gl ? ws + (long long)(chunks + c) * N
                                                             : ws + (long long)(2 * chunks + g) * N;
#pragma unroll
      for (int j = 0; j < V; ++j)
        if (n0 + j < N) dst[n0 + j] = (acc[j] + part[0][lane][j]) + (part[1][lane][j] + part[2][lane][j]);
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void colsum_finalize_kernel(const float* __restrict__ ws, const SegView sv, int N,
                                                              int chunks, float scale, float* __restrict__ out,
                                                              long long ldo, int accumulate) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long long)sv.G * N) return;
  const int g = (int)(i / N), n = (int)(i - (long long)g * N);
  const int a = sv.at(g), b = sv.at(g + 1);
  double sum = 0.0;
  if (a < b) {
    for (int c = a / CS_RC; c <= (b - 1) / CS_RC; ++c) {
      int gf, gl;
      sv.span(c * CS_RC, min((c + 1) * CS_RC, sv.M), gf, gl);  // intersects segment g, so it has a span
      const float* src = g == gf ? ws + (long long)c * N : g == gl ? ws + (long long)(chunks + c) * N
                                                                   : ws + (long long)(2 * chunks + g) * N;
      sum += (double)src[n];
    }
  }
  const float v = (float)(sum * (double)scale);
  float* o = out + (long long)g * ldo + n;
  *o = accumulate ? *o + v : v;
}

// x *= num[0] / den[0]  (device scalars: upstream loss grad / normaliser; no host sync)
__global__ void scale_by_device_scalar_kernel(float* __restrict__ x, long long n, const float* __restrict__ num,
                                              const float* __restrict__ den) {
  const float f = (num ? num[0] : 1.f) / (den ? den[0] : 1.f);
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    x[i] *= f;
}

// ---------------------------------------------------------------------------------
// Fused SGD with momentum + weight decay (torch.optim.SGD semantics, dampening 0, as built by
// wsovod/engine/defaults.py:274-318):  g += wd*p; buf = mu*buf + g; p -= lr*buf.
// Optionally refreshes the bf16 shadow of p in the same pass.  float4 per lane.
// ---------------------------------------------------------------------------------
__global__ void sgd_momentum_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ buf,
                                    long long n, float lr, float mu, float wd, float gscale,
                                    bf16_t* __restrict__ shadow) {
  const long long n4 = n >> 2;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4;
       i += (long long)gridDim.x * blockDim.x) {
    // every byte is touched once per step: non-temporal accesses keep the 2.7 GB stream out of the caches
    f32x4 pv = __builtin_nontemporal_load((const f32x4*)p + i);
    const f32x4 gv = __builtin_nontemporal_load((const f32x4*)g + i);
    f32x4 bv = __builtin_nontemporal_load((const f32x4*)buf + i);
    bv = mu * bv + (gv * gscale + wd * pv);
    pv -= lr * bv;
    __builtin_nontemporal_store(bv, (f32x4*)buf + i);
    __builtin_nontemporal_store(pv, (f32x4*)p + i);
    if (shadow) {
      bf16x4 s = {(bf16_t)pv[0], (bf16_t)pv[1], (bf16_t)pv[2], (bf16_t)pv[3]};
      ((bf16x4*)shadow)[i] = s;
    }
  }
  // tail (n not a multiple of 4)
  const long long t0 = n4 << 2;
  for (long long i = t0 + (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x) {
    float b = mu * buf[i] + (g[i] * gscale + wd * p[i]);
    buf[i] = b;
    const float pv = p[i] - lr * b;
    p[i] = pv;
    if (shadow) shadow[i] = (bf16_t)pv;
  }
}

// Multi-tensor form: one launch updates up to kSgdMax tensors; block b works on a 4096-element chunk of the tensor
// whose block range contains b (the table travels as a kernel argument).
constexpr int kSgdMax = 32, kSgdChunk = 4096;
struct SgdTable {
  float* p[kSgdMax];
  const float* g[kSgdMax];
  float* buf[kSgdMax];
  bf16_t* shadow[kSgdMax];
  long long n[kSgdMax];
  float lr[kSgdMax], wd[kSgdMax];
  int first_block[kSgdMax + 1];
  int count;
  unsigned g_bf16;  // bit k: g[k] points at bf16 values (gradients that travelled in the bf16 wire format)
  unsigned x2_shadow;  // bit k: shadow[k] is a bf16x2 copy of the parameter (include/wsovod_hip.h), n[k] a multiple of 32
  unsigned mx_shadow;  // bit k: shadow[k] is an f16mx copy with the per-tensor E8M0 byte *mx_scale[k] (round 6)
  const unsigned char* mx_scale[kSgdMax];
  const float* used[kSgdMax];  // optional device flag: 0 = no rank produced a gradient for the tensor -> left untouched
  const float* coef[kSgdMax];  // optional device scalar multiplied into the gradient scale (norm clipping coefficient)
  const float* lr_dev[kSgdMax];  // optional device scalar that replaces lr[k] (a captured step graph under an LR schedule)
  float clip[kSgdMax];         // > 0: the scaled gradient is clamped to [-clip, clip] (clip_grad_value_)
};

__global__ __launch_bounds__(256) void sgd_momentum_multi_kernel(const SgdTable t, float mu, float gscale0) {
  int k = 0;
  while (k + 1 < t.count && (int)blockIdx.x >= t.first_block[k + 1]) ++k;
  const long long base = (long long)((int)blockIdx.x - t.first_block[k]) * kSgdChunk;
  float* __restrict__ p = t.p[k];
  const float* __restrict__ g = t.g[k];
  float* __restrict__ buf = t.buf[k];
  bf16_t* __restrict__ shadow = t.shadow[k];
  const long long n = t.n[k];
  const float lr = t.lr_dev[k] ? *t.lr_dev[k] : t.lr[k], wd = t.wd[k];
  const bool gb = (t.g_bf16 >> k) & 1u;
  const bool sx2 = (t.x2_shadow >> k) & 1u;
  const bool smx = (t.mx_shadow >> k) & 1u;
  const int mx_exp = smx ? (int)*t.mx_scale[k] - 127 : 0;
  const float mx_iq = __builtin_ldexpf(1.0f, -mx_exp), mx_il = __builtin_ldexpf(1.0f, -(mx_exp - 11));
  const bf16_t* __restrict__ g16 = (const bf16_t*)t.g[k];
  if (t.used[k] && *t.used[k] == 0.f) return;  // torch.optim.SGD skips parameters whose grad is None
  const float gscale = t.coef[k] ? gscale0 * *t.coef[k] : gscale0;
  const float cv = t.clip[k];
  const bool vec = ((((uintptr_t)p | (uintptr_t)buf) & 15) == 0) && (((uintptr_t)shadow & 7) == 0) &&
                   (((uintptr_t)g & (gb ? 7 : 15)) == 0);
  for (int e = threadIdx.x * 4; e < kSgdChunk; e += 256 * 4) {
    const long long i = base + e;
    if (i >= n) break;
    if (vec && i + 3 < n) {
      f32x4 pv = __builtin_nontemporal_load((const f32x4*)(p + i));
      f32x4 gv;
      if (gb) {
        const bf16x4 h = __builtin_nontemporal_load((const bf16x4*)(g16 + i));
        gv = f32x4{(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
      } else {
        gv = __builtin_nontemporal_load((const f32x4*)(g + i));
      }
      f32x4 bv = __builtin_nontemporal_load((const f32x4*)(buf + i));
      gv = gv * gscale;
      if (cv > 0.f) {
#pragma unroll
        for (int e4 = 0; e4 < 4; ++e4) gv[e4] = fminf(fmaxf(gv[e4], -cv), cv);
      }
      bv = mu * bv + (gv + wd * pv);
      pv -= lr * bv;
      __builtin_nontemporal_store(bv, (f32x4*)(buf + i));
      __builtin_nontemporal_store(pv, (f32x4*)(p + i));
      if (shadow && smx) {  // f16mx: 8 B of fp16 hi, 4 B of e4m3 q, 4 B of e4m3 ql in the value's 128-byte group
        wsovod_mx::f16x4 h4;
        int q4, l4;
        wsovod_mx::mx_enc4(pv, mx_iq, mx_il, h4, q4, l4);
        char* grp = (char*)shadow + ((i >> 5) << 7);
        const int w = (int)(i & 31);
        *(wsovod_mx::f16x4*)(grp + 2 * w) = h4;
        *(int*)(grp + 64 + w) = q4;
        *(int*)(grp + 96 + w) = l4;
      } else if (shadow) {
        const bf16x4 hi = bf16x4{(bf16_t)pv[0], (bf16_t)pv[1], (bf16_t)pv[2], (bf16_t)pv[3]};
        if (sx2) {  // hi at slot 64 (i / 32) + i % 32, lo 32 slots further (4 consecutive values never straddle a group)
          bf16_t* q = shadow + ((i >> 5) << 6) + (i & 31);
          *(bf16x4*)q = hi;
          *(bf16x4*)(q + 32) = bf16x4{(bf16_t)(pv[0] - (float)hi[0]), (bf16_t)(pv[1] - (float)hi[1]),
                                      (bf16_t)(pv[2] - (float)hi[2]), (bf16_t)(pv[3] - (float)hi[3])};
        } else {
          *(bf16x4*)(shadow + i) = hi;
        }
      }
    } else {
      for (long long q = i; q < min(i + 4, n); ++q) {
        float gq = (gb ? (float)g16[q] : g[q]) * gscale;
        if (cv > 0.f) gq = fminf(fmaxf(gq, -cv), cv);
        const float b = mu * buf[q] + (gq + wd * p[q]);
        buf[q] = b;
        const float pv = p[q] - lr * b;
        p[q] = pv;
        if (shadow && sx2) {
          bf16_t* d = shadow + ((q >> 5) << 6) + (q & 31);
          d[0] = (bf16_t)pv;
          d[32] = (bf16_t)(pv - (float)d[0]);
        } else if (shadow) {
          shadow[q] = (bf16_t)pv;
        }
      }
    }
  }
}

// Gradient-norm clipping (torch.nn.utils.clip_grad_norm_, L2) without a host read: a fixed-order two-stage sum of
// squares -- one fp32 partial per 4096-element chunk (lanes stride the chunk, a tree over the workgroup), then per tensor
// the chunk partials added in chunk order in fp64 -- and one tiny pass that turns the sums into the coefficients
// min(1, max_norm / (norm + 1e-6)) the SGD kernel multiplies into its gradient scale.
struct SumsqTable {
  const void* g[kSgdMax];
  long long n[kSgdMax];
  int first_block[kSgdMax + 1];
  int count;
  unsigned g_bf16;
  const float* used[kSgdMax];
};

__global__ __launch_bounds__(256) void grad_sumsq_partial_kernel(const SumsqTable t, float* __restrict__ partials) {
  __shared__ float red[256];
  int k = 0;
  while (k + 1 < t.count && (int)blockIdx.x >= t.first_block[k + 1]) ++k;
  const long long base = (long long)((int)blockIdx.x - t.first_block[k]) * kSgdChunk;
  const long long n = t.n[k];
  const bool gb = (t.g_bf16 >> k) & 1u;
  const float* g = (const float*)t.g[k];
  const bf16_t* g16 = (const bf16_t*)t.g[k];
  float acc = 0.f;
  for (int e = threadIdx.x; e < kSgdChunk; e += 256) {
    const long long i = base + e;
    if (i < n) {
      const float v = gb ? (float)g16[i] : g[i];
      acc += v * v;
    }
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) partials[blockIdx.x] = red[0];
}

// one wavefront per tensor: its chunk partials in chunk order (lane-strided, then a fixed shuffle tree), fp64
__global__ __launch_bounds__(64) void grad_sumsq_tensor_kernel(const SumsqTable t, const float* __restrict__ partials,
                                                                float* __restrict__ sumsq) {
  const int k = blockIdx.x;
  double acc = 0.0;
  for (int b = t.first_block[k] + (int)threadIdx.x; b < t.first_block[k + 1]; b += 64) acc += (double)partials[b];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if (threadIdx.x == 0) sumsq[k] = (t.used[k] && *t.used[k] == 0.f) ? 0.f : (float)acc;
}

__global__ __launch_bounds__(64) void grad_clip_coef_kernel(const float* __restrict__ sumsq, int count, float grad_scale,
                                                             float max_norm, int per_tensor, float* __restrict__ coef) {
  if (per_tensor) {
    for (int k = threadIdx.x; k < count; k += 64) {
      const float norm = grad_scale * sqrtf(sumsq[k]);
      coef[k] = fminf(max_norm / (norm + 1e-6f), 1.0f);
    }
    return;
  }
  double acc = 0.0;
  for (int k = threadIdx.x; k < count; k += 64) acc += (double)sumsq[k];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  const float norm = grad_scale * (float)sqrt(acc);
  const float c = fminf(max_norm / (norm + 1e-6f), 1.0f);
  for (int k = threadIdx.x; k < count; k += 64) coef[k] = c;
}

// Gradient wire format: every trainable tensor's fp32 gradient rounded to bf16 into its slice of ONE flat buffer (the
// operand of a single RCCL all-reduce); same block->tensor table as the SGD kernel.
struct PackTable {
  const float* src[kSgdMax];
  bf16_t* dst[kSgdMax];
  long long n[kSgdMax];
  int first_block[kSgdMax + 1];
  int count;
};

__global__ __launch_bounds__(256) void pack_bf16_multi_kernel(const PackTable t) {
  int k = 0;
  while (k + 1 < t.count && (int)blockIdx.x >= t.first_block[k + 1]) ++k;
  const long long base = (long long)((int)blockIdx.x - t.first_block[k]) * kSgdChunk;
  const float* __restrict__ src = t.src[k];
  bf16_t* __restrict__ dst = t.dst[k];
  const long long n = t.n[k];
  const bool vec = (((uintptr_t)src & 15) == 0) && (((uintptr_t)dst & 7) == 0);
  for (int e = threadIdx.x * 4; e < kSgdChunk; e += 256 * 4) {
    const long long i = base + e;
    if (i >= n) break;
    if (vec && i + 3 < n) {
      const f32x4 v = __builtin_nontemporal_load((const f32x4*)(src + i));
      *(bf16x4*)(dst + i) = bf16x4{(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
    } else {
      for (long long q = i; q < min(i + 4, n); ++q) dst[q] = (bf16_t)src[q];
    }
  }
}

// ---------------------------------------------------------------------------------
// Direct gradient exchange, middle step: after the all-to-all every rank holds `n_shards` bf16 copies of ITS shard of
// the wire buffer (one per rank).  Their sum is taken in fp32 -- exact: 8 bf16 values add without rounding error that
// bf16 could see -- and rounded to bf16 once; the all-gather then distributes the reduced shards.  HBM-bound:
// (n_shards + 1) * 2 bytes per element.
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sum_shards_bf16_kernel(const bf16_t* __restrict__ src, int n_shards,
                                                              long long shard, bf16_t* __restrict__ dst) {
  const long long stride = (long long)gridDim.x * 256 * 8;
  for (long long e = ((long long)blockIdx.x * 256 + threadIdx.x) * 8; e < shard; e += stride) {
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int j = 0; j < n_shards; ++j) {
      const bf16x8 v = __builtin_nontemporal_load((const bf16x8*)(src + j * shard + e));
#pragma unroll
      for (int q = 0; q < 8; ++q) acc[q] += (float)v[q];
    }
    bf16x8 o;
#pragma unroll
    for (int q = 0; q < 8; ++q) o[q] = (bf16_t)acc[q];
    *(bf16x8*)(dst + e) = o;
  }
}

// ---------------------------------------------------------------------------------
// Backward prologue of Linear+ReLU(+Dropout): dA = dy * [y > 0] * scale, written both as
// [M][N] (operand of the dX contraction) and transposed [N][ldt] (operand of the dW
// contraction, reduction dim = proposals).  y == NULL means no mask.
// ---------------------------------------------------------------------------------
// Y_X2: y is a bf16x2 matrix (ldy in values); its hi halves decide the mask (hi = bf16(y) has y's sign and is zero only
// where y is).
template <typename TI, typename TO, typename TY = TI, bool Y_X2 = false>
__global__ __launch_bounds__(256) void mask_transpose_kernel(const TI* __restrict__ dy, long long lddy,
                                                             const TY* __restrict__ y, long long ldy, int M, int N,
                                                             float scale, TO* __restrict__ dA, long long ldda,
                                                             TO* __restrict__ dAt, long long ldt,
                                                             float* __restrict__ colsum) {
  __shared__ float tile[64][65];
  const int tid = threadIdx.x;
  const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
  const bool dy_al = ((lddy * sizeof(TI)) % 16 == 0) && (((uintptr_t)dy & 15) == 0);
  const bool y_al = y && (Y_X2 || (ldy * sizeof(TY)) % 16 == 0) && (((uintptr_t)y & 15) == 0);
  const bool da_al = dA && ((ldda * sizeof(TO)) % 16 == 0) && (((uintptr_t)dA & 15) == 0);
  const bool dt_al = dAt && ((ldt * sizeof(TO)) % 16 == 0) && (((uintptr_t)dAt & 15) == 0);
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int row = (tid >> 3) + 32 * it, cg = (tid & 7) * 8;
    const int m = m0 + row, n = n0 + cg;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = 0.f;
    if (m < M && n < N) {
      const bool full = n + 8 <= N;
      load8(dy + (long long)m * lddy + n, dy_al && full, N - n, v);
      if (y) {
        float yv[8];
        if constexpr (Y_X2) load8(y + 2 * (long long)m * ldy + ((n >> 5) << 6) + (n & 31), y_al && full, N - n, yv);
        else load8(y + (long long)m * ldy + n, y_al && full, N - n, yv);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = yv[j] > 0.f ? v[j] * scale : 0.f;
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] *= scale;
      }
      if (dA) store8(dA + (long long)m * ldda + n, da_al && full, N - n, v);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) tile[row][cg + j] = v[j];
  }
  __syncthreads();
  if (colsum) {  // bias gradient: column sums of the masked tile (rows past M hold zeros), one atomic per column and tile
    __shared__ float part[4][64];
    const int col = tid & 63, q = tid >> 6;
    float sacc = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) sacc += tile[q * 16 + r][col];
    part[q][col] = sacc;
    __syncthreads();
    if (q == 0 && n0 + col < N) {
      const float t = (part[0][col] + part[1][col]) + (part[2][col] + part[3][col]);
      if (t != 0.f) atomicAdd(colsum + n0 + col, t);
    }
  }
  if (dAt) {
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int col = (tid >> 3) + 32 * it, rg = (tid & 7) * 8;
      const int n = n0 + col, m = m0 + rg;
      if (n < N && m < M) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = tile[rg + j][col];
        store8(dAt + (long long)n * ldt + m, dt_al && m + 8 <= M, M - m, v);
      }
    }
  }
}

// out[m][n] = x[m][n] + add[row_group[m]][n]   (box_features += data_aware_features,
// roi_heads.py:762-763; the per-proposal repeat of data_aware_features_head.py:117-121 is
// never materialised)
template <typename T>
__global__ __launch_bounds__(256) void add_group_rows_kernel(const T* __restrict__ x, long long ldx,
                                                             const int* __restrict__ row_group,
                                                             const float* __restrict__ add, long long lda, int M, int N,
                                                             T* __restrict__ out, long long ldo) {
  // a workgroup owns RB consecutive rows; a lane owns 8 columns (16 B of bf16) of each: RB independent 16-byte loads
  // in flight per lane, no per-element index division
  constexpr int RB = 8;
  const bool x_al = ((ldx * sizeof(T)) % 16 == 0) && (((uintptr_t)x & 15) == 0);
  const bool o_al = ((ldo * sizeof(T)) % 16 == 0) && (((uintptr_t)out & 15) == 0);
  const bool a_al = ((lda * sizeof(float)) % 16 == 0) && (((uintptr_t)add & 15) == 0);
  for (int m0 = blockIdx.x * RB; m0 < M; m0 += gridDim.x * RB) {
    for (int n = threadIdx.x * 8; n < N; n += 256 * 8) {
      const bool full = n + 8 <= N;
      float v[RB][8];
#pragma unroll
      for (int r = 0; r < RB; ++r)
        if (m0 + r < M) load8(x + (long long)(m0 + r) * ldx + n, x_al && full, N - n, v[r]);
#pragma unroll
      for (int r = 0; r < RB; ++r) {
        if (m0 + r >= M) continue;
        float a[8];
        load8(add + (long long)row_group[m0 + r] * lda + n, a_al && full, N - n, a);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[r][j] += a[j];
        store8(out + (long long)(m0 + r) * ldo + n, o_al && full, N - n, v[r]);
      }
    }
  }
}

// out[r][c] = x[r][c] * row_scale[r]  (L2-normalised class text embeddings,
// open_vocabulary_classifier.py:59-60,87-89); rows >= R of out are left untouched.
template <typename TO>
__global__ void scale_rows_kernel(const float* __restrict__ x, long long ldx, const float* __restrict__ row_scale,
                                  int R, int Cc, TO* __restrict__ out, long long ldo) {
  const long long total = (long long)R * Cc;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int r = (int)(i / Cc), c = (int)(i - (long long)r * Cc);
    out[(long long)r * ldo + c] = from_f32<TO>(x[(long long)r * ldx + c] * row_scale[r]);
  }
}

}  // namespace

extern "C" {

static int mask_transpose_impl(const void* dy, long long lddy, const void* y, long long ldy, int in_dtype, int M, int N,
                               float scale, void* dA, long long ldda, void* dAt, long long ldt, int out_dtype,
                               float* colsum, wsovod_stream_t stream, int y_dtype = -1) {
  if (M == 0 || N == 0) return WSOVOD_OK;
  if (y_dtype < 0) y_dtype = in_dtype;
  WS_CHECK_ARG(dy && (dA || dAt || colsum), "wsovod_mask_transpose: null pointer");
  static int slot = wsovod::prof_slot("mask_transpose");
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid(ceil_div(N, 64), ceil_div(M, 64));
  wsovod::ProfScope prof(slot, s, 0.0, (double)M * N * 4.0 * 3);
#define MT(TI, TO) hipLaunchKernelGGL((mask_transpose_kernel<TI, TO>), grid, dim3(256), 0, s, (const TI*)dy, lddy, (const TI*)y, ldy, M, N, scale, (TO*)dA, ldda, (TO*)dAt, ldt, colsum)
  if (y && y_dtype == WSOVOD_BF16X2) {
    WS_CHECK_ARG(in_dtype == WSOVOD_F32 && N % 8 == 0 && ldy % 4 == 0, "wsovod_mask_transpose: a bf16x2 y needs an fp32 dy and N a multiple of 8");
#define MTX(TO) hipLaunchKernelGGL((mask_transpose_kernel<float, TO, bf16_t, true>), grid, dim3(256), 0, s, (const float*)dy, lddy, (const bf16_t*)y, ldy, M, N, scale, (TO*)dA, ldda, (TO*)dAt, ldt, colsum)
    if (out_dtype == WSOVOD_F32) MTX(float);
    else if (out_dtype == WSOVOD_BF16) MTX(bf16_t);
    else { wsovod::set_error("wsovod_mask_transpose: bad dtype"); return WSOVOD_ERR_INVALID_ARGUMENT; }
#undef MTX
  } else if (y && y_dtype == WSOVOD_BF16 && in_dtype == WSOVOD_F32) {  // (round 6: the plain bf16 rounding of an f16mx output)
#define MTB(TO) hipLaunchKernelGGL((mask_transpose_kernel<float, TO, bf16_t, false>), grid, dim3(256), 0, s, (const float*)dy, lddy, (const bf16_t*)y, ldy, M, N, scale, (TO*)dA, ldda, (TO*)dAt, ldt, colsum)
    if (out_dtype == WSOVOD_F32) MTB(float);
    else if (out_dtype == WSOVOD_BF16) MTB(bf16_t);
    else { wsovod::set_error("wsovod_mask_transpose: bad dtype"); return WSOVOD_ERR_INVALID_ARGUMENT; }
#undef MTB
  } else if (y && y_dtype != in_dtype) {
    wsovod::set_error("wsovod_mask_transpose: y must have dy's dtype, be bf16 next to an fp32 dy, or be bf16x2");
    return WSOVOD_ERR_INVALID_ARGUMENT;
  } else if (in_dtype == WSOVOD_F32 && out_dtype == WSOVOD_F32) MT(float, float);
  else if (in_dtype == WSOVOD_F32 && out_dtype == WSOVOD_BF16) MT(float, bf16_t);
  else if (in_dtype == WSOVOD_BF16 && out_dtype == WSOVOD_BF16) MT(bf16_t, bf16_t);
  else if (in_dtype == WSOVOD_BF16 && out_dtype == WSOVOD_F32) MT(bf16_t, float);
  else { wsovod::set_error("wsovod_mask_transpose: bad dtype"); return WSOVOD_ERR_INVALID_ARGUMENT; }
#undef MT
  WS_CHECK_LAUNCH("wsovod_mask_transpose");
  return WSOVOD_OK;
}

int wsovod_mask_transpose(const void* dy, long long lddy, const void* y, long long ldy, int in_dtype, int M, int N,
                          float scale, void* dA, long long ldda, void* dAt, long long ldt, int out_dtype,
                          wsovod_stream_t stream) {
  return mask_transpose_impl(dy, lddy, y, ldy, in_dtype, M, N, scale, dA, ldda, dAt, ldt, out_dtype, nullptr, stream);
}

int wsovod_mask_transpose_colsum(const void* dy, long long lddy, const void* y, long long ldy, int in_dtype, int M, int N,
                                 float scale, void* dA, long long ldda, void* dAt, long long ldt, int out_dtype,
                                 float* colsum, wsovod_stream_t stream) {
  WS_CHECK_ARG(colsum, "wsovod_mask_transpose_colsum: null colsum");
  return mask_transpose_impl(dy, lddy, y, ldy, in_dtype, M, N, scale, dA, ldda, dAt, ldt, out_dtype, colsum, stream);
}

int wsovod_mask_transpose_ex(const void* dy, long long lddy, int dy_dtype, const void* y, long long ldy, int y_dtype, int M,
                             int N, float scale, void* dA, long long ldda, void* dAt, long long ldt, int out_dtype,
                             float* colsum, wsovod_stream_t stream) {
  return mask_transpose_impl(dy, lddy, y, ldy, dy_dtype, M, N, scale, dA, ldda, dAt, ldt, out_dtype, colsum, stream, y_dtype);
}

int wsovod_add_group_rows(const void* x, long long ldx, int dtype, const int* row_group, const float* add,
                          long long ld_add, int M, int N, void* out, long long ldo, wsovod_stream_t stream) {
  if (M == 0 || N == 0) return WSOVOD_OK;
  WS_CHECK_ARG(x && row_group && add && out, "wsovod_add_group_rows: null pointer");
  static int slot = wsovod::prof_slot("add_group_rows");
  hipStream_t s = (hipStream_t)stream;
  wsovod::ProfScope prof(slot, s, 0.0, (double)M * N * (dtype == WSOVOD_BF16 ? 4.0 : 8.0));
  const int grid = std::min(ceil_div(M, 8), 1 << 20);  // 8 rows per block step
  if (dtype == WSOVOD_BF16X2) {
    WS_CHECK_ARG(N % 32 == 0 && ldx % 4 == 0 && ldo % 4 == 0 && ld_add % 4 == 0 &&
                     (((uintptr_t)x | (uintptr_t)out | (uintptr_t)add) & 15) == 0,
                 "wsovod_add_group_rows: bf16x2 rows need N a multiple of 32 and 16-byte aligned rows");
    wsovod::x2_add_group_rows(x, ldx, row_group, add, ld_add, M, N, out, ldo, s);
  } else if (dtype == WSOVOD_BF16)
    hipLaunchKernelGGL(add_group_rows_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, (const bf16_t*)x, ldx, row_group, add, ld_add, M, N, (bf16_t*)out, ldo);
  else if (dtype == WSOVOD_F32)
    hipLaunchKernelGGL(add_group_rows_kernel<float>, dim3(grid), dim3(256), 0, s, (const float*)x, ldx, row_group, add, ld_add, M, N, (float*)out, ldo);
  else { wsovod::set_error("wsovod_add_group_rows: bad dtype"); return WSOVOD_ERR_INVALID_ARGUMENT; }
  WS_CHECK_LAUNCH("wsovod_add_group_rows");
  return WSOVOD_OK;
}

int wsovod_scale_rows(const float* x, long long ldx, const float* row_scale, int R, int C, void* out, long long ldo,
                      int out_dtype, wsovod_stream_t stream) {
  if (R == 0 || C == 0) return WSOVOD_OK;
  WS_CHECK_ARG(x && row_scale && out, "wsovod_scale_rows: null pointer");
  static int slot = wsovod::prof_slot("scale_rows");
  hipStream_t s = (hipStream_t)stream;
  wsovod::ProfScope prof(slot, s, 0.0, (double)R * C * 8.0);
  const int grid = grid_for((long long)R * C, 256);
  if (out_dtype == WSOVOD_BF16)
    hipLaunchKernelGGL(scale_rows_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, x, ldx, row_scale, R, C, (bf16_t*)out, ldo);
  else if (out_dtype == WSOVOD_F32)
    hipLaunchKernelGGL(scale_rows_kernel<float>, dim3(grid), dim3(256), 0, s, x, ldx, row_scale, R, C, (float*)out, ldo);
  else { wsovod::set_error("wsovod_scale_rows: bad dtype"); return WSOVOD_ERR_INVALID_ARGUMENT; }
  WS_CHECK_LAUNCH("wsovod_scale_rows");
  return WSOVOD_OK;
}

int wsovod_preprocess_image(const unsigned char* img, const int* sizes, const float* mean_host,
                            const float* std_host, int N, int Hp, int Wp, float* out, wsovod_stream_t stream) {
  WS_CHECK_ARG(N >= 0 && Hp > 0 && Wp > 0, "wsovod_preprocess_image: bad shape");
  if (N == 0) return WSOVOD_OK;
  WS_CHECK_ARG(img && sizes && mean_host && std_host && out, "wsovod_preprocess_image: null pointer");
  static int slot = wsovod::prof_slot("preprocess_nchw");
  hipStream_t s = (hipStream_t)stream;
  const long long total = (long long)N * 3 * Hp * Wp;
  wsovod::ProfScope prof(slot, s, 0.0, total * 5.0);
  hipLaunchKernelGGL(preprocess_nchw_kernel, dim3(grid_for(total, 256)), dim3(256), 0, s, img, sizes, mean_host[0],
                     mean_host[1], mean_host[2], std_host[0], std_host[1], std_host[2], N, Hp, Wp, out);
  WS_CHECK_LAUNCH("wsovod_preprocess_image");
  return WSOVOD_OK;
}

int wsovod_stem_im2col(const unsigned char* img, const int* sizes, const float* mean_host, const float* std_host,
                       int N, int Hp, int Wp, void* out, int out_dtype, wsovod_stream_t stream) {
  WS_CHECK_ARG(N >= 0 && Hp > 0 && Wp > 0, "wsovod_stem_im2col: bad shape");
  if (N == 0) return WSOVOD_OK;
  WS_CHECK_ARG(img && sizes && mean_host && std_host && out, "wsovod_stem_im2col: null pointer");
  WS_CHECK_ARG(out_dtype == WSOVOD_F32 || out_dtype == WSOVOD_BF16, "wsovod_stem_im2col: bad dtype");
  static int slot = wsovod::prof_slot("stem_im2col");
  hipStream_t s = (hipStream_t)stream;
  const int Ho = (Hp - 1) / 2 + 1, Wo = (Wp - 1) / 2 + 1;
  const long long total = (long long)N * Ho * Wo * 32;
  wsovod::ProfScope prof(slot, s, 0.0, (double)N * 3 * Hp * Wp + total * (out_dtype == WSOVOD_BF16 ? 2.0 : 4.0));
  if (out_dtype == WSOVOD_BF16)
    hipLaunchKernelGGL(stem_im2col_kernel<bf16_t>, dim3(grid_for(total, 256)), dim3(256), 0, s, img, sizes,
                       mean_host[0], mean_host[1], mean_host[2], std_host[0], std_host[1], std_host[2], N, Hp, Wp, Ho,
                       Wo, (bf16_t*)out);
  else
    hipLaunchKernelGGL(stem_im2col_kernel<float>, dim3(grid_for(total, 256)), dim3(256), 0, s, img, sizes,
                       mean_host[0], mean_host[1], mean_host[2], std_host[0], std_host[1], std_host[2], N, Hp, Wp, Ho,
                       Wo, (float*)out);
  WS_CHECK_LAUNCH("wsovod_stem_im2col");
  return WSOVOD_OK;
}

int wsovod_maxpool2x2_nhwc(const void* in, int dtype, int N, int H, int W, int C, int stride, int zero_pad_br,
                           void* out, wsovod_stream_t stream) {
  WS_CHECK_ARG(dtype == WSOVOD_F32 || dtype == WSOVOD_BF16 || dtype == WSOVOD_BF16X2, "wsovod_maxpool2x2_nhwc: bad dtype");
  WS_CHECK_ARG(stride == 1 || stride == 2, "wsovod_maxpool2x2_nhwc: stride must be 1 or 2");
  const int V = dtype == WSOVOD_BF16 ? 8 : dtype == WSOVOD_BF16X2 ? 32 : 4;
  WS_CHECK_ARG(C % V == 0, "wsovod_maxpool2x2_nhwc: C=%d must be a multiple of %d", C, V);
  const int Hin = H + (zero_pad_br ? 1 : 0), Win = W + (zero_pad_br ? 1 : 0);
  const int Ho = (Hin - 2) / stride + 1, Wo = (Win - 2) / stride + 1;
  if (N == 0 || Ho <= 0 || Wo <= 0) return WSOVOD_OK;
  WS_CHECK_ARG(in && out, "wsovod_maxpool2x2_nhwc: null pointer");
  static int slot = wsovod::prof_slot("maxpool2x2_nhwc");
  hipStream_t s = (hipStream_t)stream;
  const long long total = (long long)N * Ho * Wo * (C / V);
  const double esz = dtype == WSOVOD_BF16 ? 2.0 : 4.0;
  wsovod::ProfScope prof(slot, s, 0.0, ((double)N * H * W * C + (double)N * Ho * Wo * C) * esz);
  if (dtype == WSOVOD_BF16X2)
    wsovod::x2_maxpool2x2(in, N, H, W, C, Ho, Wo, stride, zero_pad_br, out, s);
  else if (dtype == WSOVOD_BF16)
    hipLaunchKernelGGL(maxpool2x2_nhwc_kernel<bf16_t>, dim3(grid_for(total, 256)), dim3(256), 0, s, (const bf16_t*)in,
                       N, H, W, C, Ho, Wo, stride, zero_pad_br, (bf16_t*)out);
  else
    hipLaunchKernelGGL(maxpool2x2_nhwc_kernel<float>, dim3(grid_for(total, 256)), dim3(256), 0, s, (const float*)in, N,
                       H, W, C, Ho, Wo, stride, zero_pad_br, (float*)out);
  WS_CHECK_LAUNCH("wsovod_maxpool2x2_nhwc");
  return WSOVOD_OK;
}

int wsovod_maxpool2x2_nhwc_backward(const void* in, int dtype, int N, int H, int W, int C, int stride, int zero_pad_br,
                                    const float* dout, float* din, wsovod_stream_t stream) {
  WS_CHECK_ARG(dtype == WSOVOD_F32 || dtype == WSOVOD_BF16 || dtype == WSOVOD_BF16X2, "wsovod_maxpool2x2_nhwc_backward: bad dtype");
  WS_CHECK_ARG(stride == 1 || stride == 2, "wsovod_maxpool2x2_nhwc_backward: stride must be 1 or 2");
  WS_CHECK_ARG(C % (dtype == WSOVOD_BF16X2 ? 32 : 4) == 0, "wsovod_maxpool2x2_nhwc_backward: C=%d must be a multiple of 4 (bf16x2: 32)", C);
  const int Hin = H + (zero_pad_br ? 1 : 0), Win = W + (zero_pad_br ? 1 : 0);
  const int Ho = (Hin - 2) / stride + 1, Wo = (Win - 2) / stride + 1;
  if (N == 0 || H == 0 || W == 0) return WSOVOD_OK;
  WS_CHECK_ARG(in && dout && din, "wsovod_maxpool2x2_nhwc_backward: null pointer");
  static int slot = wsovod::prof_slot("maxpool2x2_nhwc_bwd");
  hipStream_t s = (hipStream_t)stream;
  const long long total = (long long)N * H * W * (C / 4);
  wsovod::ProfScope prof(slot, s, 0.0, (double)N * H * W * C * 12.0);
  const dim3 grid(grid_for(total, 256));
  if (dtype == WSOVOD_BF16X2)
    hipLaunchKernelGGL((maxpool2x2_bwd_kernel<bf16_t, true>), grid, dim3(256), 0, s, (const bf16_t*)in, N, H, W, C, Ho, Wo, stride,
                       zero_pad_br, dout, din);
  else if (dtype == WSOVOD_BF16)
    hipLaunchKernelGGL((maxpool2x2_bwd_kernel<bf16_t, false>), grid, dim3(256), 0, s, (const bf16_t*)in, N, H, W, C, Ho, Wo, stride,
                       zero_pad_br, dout, din);
  else
    hipLaunchKernelGGL((maxpool2x2_bwd_kernel<float, false>), grid, dim3(256), 0, s, (const float*)in, N, H, W, C, Ho, Wo, stride,
                       zero_pad_br, dout, din);
  WS_CHECK_LAUNCH("wsovod_maxpool2x2_nhwc_backward");
  return WSOVOD_OK;
}

static int launch_colsum(const void* x, int dtype, long long ld, const int* seg, int G, int uniform_rows, int M, int N,
                         float scale, float* out, long long ldo, int accumulate, float* workspace, hipStream_t s) {
  const int chunks = ceil_div(M, CS_RC);
  SegView sv{seg, G, uniform_rows, M};
  const dim3 grid(ceil_div(N, 64 * (dtype == WSOVOD_BF16 ? 8 : 4)), chunks);
  if (dtype == WSOVOD_BF16)
    hipLaunchKernelGGL(segment_colsum_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)x, ld, sv, N, chunks, workspace);
  else
    hipLaunchKernelGGL(segment_colsum_kernel<float>, grid, dim3(256), 0, s, (const float*)x, ld, sv, N, chunks, workspace);
  hipLaunchKernelGGL(colsum_finalize_kernel, dim3((unsigned)(((size_t)G * N + 255) / 256)), dim3(256), 0, s,
                     (const float*)workspace, sv, N, chunks, scale, out, ldo, accumulate);
  return 0;
}

long long wsovod_colsum_workspace_floats(int G, int M, int N) {
  return (2ll * ceil_div(std::max(M, 0), CS_RC) + std::max(G, 0)) * std::max(N, 0);
}

int wsovod_global_avgpool_nhwc(const void* in, int dtype, int N, int HW, int C, float* out, float* workspace,
                               wsovod_stream_t stream) {
  WS_CHECK_ARG(dtype == WSOVOD_F32 || dtype == WSOVOD_BF16, "wsovod_global_avgpool_nhwc: bad dtype");
  if (N == 0) return WSOVOD_OK;
  WS_CHECK_ARG(in && out && workspace && HW > 0 && C > 0, "wsovod_global_avgpool_nhwc: bad argument");
  WS_CHECK_ARG((long long)N * HW < (1ll << 31), "wsovod_global_avgpool_nhwc: more than 2^31 pixels");
  static int slot = wsovod::prof_slot("gap_nhwc");
  hipStream_t s = (hipStream_t)stream;
  wsovod::ProfScope prof(slot, s, 0.0, (double)N * HW * C * (dtype == WSOVOD_BF16 ? 2.0 : 4.0));
  if (launch_colsum(in, dtype, C, nullptr, N, HW, N * HW, C, 1.0f / (float)HW, out, C, 0, workspace, s) != 0) return WSOVOD_ERR_HIP;
  WS_CHECK_LAUNCH("wsovod_global_avgpool_nhwc");
  return WSOVOD_OK;
}

int wsovod_transpose_cast(const void* src, int src_dtype, long long ld_src, int R, int C, void* dst, int dst_dtype,
                          long long ld_dst, wsovod_stream_t stream) {
  if (R == 0 || C == 0) return WSOVOD_OK;
  WS_CHECK_ARG(src && dst, "wsovod_transpose_cast: null pointer");
  static int slot = wsovod::prof_slot("transpose_cast");
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid(ceil_div(C, 64), ceil_div(R, 64));
  wsovod::ProfScope prof(slot, s, 0.0, (double)R * C * ((src_dtype == WSOVOD_BF16 ? 2.0 : 4.0) + (dst_dtype == WSOVOD_BF16 ? 2.0 : 4.0)));
#define TC(TS, TD) hipLaunchKernelGGL((transpose_cast_kernel<TS, TD>), grid, dim3(256), 0, s, (const TS*)src, ld_src, R, C, (TD*)dst, ld_dst)
  if (src_dtype == WSOVOD_F32 && dst_dtype == WSOVOD_F32) TC(float, float);
  else if (src_dtype == WSOVOD_F32 && dst_dtype == WSOVOD_BF16) TC(float, bf16_t);
  else if (src_dtype == WSOVOD_BF16 && dst_dtype == WSOVOD_F32) TC(bf16_t, float);
  else if (src_dtype == WSOVOD_BF16 && dst_dtype == WSOVOD_BF16) TC(bf16_t, bf16_t);
  else { wsovod::set_error("wsovod_transpose_cast: bad dtype"); return WSOVOD_ERR_INVALID_ARGUMENT; }
#undef TC
  WS_CHECK_LAUNCH("wsovod_transpose_cast");
  return WSOVOD_OK;
}

int wsovod_cast(const void* src, int src_dtype, void* dst, int dst_dtype, long long n, wsovod_stream_t stream) {
  if (n == 0) return WSOVOD_OK;
  WS_CHECK_ARG(src && dst, "wsovod_cast: null pointer");
  static int slot = wsovod::prof_slot("cast");
  hipStream_t s = (hipStream_t)stream;
  wsovod::ProfScope prof(slot, s, 0.0, (double)n * ((src_dtype == WSOVOD_BF16 ? 2.0 : 4.0) + (dst_dtype == WSOVOD_BF16 ? 2.0 : 4.0)));
  const int grid = grid_for(n, 256);
  if (src_dtype == WSOVOD_F32 && dst_dtype == WSOVOD_BF16)
    hipLaunchKernelGGL((cast_kernel<float, bf16_t>), dim3(grid), dim3(256), 0, s, (const float*)src, (bf16_t*)dst, n);
  else if (src_dtype == WSOVOD_BF16 && dst_dtype == WSOVOD_F32)
    hipLaunchKernelGGL((cast_kernel<bf16_t, float>), dim3(grid), dim3(256), 0, s, (const bf16_t*)src, (float*)dst, n);
  else if (src_dtype == WSOVOD_F32 && dst_dtype == WSOVOD_F32)
    hipLaunchKernelGGL((cast_kernel<float, float>), dim3(grid), dim3(256), 0, s, (const float*)src, (float*)dst, n);
  else if (src_dtype == WSOVOD_BF16 && dst_dtype == WSOVOD_BF16)
    hipLaunchKernelGGL((cast_kernel<bf16_t, bf16_t>), dim3(grid), dim3(256), 0, s, (const bf16_t*)src, (bf16_t*)dst, n);
  else { wsovod::set_error("wsovod_cast: bad dtype"); return WSOVOD_ERR_INVALID_ARGUMENT; }
  WS_CHECK_LAUNCH("wsovod_cast");
  return WSOVOD_OK;
}

int wsovod_split3_bf16(const float* src, long long ld_src, int rows, int cols, void* dst, long long ld_dst,
                       long long block_stride, int side, wsovod_stream_t stream) {
  if (rows == 0 || cols == 0) return WSOVOD_OK;
  WS_CHECK_ARG(src && dst && rows > 0 && cols > 0 && ld_src >= cols && ld_dst >= cols && (side == 0 || side == 1),
               "wsovod_split3_bf16: bad argument");
  static int slot = wsovod::prof_slot("split3_bf16");
  hipStream_t s = (hipStream_t)stream;
  const long long groups = (long long)rows * ((cols + 7) / 8);
  wsovod::ProfScope prof(slot, s, 0.0, (double)rows * cols * 10.0);
  hipLaunchKernelGGL(split3_bf16_kernel, dim3(grid_for(groups, 256)), dim3(256), 0, s, src, ld_src, rows, cols,
                     (bf16_t*)dst, ld_dst, block_stride, side);
  WS_CHECK_LAUNCH("wsovod_split3_bf16");
  return WSOVOD_OK;
}

int wsovod_row_l2norm_scale(const void* x, int dtype, long long ld, int M, int D, float temperature, float eps,
                            float* row_scale, wsovod_stream_t stream) {
  if (M == 0) return WSOVOD_OK;
  WS_CHECK_ARG(x && row_scale && D > 0, "wsovod_row_l2norm_scale: bad argument");
  static int slot = wsovod::prof_slot("row_l2norm");
  hipStream_t s = (hipStream_t)stream;
  wsovod::ProfScope prof(slot, s, 0.0, (double)M * D * (dtype == WSOVOD_BF16 ? 2.0 : 4.0));
  if (dtype == WSOVOD_BF16)
    hipLaunchKernelGGL(row_l2norm_kernel<bf16_t>, dim3(ceil_div(M, 4)), dim3(256), 0, s, (const bf16_t*)x, ld, M, D, temperature, eps, row_scale);
  else
    hipLaunchKernelGGL(row_l2norm_kernel<float>, dim3(ceil_div(M, 4)), dim3(256), 0, s, (const float*)x, ld, M, D, temperature, eps, row_scale);
  WS_CHECK_LAUNCH("wsovod_row_l2norm_scale");
  return WSOVOD_OK;
}

int wsovod_row_l2norm_backward(const void* z, int dtype, long long ldz, const float* u, long long ldu, int M, int D,
                               float temperature, float eps, int relu_mask, float* dz, long long lddz,
                               wsovod_stream_t stream) {
  if (M == 0) return WSOVOD_OK;
  WS_CHECK_ARG(z && u && dz && D > 0, "wsovod_row_l2norm_backward: bad argument");
  static int slot = wsovod::prof_slot("row_l2norm_bwd");
  hipStream_t s = (hipStream_t)stream;
  wsovod::ProfScope prof(slot, s, 0.0, (double)M * D * 12.0);
  if (dtype == WSOVOD_BF16)
    hipLaunchKernelGGL(row_l2norm_bwd_kernel<bf16_t>, dim3(ceil_div(M, 4)), dim3(256), 0, s, (const bf16_t*)z, ldz, u, ldu, M, D, temperature, eps, relu_mask, dz, lddz);
  else
    hipLaunchKernelGGL(row_l2norm_bwd_kernel<float>, dim3(ceil_div(M, 4)), dim3(256), 0, s, (const float*)z, ldz, u, ldu, M, D, temperature, eps, relu_mask, dz, lddz);
  WS_CHECK_LAUNCH("wsovod_row_l2norm_backward");
  return WSOVOD_OK;
}

int wsovod_segment_colsum(const void* x, int dtype, long long ld, const int* seg_offsets, int G, int M, int N,
                          float* out, long long ldo, int accumulate, float* workspace, wsovod_stream_t stream) {
  if (G == 0 || N == 0) return WSOVOD_OK;
  WS_CHECK_ARG(x && out && seg_offsets && (workspace || M == 0), "wsovod_segment_colsum: null pointer");
  WS_CHECK_ARG(dtype == WSOVOD_F32 || dtype == WSOVOD_BF16, "wsovod_segment_colsum: bad dtype");
  static int slot = wsovod::prof_slot("segment_colsum");
  hipStream_t s = (hipStream_t)stream;
  wsovod::ProfScope prof(slot, s, 0.0, (double)M * N * (dtype == WSOVOD_BF16 ? 2.0 : 4.0));
  if (M > 0) {
    if (launch_colsum(x, dtype, ld, seg_offsets, G, 0, M, N, 1.0f, out, ldo, accumulate, workspace, s) != 0) return WSOVOD_ERR_HIP;
  }
  else if (!accumulate) (void)hipMemsetAsync(out, 0, sizeof(float) * (size_t)G * ldo, s);
  WS_CHECK_LAUNCH("wsovod_segment_colsum");
  return WSOVOD_OK;
}

int wsovod_scale_by_device_scalar(float* x, long long n, const float* num, const float* den, wsovod_stream_t stream) {
  if (n == 0) return WSOVOD_OK;
  WS_CHECK_ARG(x, "wsovod_scale_by_device_scalar: null pointer");
  static int slot = wsovod::prof_slot("scale_by_scalar");
  hipStream_t s = (hipStream_t)stream;
  wsovod::ProfScope prof(slot, s, 0.0, (double)n * 8.0);
  hipLaunchKernelGGL(scale_by_device_scalar_kernel, dim3(grid_for(n, 256)), dim3(256), 0, s, x, n, num, den);
  WS_CHECK_LAUNCH("wsovod_scale_by_device_scalar");
  return WSOVOD_OK;
}

int wsovod_sgd_momentum(float* param, const float* grad, float* momentum_buf, long long n, float lr, float momentum,
                        float weight_decay, float grad_scale, void* bf16_shadow, wsovod_stream_t stream) {
  if (n == 0) return WSOVOD_OK;
  WS_CHECK_ARG(param && grad && momentum_buf, "wsovod_sgd_momentum: null pointer");
  WS_CHECK_ARG((((uintptr_t)param | (uintptr_t)grad | (uintptr_t)momentum_buf) & 15) == 0,
               "wsovod_sgd_momentum: param/grad/momentum must be 16-byte aligned");
  WS_CHECK_ARG(!bf16_shadow || ((uintptr_t)bf16_shadow & 7) == 0, "wsovod_sgd_momentum: shadow must be 8-byte aligned");
  static int slot = wsovod::prof_slot("sgd_momentum");
  hipStream_t s = (hipStream_t)stream;
  wsovod::ProfScope prof(slot, s, 3.0 * n, (double)n * (20.0 + (bf16_shadow ? 2.0 : 0.0)));
  hipLaunchKernelGGL(sgd_momentum_kernel, dim3(grid_for(n / 4 + 1, 256)), dim3(256), 0, s, param, grad, momentum_buf,
                     n, lr, momentum, weight_decay, grad_scale, (bf16_t*)bf16_shadow);
  WS_CHECK_LAUNCH("wsovod_sgd_momentum");
  return WSOVOD_OK;
}

int wsovod_sgd_momentum_multi(const wsovod_sgd_tensor* tensors, int count, float momentum, float grad_scale,
                              wsovod_stream_t stream) {
  WS_CHECK_ARG(count >= 0 && (count == 0 || tensors), "wsovod_sgd_momentum_multi: bad table");
  static int slot = wsovod::prof_slot("sgd_momentum_multi");
  hipStream_t s = (hipStream_t)stream;
  for (int start = 0; start < count; start += kSgdMax) {
    SgdTable t;
    memset(&t, 0, sizeof(t));
    long long total = 0;
    int blocks = 0;
    for (int k = 0; k < kSgdMax && start + k < count; ++k) {
      const wsovod_sgd_tensor& d = tensors[start + k];
      WS_CHECK_ARG(d.numel >= 0 && (d.numel == 0 || (d.param && d.grad && d.momentum_buf)),
                   "wsovod_sgd_momentum_multi: null pointer in entry %d", start + k);
      t.p[k] = d.param;
      t.g[k] = d.grad;
      t.buf[k] = d.momentum_buf;
      t.shadow[k] = (bf16_t*)d.bf16_shadow;
      if (d.grad_is_bf16) t.g_bf16 |= 1u << k;
      if (d.shadow_is_bf16x2 && d.bf16_shadow) {
        WS_CHECK_ARG(d.numel % 32 == 0 && ((uintptr_t)d.bf16_shadow & 15) == 0,
                     "wsovod_sgd_momentum_multi: a bf16x2 / f16mx shadow needs numel a multiple of 32 and 16-byte alignment");
        if (d.shadow_is_bf16x2 == 2) {
          WS_CHECK_ARG(d.mx_scale, "wsovod_sgd_momentum_multi: an f16mx shadow needs its per-tensor scale byte (entry %d)", start + k);
          t.mx_shadow |= 1u << k;
          t.mx_scale[k] = d.mx_scale;
        } else {
          t.x2_shadow |= 1u << k;
        }
      }
      t.used[k] = d.used_flag;
      t.coef[k] = d.grad_coef;
      t.clip[k] = d.clip_value;
      t.lr_dev[k] = d.lr_dev;
      t.n[k] = d.numel;
      t.lr[k] = d.lr;
      t.wd[k] = d.weight_decay;
      t.first_block[k] = blocks;
      blocks += (int)ceil_div_ll(d.numel, kSgdChunk);
      total += d.numel;
      t.count = k + 1;
    }
    t.first_block[t.count] = blocks;
    if (blocks == 0) continue;
    wsovod::ProfScope prof(slot, s, 3.0 * total, (double)total * 22.0);
    hipLaunchKernelGGL(sgd_momentum_multi_kernel, dim3(blocks), dim3(256), 0, s, t, momentum, grad_scale);
    WS_CHECK_LAUNCH("wsovod_sgd_momentum_multi");
  }
  return WSOVOD_OK;
}

long long wsovod_grad_clip_workspace_floats(const wsovod_sgd_tensor* tensors, int count) {
  long long blocks = 0;
  for (int k = 0; k < count; ++k) blocks += ceil_div_ll(tensors[k].numel, kSgdChunk);
  return blocks + count;
}

int wsovod_grad_clip_coef(const wsovod_sgd_tensor* tensors, int count, float grad_scale, float max_norm, int per_tensor,
                          float* workspace, float* coef, wsovod_stream_t stream) {
  WS_CHECK_ARG(count >= 0 && (count == 0 || (tensors && workspace && coef)), "wsovod_grad_clip_coef: bad arguments");
  WS_CHECK_ARG(max_norm > 0.f, "wsovod_grad_clip_coef: max_norm must be positive");
  if (count == 0) return WSOVOD_OK;
  static int slot = wsovod::prof_slot("grad_clip_coef");
  hipStream_t s = (hipStream_t)stream;
  long long total = 0, all_blocks = 0;
  for (int k = 0; k < count; ++k) all_blocks += ceil_div_ll(tensors[k].numel, kSgdChunk);
  float* sumsq = workspace + all_blocks;
  long long block0 = 0;
  for (int k = 0; k < count; ++k) total += tensors[k].numel;
  wsovod::ProfScope prof(slot, s, 2.0 * total, (double)total * 4.0);
  for (int start = 0; start < count; start += kSgdMax) {
    SumsqTable t;
    memset(&t, 0, sizeof(t));
    int blocks = 0;
    for (int k = 0; k < kSgdMax && start + k < count; ++k) {
      const wsovod_sgd_tensor& d = tensors[start + k];
      WS_CHECK_ARG(d.numel >= 0 && (d.numel == 0 || d.grad), "wsovod_grad_clip_coef: null gradient in entry %d", start + k);
      t.g[k] = d.grad;
      t.n[k] = d.numel;
      t.used[k] = d.used_flag;
      if (d.grad_is_bf16) t.g_bf16 |= 1u << k;
      t.first_block[k] = blocks;
      blocks += (int)ceil_div_ll(d.numel, kSgdChunk);
      t.count = k + 1;
    }
    t.first_block[t.count] = blocks;
    if (blocks > 0) {
      hipLaunchKernelGGL(grad_sumsq_partial_kernel, dim3(blocks), dim3(256), 0, s, t, workspace + block0);
      WS_CHECK_LAUNCH("wsovod_grad_clip_coef (partials)");
    }
    hipLaunchKernelGGL(grad_sumsq_tensor_kernel, dim3(t.count), dim3(64), 0, s, t, workspace + block0, sumsq + start);
    WS_CHECK_LAUNCH("wsovod_grad_clip_coef (tensors)");
    block0 += blocks;
  }
  hipLaunchKernelGGL(grad_clip_coef_kernel, dim3(1), dim3(64), 0, s, sumsq, count, grad_scale, max_norm, per_tensor, coef);
  WS_CHECK_LAUNCH("wsovod_grad_clip_coef");
  return WSOVOD_OK;
}

int wsovod_pack_bf16_multi(const wsovod_pack_tensor* tensors, int count, wsovod_stream_t stream) {
  WS_CHECK_ARG(count >= 0 && (count == 0 || tensors), "wsovod_pack_bf16_multi: bad table");
  static int slot = wsovod::prof_slot("pack_bf16_multi");
  hipStream_t s = (hipStream_t)stream;
  for (int start = 0; start < count; start += kSgdMax) {
    PackTable t;
    memset(&t, 0, sizeof(t));
    long long total = 0;
    int blocks = 0;
    for (int k = 0; k < kSgdMax && start + k < count; ++k) {
      const wsovod_pack_tensor& d = tensors[start + k];
      WS_CHECK_ARG(d.numel >= 0 && (d.numel == 0 || (d.src && d.dst)), "wsovod_pack_bf16_multi: null pointer in entry %d",
                   start + k);
      t.src[k] = d.src;
      t.dst[k] = (bf16_t*)d.dst;
      t.n[k] = d.numel;
      t.first_block[k] = blocks;
      blocks += (int)ceil_div_ll(d.numel, kSgdChunk);
      total += d.numel;
      t.count = k + 1;
    }
    t.first_block[t.count] = blocks;
    if (blocks == 0) continue;
    wsovod::ProfScope prof(slot, s, 0.0, (double)total * 6.0);
    hipLaunchKernelGGL(pack_bf16_multi_kernel, dim3(blocks), dim3(256), 0, s, t);
    WS_CHECK_LAUNCH("wsovod_pack_bf16_multi");
  }
  return WSOVOD_OK;
}

int wsovod_sum_shards_bf16(const void* src, int n_shards, long long shard_elems, void* dst, wsovod_stream_t stream) {
  WS_CHECK_ARG(n_shards >= 1 && shard_elems >= 0 && (shard_elems == 0 || (src && dst)), "wsovod_sum_shards_bf16: bad arguments");
  WS_CHECK_ARG((shard_elems & 7) == 0 && ((uintptr_t)src & 15) == 0 && ((uintptr_t)dst & 15) == 0,
               "wsovod_sum_shards_bf16: shards are whole 16-byte groups (shard_elems %% 8 == 0, 16-byte aligned pointers)");
  if (shard_elems == 0) return WSOVOD_OK;
  static int slot = wsovod::prof_slot("sum_shards_bf16");
  hipStream_t s = (hipStream_t)stream;
  const long long groups = shard_elems / 8;
  const int blocks = (int)std::min<long long>(ceil_div_ll(groups, 256), 256 * 8);
  wsovod::ProfScope prof(slot, s, 0.0, (double)shard_elems * 2.0 * (n_shards + 1));
  hipLaunchKernelGGL(sum_shards_bf16_kernel, dim3(blocks), dim3(256), 0, s, (const bf16_t*)src, n_shards, shard_elems,
                     (bf16_t*)dst);
  WS_CHECK_LAUNCH("wsovod_sum_shards_bf16");
  return WSOVOD_OK;
}

}  // extern "C"

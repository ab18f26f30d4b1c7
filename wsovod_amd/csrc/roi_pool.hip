// RoI max pooling and ROIAlign over precomputed (SAM) proposals.
//
// Replaces torchvision.ops.RoIPool / detectron2.layers.ROIAlign as dispatched by
// wsovod/modeling/poolers.py:169-197,284 and the reference's own native op
// (wsovod/layers/ROILoopPool/ROILoopPool_cpu.cpp:13-123 is the algorithm restated here).
//
// NHWC kernels (the fast path, fed by the HIP backbone): one WAVEFRONT owns one
// (roi, 64-channel group).  Lane = channel, so every feature read is one coalesced
// 256-B (fp32) / 128-B (bf16) wave access; the 49 bin results per lane are transposed
// through a per-wavefront LDS tile ([64][49], stride 49 words = conflict-free) and leave
// as one contiguous, 16-B-per-lane coalesced run of the reference's (R,C,7,7) output.
// NCHW kernels (the reference's layout, kept for drop-in use): one thread per output bin.
//
// All bin/index arithmetic is fp32 + int32 exactly as the reference writes it
// (round half away from zero, floor/ceil of fp32 products), so argmax is bit-exact.
#include <float.h>

#include "common.h"
#include "f16mx.h"
#include <type_traits>

// Index results (bins, keep sets, labels) must match the reference bit for bit: no mul+add fusion anywhere in this
// file (HIP's __fmul_rn & co. are plain operators and would still be contracted under the default fp-contract=fast).
#pragma clang fp contract(off)

namespace {

struct RoiBox {
  int batch, start_w, start_h, roi_w, roi_h;
  float bin_h, bin_w;
};

// ROILoopPool_cpu.cpp:27-39
__device__ __forceinline__ RoiBox decode_roi(const float* roi, float spatial_scale, int ph, int pw) {
  RoiBox b;
  b.batch = (int)roi[0];
  b.start_w = (int)roundf(roi[1] * spatial_scale);
  b.start_h = (int)roundf(roi[2] * spatial_scale);
  const int end_w = (int)roundf(roi[3] * spatial_scale);
  const int end_h = (int)roundf(roi[4] * spatial_scale);
  b.roi_w = max(end_w - b.start_w + 1, 1);
  b.roi_h = max(end_h - b.start_h + 1, 1);
  b.bin_h = (float)b.roi_h / (float)ph;
  b.bin_w = (float)b.roi_w / (float)pw;
  return b;
}

// ROILoopPool_cpu.cpp:41-51
__device__ __forceinline__ void bin_window(const RoiBox& b, int ph, int pw, int H, int W, int& hs, int& he,
                                           int& ws, int& we) {
  hs = (int)floorf((float)ph * b.bin_h);
  ws = (int)floorf((float)pw * b.bin_w);
  he = (int)ceilf((float)(ph + 1) * b.bin_h);
  we = (int)ceilf((float)(pw + 1) * b.bin_w);
  hs = min(max(hs + b.start_h, 0), H);
  he = min(max(he + b.start_h, 0), H);
  ws = min(max(ws + b.start_w, 0), W);
  we = min(max(we + b.start_w, 0), W);
}

// ---------------------------------------------------------------------------------
// RoIPool forward, NHWC: wavefront per (roi, 64 channels)
// ---------------------------------------------------------------------------------
template <typename T, bool ARGMAX>
__global__ __launch_bounds__(256) void roi_pool_fwd_nhwc(const T* __restrict__ feat, const float* __restrict__ rois,
                                                         const float* __restrict__ roi_scale, int R, int C, int H,
                                                         int W, int PH, int PW, float spatial_scale, void* out,
                                                         int out_dtype, int* __restrict__ argmax, int cgroups) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nbins = PH * PW;
  const int per_wave_words = 64 * nbins * (ARGMAX ? 2 : 1);
  float* sval = (float*)smem + wave * per_wave_words;
  int* sarg = (int*)(sval + 64 * nbins);
  const long long item = (long long)blockIdx.x * 4 + wave;
  if (item >= (long long)R * cgroups) return;  // whole wavefront exits together
  const int r = (int)(item / cgroups);
  const int c0 = (int)(item % cgroups) * 64;
  const int c = c0 + lane;
  const bool c_ok = c < C;
  const RoiBox b = decode_roi(rois + (long long)r * 5, spatial_scale, PH, PW);
  const float scale = roi_scale ? roi_scale[r] : 1.0f;
  const T* base = feat + (long long)b.batch * H * W * C + (c_ok ? c : 0);
  for (int ph = 0; ph < PH; ++ph) {
    for (int pw = 0; pw < PW; ++pw) {
      int hs, he, ws, we;
      bin_window(b, ph, pw, H, W, hs, he, ws, we);
      const bool empty = (he <= hs) || (we <= ws);
      float maxval = empty ? 0.f : -FLT_MAX;
      int maxidx = -1;
      for (int h = hs; h < he; ++h) {
        const T* row = base + (long long)h * W * C;
        for (int w = ws; w < we; ++w) {
          const float v = to_f32(row[(long long)w * C]);
          if (v > maxval) {
            maxval = v;
            maxidx = h * W + w;
          }
        }
      }
      const int bin = ph * PW + pw;
      sval[lane * nbins + bin] = roi_scale ? maxval * scale : maxval;
      if (ARGMAX) sarg[lane * nbins + bin] = maxidx;
    }
  }
  // wavefront-private LDS tile: no barrier needed beyond the wave's own ordering
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  const int nvalid = min(64, C - c0) * nbins;
  const long long obase = ((long long)r * C + c0) * nbins;
  if (out_dtype == WSOVOD_F32) {
    float* o = (float*)out + obase;
    if ((nvalid & 3) == 0 && (obase & 3) == 0) {
      for (int i = lane * 4; i < nvalid; i += 256) *(float4*)(o + i) = *(const float4*)(sval + i);
    } else {
      for (int i = lane; i < nvalid; i += 64) o[i] = sval[i];
    }
  } else {
    bf16_t* o = (bf16_t*)out + obase;
    if ((nvalid & 3) == 0 && (obase & 3) == 0) {
      for (int i = lane * 4; i < nvalid; i += 256) {
        const float4 v = *(const float4*)(sval + i);
        bf16x4 pk = {(bf16_t)v.x, (bf16_t)v.y, (bf16_t)v.z, (bf16_t)v.w};
        *(bf16x4*)(o + i) = pk;
      }
    } else {
      for (int i = lane; i < nvalid; i += 64) o[i] = (bf16_t)sval[i];
    }
  }
  if (ARGMAX) {
    int* o = argmax + obase;
    if ((nvalid & 3) == 0 && (obase & 3) == 0) {
      for (int i = lane * 4; i < nvalid; i += 256) *(int4*)(o + i) = *(const int4*)(sarg + i);
    } else {
      for (int i = lane; i < nvalid; i += 64) o[i] = sarg[i];
    }
  }
}

// ---------------------------------------------------------------------------------
// Stride-1 2x2 max of an NHWC map: m2[n][h][w][c] = max over rows {h, min(h+1, H-1)} x columns {w, min(w+1, W-1)},
// NaN cells skipped (fmaxf).  The first level of a range-max hierarchy: the RoIPool rows kernel below covers a bin of
// >= 2 x 2 cells with ceil(nh/2) x ceil(nw/2) windows of this map instead of nh x nw cells.  With 512 rois on a 75 x 100
// map every cell lies in ~30 rois, so one pass over the map (16 bytes per lane) replaces most of the gather's requests.
// ---------------------------------------------------------------------------------
template <typename T, int V>
__global__ __launch_bounds__(256) void max2x2_s1_nhwc(const T* __restrict__ in, T* __restrict__ out, int H, int W, int C,
                                                      long long total_vec) {
  typedef T vec __attribute__((ext_vector_type(V)));
  const int cv = C / V;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total_vec;
       idx += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(idx % cv) * V;
    const long long px = idx / cv;
    const int w = (int)(px % W);
    const int h = (int)((px / W) % H);
    const long long n = px / ((long long)W * H);
    const int h1 = min(h + 1, H - 1), w1 = min(w + 1, W - 1);
    const T* pl = in + n * H * W * C + c;
    const vec a = *(const vec*)(pl + ((long long)h * W + w) * C), b = *(const vec*)(pl + ((long long)h * W + w1) * C);
    const vec d = *(const vec*)(pl + ((long long)h1 * W + w) * C), e = *(const vec*)(pl + ((long long)h1 * W + w1) * C);
    vec o;
#pragma unroll
    for (int q = 0; q < V; ++q)
      o[q] = from_f32<T>(fmaxf(fmaxf(to_f32(a[q]), to_f32(b[q])), fmaxf(to_f32(d[q]), to_f32(e[q]))));
    *(vec*)(out + px * C + c) = o;
  }
}

// The same map in STRIPS of RH rows per thread (round 5): the one-output-per-thread form re-reads every row for the row
// above it (counters: 1.44 GB fetched for the 491-MB map at 32 images); here a thread walks RH rows of its (column, channel
// vector), keeps the horizontal max of the previous row in registers and fetches RH + 1 rows for RH outputs.  Same
// grouping of the four cells -- max(max(a, b), max(d, e)) -- hence the same bits.
template <typename T, int V, int RH>
__global__ __launch_bounds__(256) void max2x2_s1_strip_nhwc(const T* __restrict__ in, T* __restrict__ out, int H, int W, int C,
                                                            long long total) {
  typedef T vec __attribute__((ext_vector_type(V)));
  const int cv = C / V, strips = (H + RH - 1) / RH;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(idx % cv) * V;
    long long t = idx / cv;
    const int w = (int)(t % W);
    t /= W;
    const int sp = (int)(t % strips);
    const long long n = t / strips;
    const int h0 = sp * RH, w1 = min(w + 1, W - 1);
    const T* pl = in + n * H * W * C + c;
    vec a[RH + 1], b[RH + 1];
#pragma unroll
    for (int r = 0; r <= RH; ++r) {
      const int h = min(h0 + r, H - 1);
      a[r] = *(const vec*)(pl + ((long long)h * W + w) * C);
      b[r] = *(const vec*)(pl + ((long long)h * W + w1) * C);
    }
    float hm[RH + 1][V];
#pragma unroll
    for (int r = 0; r <= RH; ++r)
#pragma unroll
      for (int q = 0; q < V; ++q) hm[r][q] = fmaxf(to_f32(a[r][q]), to_f32(b[r][q]));
#pragma unroll
    for (int r = 0; r < RH; ++r) {
      const int h = h0 + r;
      if (h >= H) break;
      vec o;
#pragma unroll
      for (int q = 0; q < V; ++q) o[q] = from_f32<T>(fmaxf(hm[r][q], hm[r + 1][q]));
      *(vec*)(out + ((n * H + h) * W + w) * C + c) = o;
    }
  }
}

// The strip pass FUSED with the global average pool of the same map (the data-aware head's input; round 5): a workgroup
// owns (image, strip, block of PB columns) x all channel vectors, every thread adds up the cells it loads anyway, the PB
// columns are folded through LDS in a fixed order and one partial row per workgroup goes to `part`
// [(image, strip, column block)][C]; max2x2_gap_finalize adds the partial rows of an image in index order.  Run-to-run
// bit-identical (no atomics); the 2x2 maxima are the strip kernel's.
template <typename T, int V, int RH>
__global__ __launch_bounds__(256) void max2x2_s1_strip_gap_nhwc(const T* __restrict__ in, T* __restrict__ out, int H, int W,
                                                                int C, int PB, int wblocks, float* __restrict__ part) {
  typedef T vec __attribute__((ext_vector_type(V)));
  __shared__ float red[256][V + 1];
  const int cv = C / V, strips = (H + RH - 1) / RH;
  const int CB = 256 / PB;  // channel vectors per pass (PB * CB = 256)
  const int pc = threadIdx.x / CB, ci = threadIdx.x - pc * CB;
  const int wb = blockIdx.x % wblocks;
  const int sp = (blockIdx.x / wblocks) % strips;
  const long long n = blockIdx.x / ((long long)wblocks * strips);
  const int w = wb * PB + pc, h0 = sp * RH;
  const bool wok = w < W;
  const int wc = min(w, W - 1), w1 = min(wc + 1, W - 1);
  for (int cp = 0; cp * CB < cv; ++cp) {
    const int cvec = cp * CB + ci;
    const bool ok = wok && cvec < cv;
    const int c = min(cvec, cv - 1) * V;
    float sum[V];
#pragma unroll
    for (int q = 0; q < V; ++q) sum[q] = 0.f;
    if (ok) {
      const T* pl = in + n * H * W * C + c;
      vec a[RH + 1], b[RH + 1];
#pragma unroll
      for (int r = 0; r <= RH; ++r) {
        const int h = min(h0 + r, H - 1);
        a[r] = *(const vec*)(pl + ((long long)h * W + wc) * C);
        b[r] = *(const vec*)(pl + ((long long)h * W + w1) * C);
      }
      float hm[RH + 1][V];
#pragma unroll
      for (int r = 0; r <= RH; ++r)
#pragma unroll
        for (int q = 0; q < V; ++q) hm[r][q] = fmaxf(to_f32(a[r][q]), to_f32(b[r][q]));
#pragma unroll
      for (int r = 0; r < RH; ++r) {
        const int h = h0 + r;
        if (h >= H) break;
        vec o;
#pragma unroll
        for (int q = 0; q < V; ++q) {
          o[q] = from_f32<T>(fmaxf(hm[r][q], hm[r + 1][q]));
          sum[q] += to_f32(a[r][q]);
        }
        *(vec*)(out + ((n * H + h) * W + wc) * C + c) = o;
      }
    }
#pragma unroll
    for (int q = 0; q < V; ++q) red[threadIdx.x][q] = sum[q];
    __syncthreads();
    if (pc == 0 && cvec < cv) {
      float* dst = part + (long long)blockIdx.x * C + c;
#pragma unroll
      for (int q = 0; q < V; ++q) {
        float t = red[ci][q];
        for (int k = 1; k < PB; ++k) t += red[k * CB + ci][q];
        dst[q] = t;
      }
    }
    __syncthreads();
  }
}

// (image, 16 channels) per workgroup: 16 groups of lanes take every 16th partial row (independent loads in flight), LDS folds
// the groups in index order -- a fixed summation order, whatever the launch timing
__global__ __launch_bounds__(256) void max2x2_gap_finalize(const float* __restrict__ part, int per_image, int C, float scale,
                                                           float* __restrict__ out) {
  __shared__ float red[16][17];
  const int cblocks = (C + 15) / 16;
  const int n = blockIdx.x / cblocks, c = (blockIdx.x - n * cblocks) * 16 + (threadIdx.x & 15);
  const int g = threadIdx.x >> 4;
  float t = 0.f;
  if (c < C) {
    const float* p = part + (long long)n * per_image * C + c;
    int j = g;
    for (; j + 48 < per_image; j += 64) {
      const float v0 = p[(long long)j * C], v1 = p[(long long)(j + 16) * C], v2 = p[(long long)(j + 32) * C],
                  v3 = p[(long long)(j + 48) * C];
      t += v0; t += v1; t += v2; t += v3;
    }
    for (; j < per_image; j += 16) t += p[(long long)j * C];
  }
  red[g][threadIdx.x & 15] = t;
  __syncthreads();
  if (g == 0 && c < C) {
    float r = red[0][threadIdx.x];
    for (int k = 1; k < 16; ++k) r += red[k][threadIdx.x];
    out[(long long)n * C + c] = r * scale;
  }
}

// ---------------------------------------------------------------------------------
// RoIPool forward, NHWC, pooled width PWT (7 on every shipped config): one WORKGROUP per
// (roi, 64-channel group), one wavefront per pooled row ph.  The PWT bins of the row are
// scanned together: each inner iteration issues PWT independent coalesced loads (one per
// bin) before any compare, so a wavefront keeps PWT loads in flight instead of one.  The
// scan order inside every bin is still (h ascending, w ascending) with a strict '>' --
// the reference's first-maximum argmax semantics (ROILoopPool_cpu.cpp:63-71).
// ---------------------------------------------------------------------------------
// v_max_f32 as the hardware does it: a NaN operand yields the other one (the reference's `v > maxval` never takes a NaN
// cell either).  fmaxf() means the same but makes hipcc re-canonicalise the loop-carried accumulators every round.
__device__ __forceinline__ float hw_max(float a, float b) {
  float r;
  asm("v_max_f32_e32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ float hw_max3(float a, float b, float c) {
  float r;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }

template <typename T, bool ARGMAX, int PWT, int CPL, bool OBF = false>
__global__ void roi_pool_fwd_nhwc_rows(const T* __restrict__ feat, const float* __restrict__ rois,
                                       const float* __restrict__ roi_scale, int C, int H, int W, int PH,
                                       float spatial_scale, void* out, int out_dtype, int* __restrict__ argmax,
                                       int cgroups, void* out_hi = nullptr, const T* __restrict__ m2 = nullptr,
                                       int R = 0, int xcd_per_group = 0) {
  // lane = CPL adjacent channels (one 8- or 16-byte load), workgroup = 64*CPL channels of one roi
  typedef T vec2 __attribute__((ext_vector_type(CPL)));
  constexpr int CG = 64 * CPL;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // the pooled row is the same for all lanes of a wavefront: said so explicitly, every window bound, loop count and row
  // address below lives in scalar registers and the loads take (scalar base, lane offset) addresses
  const int lane = threadIdx.x & 63, ph = uni(threadIdx.x >> 6);
  const int nbins = PH * PWT;
  // output transpose tile [channel][bin]: fp32, or (OBF: bf16 output, no argmax) the output bits themselves: half the
  // LDS per workgroup, twice the workgroups (wavefronts with loads in flight) per CU
  typedef typename std::conditional<OBF, bf16_t, float>::type sval_t;
  sval_t* sval = (sval_t*)smem;
  int* sarg = (int*)(smem + (size_t)CG * nbins * sizeof(sval_t));
  int r = blockIdx.x / cgroups;
  int c0 = (blockIdx.x - r * cgroups) * CG;
  if (xcd_per_group) {
    // XCD-aware order (workgroups go round-robin over the 8 XCDs): channel group g is served by `xcd_per_group`
    // XCDs only, so the slice of the map an XCD's private 4-MB L2 sees is 1 / cgroups of every image
    const int xcd = blockIdx.x & 7, blk = blockIdx.x >> 3;
    c0 = (xcd / xcd_per_group) * CG;
    r = blk * xcd_per_group + xcd % xcd_per_group;
    if (r >= R) return;  // (whole workgroup)
  }
  const int c = c0 + lane * CPL;
  const RoiBox b = decode_roi(rois + (long long)r * 5, spatial_scale, PH, PWT);
  const float scale = roi_scale ? roi_scale[r] : 1.0f;
  const unsigned lane_off = (unsigned)(c < C ? c : 0) * (unsigned)sizeof(T);  // bytes: (scalar row address) + (lane offset)
  const long long img_off = (long long)b.batch * H * W * C;  // (scalar)
  const T* base = feat + img_off;
  int hs, he, ws[PWT], we[PWT];
  float maxv[PWT][CPL];
  int maxi[PWT][CPL];
  int bw = 0;
#pragma unroll
  for (int pw = 0; pw < PWT; ++pw) {
    bin_window(b, ph, pw, H, W, hs, he, ws[pw], we[pw]);
    ws[pw] = uni(ws[pw]);
    we[pw] = uni(we[pw]);
  }
  hs = uni(hs);
  he = uni(he);
#pragma unroll
  for (int pw = 0; pw < PWT; ++pw) {
    const bool empty = (he <= hs) || (we[pw] <= ws[pw]);
#pragma unroll
    for (int q = 0; q < CPL; ++q) {
      maxv[pw][q] = empty ? 0.f : -FLT_MAX;
      maxi[pw][q] = -1;
    }
    bw = max(bw, we[pw] - ws[pw]);
  }
  // Values only, every bin of this pooled row at least 2 x 2 cells: the bin is covered by windows of the stride-1
  // 2x2-max map `m2` (max2x2_s1_nhwc) anchored every second row / column, the last anchor clamped to the bin's end --
  // a quarter of the loads of the cell scan, no per-load predicate.  max() is order-free and the windows cover exactly
  // the bin's cells, NaN cells are skipped at both levels as `v > maxval` skips them: the same values bit for bit.
  bool use_m2 = false;
  constexpr int VB = CPL * (int)sizeof(T);  // bytes per lane and load
  constexpr bool M2OK = !ARGMAX && (VB == 8 || VB == 16);
  if constexpr (M2OK) {
    if (m2 != nullptr && he - hs >= 2) {
      int nwmin = we[0] - ws[0];
#pragma unroll
      for (int pw = 1; pw < PWT; ++pw) nwmin = min(nwmin, we[pw] - ws[pw]);
      use_m2 = nwmin >= 2;
    }
  }
  if constexpr (M2OK) if (use_m2) {
    // U windows per bin in flight (U * PWT independent loads per wavefront before the first compare).
    // Buffer loads: (resource of the image's map, lane offset in a VGPR, window offset in an SGPR) -- no per-load
    // address registers, so the kernel stays within the 64 VGPRs that let 8 wavefronts share a SIMD (a 7-wavefront
    // workgroup is resident 4 times per CU instead of twice: the launch is bound by latency, not by bandwidth).
    constexpr int U = VB == 8 ? 2 : 1;
    typedef int ivec __attribute__((ext_vector_type(VB / 4)));
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(m2 + img_off), 0, H * W * C * (int)sizeof(T), 0x00020000);
    const int nh2 = (he - hs + 1) >> 1, nw2 = (bw + 1) >> 1, nit = nh2 * nw2;
    const int cb = C * (int)sizeof(T), wcb = W * cb;  // byte offsets inside the image's map (< 2^31: launcher), scalar
    int colo[PWT], cmax[PWT];
#pragma unroll
    for (int pw = 0; pw < PWT; ++pw) colo[pw] = ws[pw] * cb, cmax[pw] = (we[pw] - 2) * cb;
    int i = 0, j = 0;
    for (int it = 0; it < nit; it += U) {
      ivec raw[U][PWT];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int rowo = min(hs + 2 * i, he - 2) * wcb, jo = 2 * j * cb;
#pragma unroll
        for (int pw = 0; pw < PWT; ++pw) {
          const int so = rowo + min(colo[pw] + jo, cmax[pw]);
          if constexpr (VB == 8)
            raw[u][pw] = __builtin_bit_cast(ivec, __builtin_amdgcn_raw_buffer_load_b64(rs, (int)lane_off, so, 0));
          else
            raw[u][pw] = __builtin_bit_cast(ivec, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)lane_off, so, 0));
        }
        if (U > 1 && it + u + 1 < nit) {  // (scalar) next window position, the last one repeated (max is idempotent)
          if (++j == nw2) j = 0, ++i;
        }
      }
      if (U == 1) {
        if (++j == nw2) j = 0, ++i;
      }
#pragma unroll
      for (int pw = 0; pw < PWT; ++pw) {
        vec2 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = __builtin_bit_cast(vec2, raw[u][pw]);
#pragma unroll
        for (int q = 0; q < CPL; ++q) {
          if constexpr (U == 2)
            maxv[pw][q] = hw_max3(maxv[pw][q], to_f32(v[0][q]), to_f32(v[1][q]));
          else
            maxv[pw][q] = hw_max(maxv[pw][q], to_f32(v[0][q]));
        }
      }
    }
  }
  if (!use_m2)
  for (int h = hs; h < he; ++h) {
    const T* row = base + (long long)h * W * C;
    for (int j = 0; j < bw; ++j) {
      vec2 v[PWT];
#pragma unroll
      for (int pw = 0; pw < PWT; ++pw) {
        const int w = min(ws[pw] + j, W - 1);  // clamped: lanes past the bin load a valid cell and ignore it
        v[pw] = *(const vec2*)((const char*)(row + (long long)w * C) + lane_off);
      }
#pragma unroll
      for (int pw = 0; pw < PWT; ++pw) {
        const int w = ws[pw] + j;
        if (w < we[pw]) {
#pragma unroll
          for (int q = 0; q < CPL; ++q) {
            const float vq = to_f32(v[pw][q]);
            if (vq > maxv[pw][q]) {
              maxv[pw][q] = vq;
              maxi[pw][q] = h * W + w;
            }
          }
        }
      }
    }
  }
#pragma unroll
  for (int pw = 0; pw < PWT; ++pw)
#pragma unroll
    for (int q = 0; q < CPL; ++q) {
      sval[(lane * CPL + q) * nbins + ph * PWT + pw] = (sval_t)(roi_scale ? maxv[pw][q] * scale : maxv[pw][q]);
      if (ARGMAX) sarg[(lane * CPL + q) * nbins + ph * PWT + pw] = maxi[pw][q];
    }
  __syncthreads();
  const int nthreads = blockDim.x, tid = threadIdx.x;
  const int nvalid = min(CG, C - c0) * nbins;
  const long long obase = ((long long)r * C + c0) * nbins;
  const bool vec = (nvalid & 3) == 0 && (obase & 3) == 0;
  if constexpr (OBF) {  // (launcher: bf16 output, no argmax) the tile already holds the output bits
    bf16_t* o = (bf16_t*)out + obase;
    if ((nvalid & 7) == 0 && (obase & 7) == 0)
      for (int i = tid * 8; i < nvalid; i += nthreads * 8)
        __builtin_nontemporal_store(*(const bf16x8*)(sval + i), (bf16x8*)(o + i));
    else
      for (int i = tid; i < nvalid; i += nthreads) o[i] = sval[i];
    return;
  } else if (out_dtype == WSOVOD_F32) {
    float* o = (float*)out + obase;
    if (vec)
      for (int i = tid * 4; i < nvalid; i += nthreads * 4) *(float4*)(o + i) = *(const float4*)(sval + i);
    else
      for (int i = tid; i < nvalid; i += nthreads) o[i] = sval[i];
  } else if (out_dtype == WSOVOD_F16MX) {
    // unit-scale f16mx (include/wsovod_hip.h, round 6): 8 values per lane = 16 B of fp16 hi + 8 B of e4m3 q + 8 B of e4m3 ql of
    // one 128-byte group, and -- `out_hi` -- the plain bf16 rounding, the operand of the first FC layer's weight gradient
    char* o = (char*)out;
    for (int i = tid * 8; i < nvalid; i += nthreads * 8) {
      wsovod_mx::f16x4 h0, h1;
      int q0, q1, l0, l1;
      const f32x4 v0 = *(const f32x4*)(sval + i), v1 = *(const f32x4*)(sval + i + 4);
      wsovod_mx::mx_enc4_unit(v0, h0, q0, l0);
      wsovod_mx::mx_enc4_unit(v1, h1, q1, l1);
      const long long k = obase + i;
      char* d = o + wsovod_mx::mx_group(k);
      const int w = (int)(k & 31);
      __builtin_nontemporal_store(wsovod_mx::f16x8{h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]}, (wsovod_mx::f16x8*)(d + 2 * w));
      __builtin_nontemporal_store(wsovod_mx::i32x2{q0, q1}, (wsovod_mx::i32x2*)(d + 64 + w));
      __builtin_nontemporal_store(wsovod_mx::i32x2{l0, l1}, (wsovod_mx::i32x2*)(d + 96 + w));
      if (out_hi)
        __builtin_nontemporal_store(bf16x8{(bf16_t)v0[0], (bf16_t)v0[1], (bf16_t)v0[2], (bf16_t)v0[3], (bf16_t)v1[0], (bf16_t)v1[1],
                                           (bf16_t)v1[2], (bf16_t)v1[3]}, (bf16x8*)((bf16_t*)out_hi + k));
    }
  } else if (out_dtype == WSOVOD_BF16X2 || out_dtype == WSOVOD_BF16X2P) {
    // bf16x2 (include/wsovod_hip.h): the run [obase, obase + nvalid) covers whole 32-value groups (launcher); 8 values per
    // lane = 16 B of hi and 16 B of lo half a line further, streamed past L2 like the bf16 form.  PLANAR (round 5): hi to the
    // first bf16 matrix of the output, lo to the second (`out_hi` = its base, set by the launcher): the hi plane is at once
    // the plain bf16 operand of the first FC layer's weight gradient -- no third store per value
    bf16_t* o = (bf16_t*)out;
    const bool planar = out_dtype == WSOVOD_BF16X2P;
    for (int i = tid * 8; i < nvalid; i += nthreads * 8) {
      const f32x4 q0 = *(const f32x4*)(sval + i), q1 = *(const f32x4*)(sval + i + 4);
      bf16x8 hi, lo;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        hi[e] = (bf16_t)q0[e];
        hi[4 + e] = (bf16_t)q1[e];
        const float h0 = (float)hi[e], h1 = (float)hi[4 + e];
        lo[e] = (bf16_t)(__builtin_isinf(h0) ? 0.f : q0[e] - h0);
        lo[4 + e] = (bf16_t)(__builtin_isinf(h1) ? 0.f : q1[e] - h1);
      }
      const long long k = obase + i;
      if (planar) {
        __builtin_nontemporal_store(hi, (bf16x8*)(o + k));
        __builtin_nontemporal_store(lo, (bf16x8*)((bf16_t*)out_hi + k));
        continue;
      }
      bf16_t* d = o + ((k >> 5) << 6) + (k & 31);
      __builtin_nontemporal_store(hi, (bf16x8*)d);
      __builtin_nontemporal_store(lo, (bf16x8*)(d + 32));
      if (out_hi) __builtin_nontemporal_store(hi, (bf16x8*)((bf16_t*)out_hi + k));  // plain bf16 copy (see the launcher)
    }
  } else {
    bf16_t* o = (bf16_t*)out + obase;
    if (vec && (nvalid & 7) == 0 && (obase & 7) == 0)
      for (int i = tid * 8; i < nvalid; i += nthreads * 8) {
        const f32x4 q0 = *(const f32x4*)(sval + i), q1 = *(const f32x4*)(sval + i + 4);
        const bf16x8 pk = {(bf16_t)q0[0], (bf16_t)q0[1], (bf16_t)q0[2], (bf16_t)q0[3],
                           (bf16_t)q1[0], (bf16_t)q1[1], (bf16_t)q1[2], (bf16_t)q1[3]};
        // 16 bytes per lane, non-temporal: the pooled tensor (822 MB at 32 images) is written once and read by the
        // next kernel; it must not evict the feature map the neighbouring rois re-read from L2 (0.896 -> 0.832 ms)
        __builtin_nontemporal_store(pk, (bf16x8*)(o + i));
      }
    else if (vec)
      for (int i = tid * 4; i < nvalid; i += nthreads * 4) {
        const float4 q = *(const float4*)(sval + i);
        bf16x4 pk = {(bf16_t)q.x, (bf16_t)q.y, (bf16_t)q.z, (bf16_t)q.w};
        *(bf16x4*)(o + i) = pk;
      }
    else
      for (int i = tid; i < nvalid; i += nthreads) o[i] = (bf16_t)sval[i];
  }
  if (ARGMAX) {
    int* o = argmax + obase;
    if (vec)
      for (int i = tid * 4; i < nvalid; i += nthreads * 4) *(int4*)(o + i) = *(const int4*)(sarg + i);
    else
      for (int i = tid; i < nvalid; i += nthreads) o[i] = sarg[i];
  }
}

// ---------------------------------------------------------------------------------
// RoIPool forward, NCHW: thread per output bin (grid-stride)
// ---------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void roi_pool_fwd_nchw(const T* __restrict__ feat, const float* __restrict__ rois,
                                                         const float* __restrict__ roi_scale, long long total, int C,
                                                         int H, int W, int PH, int PW, float spatial_scale, void* out,
                                                         int out_dtype, int* __restrict__ argmax, int nhwc) {
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int pw = (int)(idx % PW);
    const int ph = (int)((idx / PW) % PH);
    const int c = (int)((idx / ((long long)PW * PH)) % C);
    const int r = (int)(idx / ((long long)PW * PH * C));
    const RoiBox b = decode_roi(rois + (long long)r * 5, spatial_scale, PH, PW);
    int hs, he, ws, we;
    bin_window(b, ph, pw, H, W, hs, he, ws, we);
    const bool empty = (he <= hs) || (we <= ws);
    float maxval = empty ? 0.f : -FLT_MAX;
    int maxidx = -1;
    // element (h,w) of channel c: plane[(h*W+w)*estride]
    const T* plane = nhwc ? feat + (long long)b.batch * H * W * C + c : feat + ((long long)b.batch * C + c) * H * W;
    const long long estride = nhwc ? C : 1;
    for (int h = hs; h < he; ++h)
      for (int w = ws; w < we; ++w) {
        const float v = to_f32(plane[(long long)(h * W + w) * estride]);
        if (v > maxval) {
          maxval = v;
          maxidx = h * W + w;
        }
      }
    if (roi_scale) maxval *= roi_scale[r];
    if (out_dtype == WSOVOD_F32)
      ((float*)out)[idx] = maxval;
    else
      ((bf16_t*)out)[idx] = (bf16_t)maxval;
    if (argmax) argmax[idx] = maxidx;
  }
}

// ---------------------------------------------------------------------------------
// ROILoopPool, the reference's own native op in its 3-output CUDA form (ROILoopPool_cuda.cu:9-204; the CPU source
// implements only the first output): per roi and bin
//   region : max over the bin (as RoIPool, but with the accumulator starting at 0: inputs are post-ReLU)
//   frame  : the same bin without the cells STRICTLY inside the roi shrunk by context_ratio about its centre
//   context: the bin of the roi GROWN by context_ratio, without the cells strictly inside the roi itself
// out / argmax: (3R, C, PH, PW) = [region | frame | context].
//
// NHWC form of the plain RoIPool rows kernel above (the reference runs one thread per output bin over an NCHW map):
// workgroup = (roi, 64 * CPL channels), wavefront = pooled row, lane = CPL adjacent channels, so every cell is one
// coalesced wave access; the PWT bins of a row advance together (PWT independent loads in flight); region and frame
// share the bins and therefore every cell load -- one scan feeds both accumulators, the frame's hole is a wave-uniform
// test; the context ring is a second scan over the grown roi's bins.  The three (CG, PH, PW) result tiles go through
// one LDS transpose tile each and leave as contiguous runs of the reference's (3R, C, PH, PW) layout.  Scan order per
// bin is (h ascending, w ascending) with a strict '>' from 0: first-maximum argmax, -1 for an empty / all-zero bin.
// ---------------------------------------------------------------------------------
struct LoopRects {
  int batch;
  int sw, sh, ew, eh;      // the roi itself, rounded: bins of region / frame, hole of context
  int swi, shi, ewi, ehi;  // shrunk: hole of frame
  int swo, sho, ewo, eho;  // grown: bins of context
};

__device__ __forceinline__ LoopRects loop_rects(const float* roi, float spatial_scale, float context_ratio, int H, int W) {
  LoopRects q;
  q.batch = (int)roi[0];
  const float x1 = roi[1], y1 = roi[2], x2 = roi[3], y2 = roi[4];
  const float rw = x2 - x1, rh = y2 - y1;
  const float in_rw = rw - rw / context_ratio, in_rh = rh - rh / context_ratio;    // inner residuals
  const float out_rw = rw * context_ratio - rw, out_rh = rh * context_ratio - rh;  // outer residuals
  const float xmax = (float)(1.0 * W / spatial_scale), ymax = (float)(1.0 * H / spatial_scale);
  const float x1i = fminf(fmaxf(x1 + in_rw / 2, 0.f), xmax), y1i = fminf(fmaxf(y1 + in_rh / 2, 0.f), ymax);
  const float x2i = fminf(fmaxf(x2 - in_rw / 2, 0.f), xmax), y2i = fminf(fmaxf(y2 - in_rh / 2, 0.f), ymax);
  const float x1o = fminf(fmaxf(x1 - out_rw / 2, 0.f), xmax), y1o = fminf(fmaxf(y1 - out_rh / 2, 0.f), ymax);
  const float x2o = fminf(fmaxf(x2 + out_rw / 2, 0.f), xmax), y2o = fminf(fmaxf(y2 + out_rh / 2, 0.f), ymax);
  q.sw = (int)roundf(x1 * spatial_scale), q.sh = (int)roundf(y1 * spatial_scale);
  q.ew = (int)roundf(x2 * spatial_scale), q.eh = (int)roundf(y2 * spatial_scale);
  q.swi = (int)roundf(x1i * spatial_scale), q.shi = (int)roundf(y1i * spatial_scale);
  q.ewi = (int)roundf(x2i * spatial_scale), q.ehi = (int)roundf(y2i * spatial_scale);
  q.swo = (int)roundf(x1o * spatial_scale), q.sho = (int)roundf(y1o * spatial_scale);
  q.ewo = (int)roundf(x2o * spatial_scale), q.eho = (int)roundf(y2o * spatial_scale);
  return q;
}

// One pooled row `ph` of the bins of the rectangle (sw, sh, ew, eh), pooled columns [pw0, pw0 + npw): the cells of a bin
// strictly inside (hw0, hh0, hw1, hh1) are skipped by accumulator B (and by accumulator A too when SKIP_A); results go to
// the [channel][bin] LDS tiles.  TWO = both accumulators wanted (region + frame), else only B (context).
template <typename T, int PWT, int CPL, bool TWO>
__device__ __forceinline__ void loop_pool_row(const T* __restrict__ base, int C, int H, int W, int PH, int PW, int ph,
                                              int pw0, int npw, int sw, int sh, int ew, int eh, int hw0, int hh0, int hw1,
                                              int hh1, int lane, float* svalA, int* sargA, float* svalB, int* sargB) {
  typedef T vecc __attribute__((ext_vector_type(CPL)));
  const int roi_w = max(ew - sw + 1, 1), roi_h = max(eh - sh + 1, 1);
  const float bin_h = (float)roi_h / (float)PH, bin_w = (float)roi_w / (float)PW;
  const int hs = min(max((int)floorf((float)ph * bin_h) + sh, 0), H);
  const int he = min(max((int)ceilf((float)(ph + 1) * bin_h) + sh, 0), H);
  int ws[PWT], we[PWT], bw = 0;
  float mvA[PWT][CPL], mvB[PWT][CPL];
  int miA[PWT][CPL], miB[PWT][CPL];
#pragma unroll
  for (int k = 0; k < PWT; ++k) {
    const int pw = min(pw0 + k, PW - 1);
    ws[k] = min(max((int)floorf((float)pw * bin_w) + sw, 0), W);
    we[k] = k < npw ? min(max((int)ceilf((float)(pw + 1) * bin_w) + sw, 0), W) : ws[k];  // (past the chunk: empty)
    bw = max(bw, we[k] - ws[k]);
#pragma unroll
    for (int q = 0; q < CPL; ++q) {
      mvA[k][q] = mvB[k][q] = 0.f;
      miA[k][q] = miB[k][q] = -1;
    }
  }
  for (int h = hs; h < he; ++h) {
    const T* row = base + (long long)h * W * C;
    const bool in_h = h > hh0 && h < hh1;
    for (int j = 0; j < bw; ++j) {
      vecc v[PWT];
#pragma unroll
      for (int k = 0; k < PWT; ++k) v[k] = *(const vecc*)(row + (long long)min(ws[k] + j, W - 1) * C);
#pragma unroll
      for (int k = 0; k < PWT; ++k) {
        const int w = ws[k] + j;
        if (w < we[k]) {
          const bool hole = in_h && w > hw0 && w < hw1;
#pragma unroll
          for (int q = 0; q < CPL; ++q) {
            const float vq = to_f32(v[k][q]);
            if (TWO && vq > mvA[k][q]) {
              mvA[k][q] = vq;
              miA[k][q] = h * W + w;
            }
            if (!hole && vq > mvB[k][q]) {
              mvB[k][q] = vq;
              miB[k][q] = h * W + w;
            }
          }
        }
      }
    }
  }
  const int nbins = PH * PW;
#pragma unroll
  for (int k = 0; k < PWT; ++k)
    if (k < npw) {
#pragma unroll
      for (int q = 0; q < CPL; ++q) {
        const int o = (lane * CPL + q) * nbins + ph * PW + pw0 + k;
        if (TWO && svalA) {
          svalA[o] = mvA[k][q];
          sargA[o] = miA[k][q];
        }
        if (svalB) {
          svalB[o] = mvB[k][q];
          sargB[o] = miB[k][q];
        }
      }
    }
}

template <typename T, int CPL>
__global__ void roi_loop_pool_fwd_nhwc_rows(const T* __restrict__ feat, const float* __restrict__ rois, int R, int C,
                                            int H, int W, int PH, int PW, float spatial_scale, float context_ratio,
                                            float* __restrict__ out, int* __restrict__ argmax, int cgroups, int share) {
  constexpr int PWT = 7, CG = 64 * CPL;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
  const int nbins = PH * PW;
  float* svalA = (float*)smem;  // tile A: region, later context; tile B: frame
  int* sargA = (int*)(svalA + CG * nbins);
  float* svalB = (float*)(sargA + CG * nbins);
  int* sargB = (int*)(svalB + CG * nbins);
  const int r = blockIdx.x / cgroups;
  const int c0 = (blockIdx.x - r * cgroups) * CG;
  const int c = c0 + lane * CPL;
  const LoopRects q = loop_rects(rois + (long long)r * 5, spatial_scale, context_ratio, H, W);
  const T* base = feat + (long long)q.batch * H * W * C + (c + CPL <= C ? c : 0);  // (lanes past C read channel 0, never stored)
  const long long part = (long long)R * C * nbins;
  const int nvalid = min(CG, C - c0) * nbins;
  const long long obase = ((long long)r * C + c0) * nbins;
  const bool vec = (nvalid & 3) == 0 && (obase & 3) == 0;
  auto flush = [&](const float* sv, const int* sa, long long off) {
    float* o = out + off + obase;
    int* oa = argmax + off + obase;
    if (vec) {
      for (int i = threadIdx.x * 4; i < nvalid; i += blockDim.x * 4) {
        *(float4*)(o + i) = *(const float4*)(sv + i);
        *(int4*)(oa + i) = *(const int4*)(sa + i);
      }
    } else {
      for (int i = threadIdx.x; i < nvalid; i += blockDim.x) {
        o[i] = sv[i];
        oa[i] = sa[i];
      }
    }
  };
  // region + frame: one scan of the roi's own bins feeds both (frame skips the cells strictly inside the shrunk roi).
  // share = 0 (a pooled tile too large for two LDS tiles): one tile, the two outputs in two scans
  if (!share) svalB = nullptr, sargB = nullptr;
  for (int ph = wave; ph < PH; ph += nwaves)
    for (int pw0 = 0; pw0 < PW; pw0 += PWT)
      loop_pool_row<T, PWT, CPL, true>(base, C, H, W, PH, PW, ph, pw0, min(PWT, PW - pw0), q.sw, q.sh, q.ew, q.eh, q.swi,
                                       q.shi, q.ewi, q.ehi, lane, svalA, sargA, svalB, sargB);
  __syncthreads();
  flush(svalA, sargA, 0);
  if (share) {
    flush(svalB, sargB, part);
  } else {
    __syncthreads();
    for (int ph = wave; ph < PH; ph += nwaves)
      for (int pw0 = 0; pw0 < PW; pw0 += PWT)
        loop_pool_row<T, PWT, CPL, false>(base, C, H, W, PH, PW, ph, pw0, min(PWT, PW - pw0), q.sw, q.sh, q.ew, q.eh, q.swi,
                                          q.shi, q.ewi, q.ehi, lane, nullptr, nullptr, svalA, sargA);
    __syncthreads();
    flush(svalA, sargA, part);
  }
  __syncthreads();
  // context: the bins of the grown roi without the cells strictly inside the roi itself
  for (int ph = wave; ph < PH; ph += nwaves)
    for (int pw0 = 0; pw0 < PW; pw0 += PWT)
      loop_pool_row<T, PWT, CPL, false>(base, C, H, W, PH, PW, ph, pw0, min(PWT, PW - pw0), q.swo, q.sho, q.ewo, q.eho,
                                        q.sw, q.sh, q.ew, q.eh, lane, nullptr, nullptr, svalA, sargA);
  __syncthreads();
  flush(svalA, sargA, 2 * part);
}

// RoIPool backward: scatter-add through argmax (ROILoopPool_cpu.cpp:82-123).
__global__ __launch_bounds__(256) void roi_pool_bwd(const float* __restrict__ grad_out, const float* __restrict__ rois,
                                                    const float* __restrict__ roi_scale,
                                                    const int* __restrict__ argmax, long long total, int C, int H,
                                                    int W, int PH, int PW, int nhwc, float* grad_in) {
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int am = argmax[idx];
    if (am < 0 || am >= H * W) continue;  // -1 = empty bin; anything else outside the map is never scattered
    const int c = (int)((idx / ((long long)PW * PH)) % C);
    const int r = (int)(idx / ((long long)PW * PH * C));
    const int n = (int)rois[(long long)r * 5];
    float g = grad_out[idx];
    if (roi_scale) g *= roi_scale[r];
    const long long off = nhwc ? ((long long)n * H * W + am) * C + c : ((long long)n * C + c) * H * W + am;
    atomicAdd(grad_in + off, g);
  }
}

// ---------------------------------------------------------------------------------
// ROIAlign (torchvision roi_align semantics; detectron2 ROIAlign(aligned=...))
// ---------------------------------------------------------------------------------
struct AlignBox {
  int batch;
  float start_w, start_h, bin_h, bin_w;
  int grid_h, grid_w;
  float inv_count;
};
__device__ __forceinline__ AlignBox decode_align(const float* roi, float spatial_scale, int PH, int PW,
                                                 int sampling_ratio, int aligned) {
  AlignBox a;
  a.batch = (int)roi[0];
  const float offset = aligned ? 0.5f : 0.0f;
  a.start_w = roi[1] * spatial_scale - offset;
  a.start_h = roi[2] * spatial_scale - offset;
  const float end_w = roi[3] * spatial_scale - offset;
  const float end_h = roi[4] * spatial_scale - offset;
  float rw = end_w - a.start_w, rh = end_h - a.start_h;
  if (!aligned) {
    rw = fmaxf(rw, 1.f);
    rh = fmaxf(rh, 1.f);
  }
  a.bin_h = rh / (float)PH;
  a.bin_w = rw / (float)PW;
  a.grid_h = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(rh / (float)PH);
  a.grid_w = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(rw / (float)PW);
  a.inv_count = 1.0f / (float)max(a.grid_h * a.grid_w, 1);
  return a;
}
struct Bilinear {
  int yl, yh, xl, xh;
  float w1, w2, w3, w4;
  bool valid;
};
__device__ __forceinline__ Bilinear bilinear_setup(float y, float x, int H, int W) {
  Bilinear s;
  s.valid = !(y < -1.0f || y > (float)H || x < -1.0f || x > (float)W);
  if (y <= 0.f) y = 0.f;
  if (x <= 0.f) x = 0.f;
  s.yl = (int)y;
  s.xl = (int)x;
  if (s.yl >= H - 1) {
    s.yh = s.yl = H - 1;
    y = (float)s.yl;
  } else {
    s.yh = s.yl + 1;
  }
  if (s.xl >= W - 1) {
    s.xh = s.xl = W - 1;
    x = (float)s.xl;
  } else {
    s.xh = s.xl + 1;
  }
  const float ly = y - (float)s.yl, lx = x - (float)s.xl;
  const float hy = 1.f - ly, hx = 1.f - lx;
  s.w1 = hy * hx;
  s.w2 = hy * lx;
  s.w3 = ly * hx;
  s.w4 = ly * lx;
  return s;
}

template <typename T>
__global__ __launch_bounds__(256) void roi_align_fwd_nhwc(const T* __restrict__ feat, const float* __restrict__ rois,
                                                          const float* __restrict__ roi_scale, int R, int C, int H,
                                                          int W, int PH, int PW, float spatial_scale,
                                                          int sampling_ratio, int aligned, void* out, int out_dtype,
                                                          int cgroups) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nbins = PH * PW;
  float* sval = (float*)smem + wave * 64 * nbins;
  const long long item = (long long)blockIdx.x * 4 + wave;
  if (item >= (long long)R * cgroups) return;
  const int r = (int)(item / cgroups);
  const int c0 = (int)(item % cgroups) * 64;
  const int c = c0 + lane;
  const bool c_ok = c < C;
  const AlignBox a = decode_align(rois + (long long)r * 5, spatial_scale, PH, PW, sampling_ratio, aligned);
  const float scale = roi_scale ? roi_scale[r] : 1.0f;
  const T* base = feat + (long long)a.batch * H * W * C + (c_ok ? c : 0);
  for (int ph = 0; ph < PH; ++ph)
    for (int pw = 0; pw < PW; ++pw) {
      float acc = 0.f;
      for (int iy = 0; iy < a.grid_h; ++iy) {
        const float y = a.start_h + (float)ph * a.bin_h + ((float)iy + .5f) * a.bin_h / (float)a.grid_h;
        for (int ix = 0; ix < a.grid_w; ++ix) {
          const float x = a.start_w + (float)pw * a.bin_w + ((float)ix + .5f) * a.bin_w / (float)a.grid_w;
          const Bilinear s = bilinear_setup(y, x, H, W);
          if (!s.valid) continue;
          const float v1 = to_f32(base[((long long)s.yl * W + s.xl) * C]);
          const float v2 = to_f32(base[((long long)s.yl * W + s.xh) * C]);
          const float v3 = to_f32(base[((long long)s.yh * W + s.xl) * C]);
          const float v4 = to_f32(base[((long long)s.yh * W + s.xh) * C]);
          acc += s.w1 * v1 + s.w2 * v2 + s.w3 * v3 + s.w4 * v4;
        }
      }
      acc *= a.inv_count;
      sval[lane * nbins + ph * PW + pw] = roi_scale ? acc * scale : acc;
    }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  const int nvalid = min(64, C - c0) * nbins;
  const long long obase = ((long long)r * C + c0) * nbins;
  if (out_dtype == WSOVOD_F32) {
    float* o = (float*)out + obase;
    for (int i = lane; i < nvalid; i += 64) o[i] = sval[i];
  } else {
    bf16_t* o = (bf16_t*)out + obase;
    for (int i = lane; i < nvalid; i += 64) o[i] = (bf16_t)sval[i];
  }
}

// ---------------------------------------------------------------------------------
// ROIAlign forward, NHWC, pooled width PWT (7 on every shipped config): one WORKGROUP per (roi, 64*CPL channels),
// one wavefront per pooled row ph, CPL adjacent channels per lane (one 8-byte load).  The PWT bins of the row advance
// through their sample grids together: every inner iteration issues 4*PWT independent coalesced loads (the four
// bilinear taps of one sample of each bin) before any arithmetic, instead of the dependent one-sample-at-a-time chain
// of the generic kernel.  Per-sample arithmetic and the accumulation order (iy outer, ix inner) are those of the
// generic kernel / torchvision's roi_align, so both produce the same bits.
// ---------------------------------------------------------------------------------
struct AlignAxis {
  int lo, hi;
  float l, h;
  bool valid;
};
__device__ __forceinline__ AlignAxis align_axis(float v, int L) {
  AlignAxis s;
  s.valid = !(v < -1.0f || v > (float)L);
  if (v <= 0.f) v = 0.f;
  s.lo = (int)v;
  if (s.lo >= L - 1) {
    s.hi = s.lo = L - 1;
    v = (float)s.lo;
  } else {
    s.hi = s.lo + 1;
  }
  s.l = v - (float)s.lo;
  s.h = 1.f - s.l;
  return s;
}

constexpr int kAlignMaxGrid = 64;  // sample columns per bin the row kernel keeps in its LDS table (7 KB); wider rois recompute
constexpr int kSepMax = 24;        // cells per bin and axis of the separable form (rois up to ~150 cells wide / high)
constexpr int kAlignTabBytes = 7 * kAlignMaxGrid * 16 + (7 + 16) * kSepMax * 4 + (2 * 7 + 2 * 16 + 1) * 4 + 12;

// U = columns of every bin per loop iteration; WPE = wavefronts per SIMD the register allocation has to leave room for.
template <typename T, int PWT, int CPL, bool OBF = false, int U = 1, int WPE = 4>
__global__ __launch_bounds__(512, WPE) void roi_align_fwd_nhwc_rows(const T* __restrict__ feat, const float* __restrict__ rois,
                                                               const float* __restrict__ roi_scale, int C, int H, int W,
                                                               int PH, float spatial_scale, int sampling_ratio,
                                                               int aligned, void* out, int out_dtype, int cgroups,
                                                               void* out_hi = nullptr) {
  typedef T vecc __attribute__((ext_vector_type(CPL)));
  constexpr int CG = 64 * CPL;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, ph = threadIdx.x >> 6;
  const int nbins = PH * PWT;
  // output transpose tile [channel][bin]: fp32, or (OBF: bf16 output) already rounded -- half the LDS, so that the
  // 512-channel form of the kernel (16-byte loads) still fits two to three workgroups per CU
  typedef typename std::conditional<OBF, bf16_t, float>::type sval_t;
  sval_t* sval = (sval_t*)smem;
  // The sample columns are the same for every lane (lane = channels), every pooled row and every sample row of the
  // roi: their bilinear set-up (x position with its fp32 division, clamping, weights) is computed ONCE per workgroup
  // by PWT*grid_w lanes into this table and read back as LDS broadcasts, instead of ~20 VALU instructions per bin and
  // loop iteration on all 64 lanes (the kernel was VALU-bound on exactly that).  Record = {lo*C, hi*C, l, h}; lo < 0
  // marks a sample outside the map.
  int4* xtab = (int4*)(smem + (((size_t)CG * nbins * sizeof(sval_t) + 15) & ~(size_t)15));
  const int r = blockIdx.x / cgroups;
  const int c0 = (blockIdx.x - r * cgroups) * CG;
  const int c = c0 + lane * CPL;
  const AlignBox a = decode_align(rois + (long long)r * 5, spatial_scale, PH, PWT, sampling_ratio, aligned);
  const float scale = roi_scale ? roi_scale[r] : 1.0f;
  auto x_record = [&](int pw, int ix) {
    const float x = a.start_w + (float)pw * a.bin_w + ((float)ix + .5f) * a.bin_w / (float)a.grid_w;
    const AlignAxis ax = align_axis(x, W);
    return int4{ax.valid ? ax.lo * C : -1, ax.hi * C, __float_as_int(ax.l), __float_as_int(ax.h)};
  };
  const bool tabled = a.grid_w <= kAlignMaxGrid;  // uniform over the workgroup
  if (tabled)
    for (int e = threadIdx.x; e < PWT * a.grid_w; e += blockDim.x) xtab[e] = x_record(e / a.grid_w, e % a.grid_w);
  __syncthreads();
  const T* base = feat + (long long)a.batch * H * W * C + (c < C ? c : 0);
  float acc[PWT][CPL];
#pragma unroll
  for (int pw = 0; pw < PWT; ++pw)
#pragma unroll
    for (int q = 0; q < CPL; ++q) acc[pw][q] = 0.f;
  // ---- separable form.  sum over the samples of a bin of (w1 v1 + w2 v2 + w3 v3 + w4 v4) = sum over the CELLS the bin
  // touches of Wy[row] * Wx[col] * f[row][col], with Wy[row] = (sum of hy over sample rows whose low row it is) + (sum of
  // ly over those whose high row it is), Wx likewise: every cell is loaded ONCE per bin instead of once per tap of every
  // sample that touches it (4 taps x ~1 sample per cell: a quarter of the L2 requests, which bound this kernel).  Same
  // value up to fp32 association.  Wx of the 7 bins is built by wavefront pw, Wy of pooled row ph by wavefront ph; rois
  // whose bins span more than kSepMax cells per axis keep the per-sample loop below.
  float* wx = (float*)(xtab + PWT * kAlignMaxGrid);  // [PWT][kSepMax] column weights
  float* wy = wx + PWT * kSepMax;                    // [PH][kSepMax] row weights
  int* meta = (int*)(wy + 16 * kSepMax);             // [0..PWT): first column, [PWT..2PWT): columns; then rows likewise
  int* fits = meta + 2 * PWT + 2 * 16;
  {
    if (threadIdx.x == 0) *fits = 1;
    __syncthreads();
    if (lane == 0) {
      // row weights of this wavefront's pooled row
      int first = -1, last = -1;
      for (int iy = 0; iy < a.grid_h; ++iy) {
        const float y = a.start_h + (float)ph * a.bin_h + ((float)iy + .5f) * a.bin_h / (float)a.grid_h;
        const AlignAxis ay = align_axis(y, H);
        if (!ay.valid) continue;
        first = first < 0 ? ay.lo : min(first, ay.lo);  // (a malformed roi, end < start, walks its samples backwards)
        last = max(last, ay.hi);
      }
      const int n = first < 0 ? 0 : last - first + 1;
      meta[2 * PWT + ph] = first;
      meta[2 * PWT + 16 + ph] = n;
      if (n > kSepMax) *fits = 0;
      else {
        for (int i = 0; i < n; ++i) wy[ph * kSepMax + i] = 0.f;
        for (int iy = 0; iy < a.grid_h; ++iy) {
          const float y = a.start_h + (float)ph * a.bin_h + ((float)iy + .5f) * a.bin_h / (float)a.grid_h;
          const AlignAxis ay = align_axis(y, H);
          if (!ay.valid) continue;
          wy[ph * kSepMax + ay.lo - first] += ay.h;
          wy[ph * kSepMax + ay.hi - first] += ay.l;
        }
      }
      if (ph < PWT) {  // column weights of bin pw = ph (PH >= PWT wavefronts on every shipped config; checked by the host)
        const int pw = ph;
        int f2 = -1, l2 = -1;
        for (int ix = 0; ix < a.grid_w; ++ix) {
          const float x = a.start_w + (float)pw * a.bin_w + ((float)ix + .5f) * a.bin_w / (float)a.grid_w;
          const AlignAxis ax = align_axis(x, W);
          if (!ax.valid) continue;
          f2 = f2 < 0 ? ax.lo : min(f2, ax.lo);
          l2 = max(l2, ax.hi);
        }
        const int n2 = f2 < 0 ? 0 : l2 - f2 + 1;
        meta[pw] = f2;
        meta[PWT + pw] = n2;
        if (n2 > kSepMax) *fits = 0;
        else {
          for (int i = 0; i < n2; ++i) wx[pw * kSepMax + i] = 0.f;
          for (int ix = 0; ix < a.grid_w; ++ix) {
            const float x = a.start_w + (float)pw * a.bin_w + ((float)ix + .5f) * a.bin_w / (float)a.grid_w;
            const AlignAxis ax = align_axis(x, W);
            if (!ax.valid) continue;
            wx[pw * kSepMax + ax.lo - f2] += ax.h;
            wx[pw * kSepMax + ax.hi - f2] += ax.l;
          }
        }
      }
    }
    __syncthreads();
    if (*fits) {
      int x0[PWT], nc[PWT], ncmax = 0;
#pragma unroll
      for (int pw = 0; pw < PWT; ++pw) {  // (the same for every lane: kept in scalar registers)
        x0[pw] = __builtin_amdgcn_readfirstlane(max(meta[pw], 0));
        nc[pw] = __builtin_amdgcn_readfirstlane(meta[PWT + pw]);
        ncmax = max(ncmax, nc[pw]);
      }
      const int y0 = __builtin_amdgcn_readfirstlane(meta[2 * PWT + ph]), nr = __builtin_amdgcn_readfirstlane(meta[2 * PWT + 16 + ph]);
      for (int ri = 0; ri < nr; ++ri) {
        const float wyv = wy[ph * kSepMax + ri];
        const T* row = base + (long long)(y0 + ri) * W * C;
        for (int ci = 0; ci < ncmax; ci += U) {  // U columns of every bin per iteration: 7 U loads in flight per wavefront
          vecc v[U][PWT];
          float w[U][PWT];
#pragma unroll
          for (int u = 0; u < U; ++u)
#pragma unroll
            for (int pw = 0; pw < PWT; ++pw) {
              const bool in = ci + u < nc[pw];
              v[u][pw] = *(const vecc*)(row + (long long)min(x0[pw] + ci + u, W - 1) * C);  // past the bin: weight 0
              w[u][pw] = in ? wyv * wx[pw * kSepMax + min(ci + u, kSepMax - 1)] : 0.f;
            }
#pragma unroll
          for (int u = 0; u < U; ++u)
#pragma unroll
            for (int pw = 0; pw < PWT; ++pw)
#pragma unroll
              for (int q = 0; q < CPL; ++q) acc[pw][q] += w[u][pw] * to_f32(v[u][pw][q]);
        }
      }
    }
  }
  const bool separable_done = *fits != 0;  // uniform over the workgroup
  for (int iy = 0; iy < (separable_done ? 0 : a.grid_h); ++iy) {
    const float y = a.start_h + (float)ph * a.bin_h + ((float)iy + .5f) * a.bin_h / (float)a.grid_h;
    const AlignAxis ay = align_axis(y, H);
    if (!ay.valid) continue;  // uniform over the wavefront (one roi, one pooled row)
    const T* r0 = base + (long long)ay.lo * W * C;
    const T* r1 = base + (long long)ay.hi * W * C;
    // (rois wider than kSepMax cells per bin: rare, so one bin at a time -- four loads in flight -- and the register
    // count of the kernel is set by the separable loop above)
    for (int ix = 0; ix < a.grid_w; ++ix) {
#pragma unroll
      for (int pw = 0; pw < PWT; ++pw) {
        const int4 rec = tabled ? xtab[pw * a.grid_w + ix] : x_record(pw, ix);
        if (rec.x < 0) continue;  // (uniform over the wavefront)
        const vecc v1 = *(const vecc*)(r0 + rec.x), v2 = *(const vecc*)(r0 + rec.y);
        const vecc v3 = *(const vecc*)(r1 + rec.x), v4 = *(const vecc*)(r1 + rec.y);
        const float xl = __int_as_float(rec.z), xh = __int_as_float(rec.w);
        const float w1 = ay.h * xh, w2 = ay.h * xl, w3 = ay.l * xh, w4 = ay.l * xl;
#pragma unroll
        for (int q = 0; q < CPL; ++q)
          acc[pw][q] += w1 * to_f32(v1[q]) + w2 * to_f32(v2[q]) + w3 * to_f32(v3[q]) + w4 * to_f32(v4[q]);
      }
    }
  }
#pragma unroll
  for (int pw = 0; pw < PWT; ++pw)
#pragma unroll
    for (int q = 0; q < CPL; ++q) {
      const float v = acc[pw][q] * a.inv_count;
      sval[(lane * CPL + q) * nbins + ph * PWT + pw] = (sval_t)(roi_scale ? v * scale : v);
    }
  __syncthreads();
  const int nthreads = blockDim.x, tid = threadIdx.x;
  const int nvalid = min(CG, C - c0) * nbins;
  const long long obase = ((long long)r * C + c0) * nbins;
  const bool vec = (nvalid & 3) == 0 && (obase & 3) == 0;
  if constexpr (OBF) {  // (launcher: out_dtype is bf16) the tile already holds the output bits
    bf16_t* o = (bf16_t*)out + obase;
    if ((nvalid & 7) == 0 && (obase & 7) == 0)
      for (int i = tid * 8; i < nvalid; i += nthreads * 8)
        __builtin_nontemporal_store(*(const bf16x8*)(sval + i), (bf16x8*)(o + i));
    else
      for (int i = tid; i < nvalid; i += nthreads) o[i] = sval[i];
    return;
  } else if (out_dtype == WSOVOD_F32) {
    float* o = (float*)out + obase;
    if (vec)
      for (int i = tid * 4; i < nvalid; i += nthreads * 4) *(float4*)(o + i) = *(const float4*)(sval + i);
    else
      for (int i = tid; i < nvalid; i += nthreads) o[i] = sval[i];
  } else if (out_dtype == WSOVOD_F16MX) {
    // unit-scale f16mx (include/wsovod_hip.h, round 6): 8 values per lane = 16 B of fp16 hi + 8 B of e4m3 q + 8 B of e4m3 ql of
    // one 128-byte group, and -- `out_hi` -- the plain bf16 rounding, the operand of the first FC layer's weight gradient
    char* o = (char*)out;
    for (int i = tid * 8; i < nvalid; i += nthreads * 8) {
      wsovod_mx::f16x4 h0, h1;
      int q0, q1, l0, l1;
      const f32x4 v0 = *(const f32x4*)(sval + i), v1 = *(const f32x4*)(sval + i + 4);
      wsovod_mx::mx_enc4_unit(v0, h0, q0, l0);
      wsovod_mx::mx_enc4_unit(v1, h1, q1, l1);
      const long long k = obase + i;
      char* d = o + wsovod_mx::mx_group(k);
      const int w = (int)(k & 31);
      __builtin_nontemporal_store(wsovod_mx::f16x8{h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]}, (wsovod_mx::f16x8*)(d + 2 * w));
      __builtin_nontemporal_store(wsovod_mx::i32x2{q0, q1}, (wsovod_mx::i32x2*)(d + 64 + w));
      __builtin_nontemporal_store(wsovod_mx::i32x2{l0, l1}, (wsovod_mx::i32x2*)(d + 96 + w));
      if (out_hi)
        __builtin_nontemporal_store(bf16x8{(bf16_t)v0[0], (bf16_t)v0[1], (bf16_t)v0[2], (bf16_t)v0[3], (bf16_t)v1[0], (bf16_t)v1[1],
                                           (bf16_t)v1[2], (bf16_t)v1[3]}, (bf16x8*)((bf16_t*)out_hi + k));
    }
  } else if (out_dtype == WSOVOD_BF16X2 || out_dtype == WSOVOD_BF16X2P) {
    // bf16x2 (include/wsovod_hip.h): the run [obase, obase + nvalid) covers whole 32-value groups (launcher); 8 values per
    // lane = 16 B of hi and 16 B of lo half a line further, streamed past L2 like the bf16 form.  PLANAR (round 5): hi to the
    // first bf16 matrix of the output, lo to the second (`out_hi` = its base, set by the launcher): the hi plane is at once
    // the plain bf16 operand of the first FC layer's weight gradient -- no third store per value
    bf16_t* o = (bf16_t*)out;
    const bool planar = out_dtype == WSOVOD_BF16X2P;
    for (int i = tid * 8; i < nvalid; i += nthreads * 8) {
      const f32x4 q0 = *(const f32x4*)(sval + i), q1 = *(const f32x4*)(sval + i + 4);
      bf16x8 hi, lo;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        hi[e] = (bf16_t)q0[e];
        hi[4 + e] = (bf16_t)q1[e];
        const float h0 = (float)hi[e], h1 = (float)hi[4 + e];
        lo[e] = (bf16_t)(__builtin_isinf(h0) ? 0.f : q0[e] - h0);
        lo[4 + e] = (bf16_t)(__builtin_isinf(h1) ? 0.f : q1[e] - h1);
      }
      const long long k = obase + i;
      if (planar) {
        __builtin_nontemporal_store(hi, (bf16x8*)(o + k));
        __builtin_nontemporal_store(lo, (bf16x8*)((bf16_t*)out_hi + k));
        continue;
      }
      bf16_t* d = o + ((k >> 5) << 6) + (k & 31);
      __builtin_nontemporal_store(hi, (bf16x8*)d);
      __builtin_nontemporal_store(lo, (bf16x8*)(d + 32));
      if (out_hi) __builtin_nontemporal_store(hi, (bf16x8*)((bf16_t*)out_hi + k));  // plain bf16 copy (see the launcher)
    }
  } else {
    bf16_t* o = (bf16_t*)out + obase;
    if (vec && (nvalid & 7) == 0 && (obase & 7) == 0)
      for (int i = tid * 8; i < nvalid; i += nthreads * 8) {
        const f32x4 q0 = *(const f32x4*)(sval + i), q1 = *(const f32x4*)(sval + i + 4);
        const bf16x8 pk = {(bf16_t)q0[0], (bf16_t)q0[1], (bf16_t)q0[2], (bf16_t)q0[3],
                           (bf16_t)q1[0], (bf16_t)q1[1], (bf16_t)q1[2], (bf16_t)q1[3]};
        __builtin_nontemporal_store(pk, (bf16x8*)(o + i));  // as the RoIPool kernel: streamed once, keep the map in L2
      }
    else if (vec)
      for (int i = tid * 4; i < nvalid; i += nthreads * 4) {
        const float4 q = *(const float4*)(sval + i);
        bf16x4 pk = {(bf16_t)q.x, (bf16_t)q.y, (bf16_t)q.z, (bf16_t)q.w};
        *(bf16x4*)(o + i) = pk;
      }
    else
      for (int i = tid; i < nvalid; i += nthreads) o[i] = (bf16_t)sval[i];
  }
}

template <typename T>
__global__ __launch_bounds__(256) void roi_align_fwd_nchw(const T* __restrict__ feat, const float* __restrict__ rois,
                                                          const float* __restrict__ roi_scale, long long total, int C,
                                                          int H, int W, int PH, int PW, float spatial_scale,
                                                          int sampling_ratio, int aligned, void* out, int out_dtype) {
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int pw = (int)(idx % PW);
    const int ph = (int)((idx / PW) % PH);
    const int c = (int)((idx / ((long long)PW * PH)) % C);
    const int r = (int)(idx / ((long long)PW * PH * C));
    const AlignBox a = decode_align(rois + (long long)r * 5, spatial_scale, PH, PW, sampling_ratio, aligned);
    const T* plane = feat + ((long long)a.batch * C + c) * H * W;
    float acc = 0.f;
    for (int iy = 0; iy < a.grid_h; ++iy) {
      const float y = a.start_h + (float)ph * a.bin_h + ((float)iy + .5f) * a.bin_h / (float)a.grid_h;
      for (int ix = 0; ix < a.grid_w; ++ix) {
        const float x = a.start_w + (float)pw * a.bin_w + ((float)ix + .5f) * a.bin_w / (float)a.grid_w;
        const Bilinear s = bilinear_setup(y, x, H, W);
        if (!s.valid) continue;
        acc += s.w1 * to_f32(plane[s.yl * W + s.xl]) + s.w2 * to_f32(plane[s.yl * W + s.xh]) +
               s.w3 * to_f32(plane[s.yh * W + s.xl]) + s.w4 * to_f32(plane[s.yh * W + s.xh]);
      }
    }
    acc *= a.inv_count;
    if (roi_scale) acc *= roi_scale[r];
    if (out_dtype == WSOVOD_F32)
      ((float*)out)[idx] = acc;
    else
      ((bf16_t*)out)[idx] = (bf16_t)acc;
  }
}

__global__ __launch_bounds__(256) void roi_align_bwd(const float* __restrict__ grad_out,
                                                     const float* __restrict__ rois,
                                                     const float* __restrict__ roi_scale, long long total, int C,
                                                     int H, int W, int PH, int PW, float spatial_scale,
                                                     int sampling_ratio, int aligned, int nhwc, float* grad_in) {
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int pw = (int)(idx % PW);
    const int ph = (int)((idx / PW) % PH);
    const int c = (int)((idx / ((long long)PW * PH)) % C);
    const int r = (int)(idx / ((long long)PW * PH * C));
    const AlignBox a = decode_align(rois + (long long)r * 5, spatial_scale, PH, PW, sampling_ratio, aligned);
    float g = grad_out[idx] * a.inv_count;
    if (roi_scale) g *= roi_scale[r];
    for (int iy = 0; iy < a.grid_h; ++iy) {
      const float y = a.start_h + (float)ph * a.bin_h + ((float)iy + .5f) * a.bin_h / (float)a.grid_h;
      for (int ix = 0; ix < a.grid_w; ++ix) {
        const float x = a.start_w + (float)pw * a.bin_w + ((float)ix + .5f) * a.bin_w / (float)a.grid_w;
        const Bilinear s = bilinear_setup(y, x, H, W);
        if (!s.valid) continue;
        auto off = [&](int yy, int xx) -> long long {
          return nhwc ? (((long long)a.batch * H + yy) * W + xx) * C + c
                      : (((long long)a.batch * C + c) * H + yy) * W + xx;
        };
        atomicAdd(grad_in + off(s.yl, s.xl), g * s.w1);
        atomicAdd(grad_in + off(s.yl, s.xh), g * s.w2);
        atomicAdd(grad_in + off(s.yh, s.xl), g * s.w3);
        atomicAdd(grad_in + off(s.yh, s.xh), g * s.w4);
      }
    }
  }
}

// convert_boxes_to_pooler_format (poolers.py:74-108): (M,4) boxes of all images, concatenated, -> (M,5)
// [image index as float, x0, y0, x1, y1]; the image of row r is found in the prefix offsets.  Optionally also the
// objectness scale of roi_heads.py:733-739, objectness + 1.  One launch instead of a full_like + cat per image.
__global__ __launch_bounds__(256) void format_rois_kernel(const float* __restrict__ boxes, const int* __restrict__ seg,
                                                          int G, int M, const float* __restrict__ objectness,
                                                          float* __restrict__ rois, float* __restrict__ scale) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= M) return;
  int lo = 0, hi = G;  // largest g in [0, G) with seg[g] <= r (empty images share an offset with their successor)
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (seg[mid] <= r) lo = mid; else hi = mid;
  }
  const float4 b = ((const float4*)boxes)[r];
  float* o = rois + (long long)r * 5;
  o[0] = (float)lo;
  o[1] = b.x;
  o[2] = b.y;
  o[3] = b.z;
  o[4] = b.w;
  if (scale) scale[r] = objectness[r] + 1.0f;
}

int check_common(const char* fn, const void* feat, int dtype, int layout, const float* rois, int R, int N, int C,
                 int H, int W, int ph, int pw, const void* out) {
  WS_CHECK_ARG(dtype == WSOVOD_F32 || dtype == WSOVOD_BF16, "%s: bad dtype %d", fn, dtype);
  WS_CHECK_ARG(layout == WSOVOD_NCHW || layout == WSOVOD_NHWC, "%s: bad layout %d", fn, layout);
  WS_CHECK_ARG(R >= 0 && N >= 0 && C > 0 && H > 0 && W > 0 && ph > 0 && pw > 0, "%s: bad shape", fn);
  WS_CHECK_ARG((long long)H * W < (1ll << 31), "%s: H*W overflows int32 argmax", fn);
  if (R > 0) WS_CHECK_ARG(feat && rois && out, "%s: null pointer", fn);
  return WSOVOD_OK;
}

}  // namespace

extern "C" {

int wsovod_format_rois(const float* boxes, const int* seg_offsets, int G, int M, const float* objectness, float* rois,
                       float* roi_scale, wsovod_stream_t stream) {
  WS_CHECK_ARG(G >= 0 && M >= 0, "wsovod_format_rois: bad shape");
  if (M == 0) return WSOVOD_OK;
  WS_CHECK_ARG(G > 0 && boxes && seg_offsets && rois, "wsovod_format_rois: null pointer");
  WS_CHECK_ARG(((uintptr_t)boxes & 15) == 0, "wsovod_format_rois: boxes must be 16-byte aligned");
  WS_CHECK_ARG(!roi_scale == !objectness, "wsovod_format_rois: objectness and roi_scale go together");
  static int slot = wsovod::prof_slot("format_rois");
  hipStream_t s = (hipStream_t)stream;
  wsovod::ProfScope prof(slot, s, 0.0, (double)M * (36.0 + (roi_scale ? 8.0 : 0.0)));
  hipLaunchKernelGGL(format_rois_kernel, dim3(ceil_div(M, 256)), dim3(256), 0, s, boxes, seg_offsets, G, M, objectness,
                     rois, roi_scale);
  WS_CHECK_LAUNCH("wsovod_format_rois");
  return WSOVOD_OK;
}

int wsovod_roi_pool_forward(const void* feat, int dtype, int layout, const float* rois, const float* roi_scale, int R,
                            int N, int C, int H, int W, int ph, int pw, float spatial_scale, void* out,
                            int out_dtype, int* argmax, wsovod_stream_t stream) {
  return wsovod_roi_pool_forward_x2hi(feat, dtype, layout, rois, roi_scale, R, N, C, H, W, ph, pw, spatial_scale, out,
                                      out_dtype, argmax, nullptr, stream);
}

int wsovod_roi_pool_forward_x2hi(const void* feat, int dtype, int layout, const float* rois, const float* roi_scale, int R,
                                 int N, int C, int H, int W, int ph, int pw, float spatial_scale, void* out,
                                 int out_dtype, int* argmax, void* out_hi, wsovod_stream_t stream) {
  return wsovod_roi_pool_forward_ws(feat, dtype, layout, rois, roi_scale, R, N, C, H, W, ph, pw, spatial_scale, out, out_dtype,
                                    argmax, out_hi, nullptr, 0, stream);
}

long long wsovod_roi_pool_workspace_bytes(int dtype, int layout, int R, int N, int C, int H, int W, int ph, int pw,
                                          int want_argmax) {
  // the 2x2-max map pays when the rois re-read the map many times over (512 boxes on 75 x 100 cells: ~30x); a handful of
  // boxes keeps the direct scan.  Values only: argmax needs the first maximum in scan order, i.e. the cells themselves.
  const int v = dtype == WSOVOD_BF16 ? 8 : 4;
  if (want_argmax || layout != WSOVOD_NHWC || pw != 7 || ph > 16 || (dtype != WSOVOD_BF16 && dtype != WSOVOD_F32)) return 0;
  if (C % v != 0 || H < 2 || W < 2 || N <= 0 || (long long)R * 49 < (long long)N * H * W / 2) return 0;
  const char* e = getenv("WSOVOD_ROIPOOL_M2");
  if (e && atoi(e) == 0) return 0;
  return (long long)N * H * W * C * (dtype == WSOVOD_BF16 ? 2 : 4);
}

static int roi_pool_forward_impl(const void* feat, int dtype, int layout, const float* rois, const float* roi_scale, int R,
                                 int N, int C, int H, int W, int ph, int pw, float spatial_scale, void* out, int out_dtype,
                                 int* argmax, void* out_hi, void* workspace, long long workspace_bytes, bool m2_ready,
                                 wsovod_stream_t stream);

int wsovod_roi_pool_forward_ws(const void* feat, int dtype, int layout, const float* rois, const float* roi_scale, int R,
                               int N, int C, int H, int W, int ph, int pw, float spatial_scale, void* out, int out_dtype,
                               int* argmax, void* out_hi, void* workspace, long long workspace_bytes,
                               wsovod_stream_t stream) {
  return roi_pool_forward_impl(feat, dtype, layout, rois, roi_scale, R, N, C, H, W, ph, pw, spatial_scale, out, out_dtype,
                               argmax, out_hi, workspace, workspace_bytes, false, stream);
}

int wsovod_roi_pool_forward_m2(const void* feat, int dtype, int layout, const float* rois, const float* roi_scale, int R,
                               int N, int C, int H, int W, int ph, int pw, float spatial_scale, void* out, int out_dtype,
                               int* argmax, void* out_hi, const void* m2, long long m2_bytes, wsovod_stream_t stream) {
  WS_CHECK_ARG(m2 != nullptr, "wsovod_roi_pool_forward_m2: null map");
  return roi_pool_forward_impl(feat, dtype, layout, rois, roi_scale, R, N, C, H, W, ph, pw, spatial_scale, out, out_dtype,
                               argmax, out_hi, (void*)m2, m2_bytes, true, stream);
}

long long wsovod_max2x2_gap_workspace_floats(int dtype, int N, int C, int H, int W) {
  const int v = dtype == WSOVOD_BF16 ? 8 : 4;
  if ((dtype != WSOVOD_BF16 && dtype != WSOVOD_F32) || C % v != 0 || N <= 0 || H < 1 || W < 1) return 0;
  const int cv = C / v;
  int pb = 1;
  while (pb < 256 && 256 / (pb * 2) >= cv) pb *= 2;  // PB columns x CB = 256 / PB channel vectors, CB >= cv where possible
  return (long long)N * ceil_div(H, 8) * ceil_div(W, pb) * C;
}

int wsovod_max2x2_gap_nhwc(const void* feat, int dtype, int N, int C, int H, int W, void* m2_out, float* gap_out,
                           float* gap_workspace, wsovod_stream_t stream) {
  WS_CHECK_ARG(dtype == WSOVOD_BF16 || dtype == WSOVOD_F32, "wsovod_max2x2_gap_nhwc: bad dtype");
  if (N == 0) return WSOVOD_OK;
  const int v = dtype == WSOVOD_BF16 ? 8 : 4;
  WS_CHECK_ARG(feat && m2_out && gap_out && gap_workspace && C > 0 && C % v == 0 && H >= 1 && W >= 1 &&
                   (((uintptr_t)feat | (uintptr_t)m2_out) & 15) == 0,
               "wsovod_max2x2_gap_nhwc: bad argument (NHWC map, C a multiple of %d, 16-byte aligned)", v);
  WS_CHECK_ARG((long long)N * H * W * C * (dtype == WSOVOD_BF16 ? 2 : 4) < (1ll << 40), "wsovod_max2x2_gap_nhwc: map too large");
  static int slot = wsovod::prof_slot("roi_pool_max2x2_map_gap");
  hipStream_t s = (hipStream_t)stream;
  const double esz = dtype == WSOVOD_BF16 ? 2.0 : 4.0;
  wsovod::ProfScope prof(slot, s, 0.0, 2.0 * (double)N * C * H * W * esz);
  constexpr int RH = 8;
  const int cv = C / v;
  int pb = 1;
  while (pb < 256 && 256 / (pb * 2) >= cv) pb *= 2;
  const int wblocks = ceil_div(W, pb), strips = ceil_div(H, RH);
  const long long blocks = (long long)N * strips * wblocks;
  WS_CHECK_ARG(blocks < (1ll << 31), "wsovod_max2x2_gap_nhwc: too many workgroups");
  if (dtype == WSOVOD_BF16)
    hipLaunchKernelGGL((max2x2_s1_strip_gap_nhwc<bf16_t, 8, RH>), dim3((unsigned)blocks), dim3(256), 0, s, (const bf16_t*)feat,
                       (bf16_t*)m2_out, H, W, C, pb, wblocks, gap_workspace);
  else
    hipLaunchKernelGGL((max2x2_s1_strip_gap_nhwc<float, 4, RH>), dim3((unsigned)blocks), dim3(256), 0, s, (const float*)feat,
                       (float*)m2_out, H, W, C, pb, wblocks, gap_workspace);
  hipLaunchKernelGGL(max2x2_gap_finalize, dim3(N * ceil_div(C, 16)), dim3(256), 0, s, gap_workspace, strips * wblocks, C,
                     1.0f / (float)((long long)H * W), gap_out);
  WS_CHECK_LAUNCH("wsovod_max2x2_gap_nhwc");
  return WSOVOD_OK;
}

static int roi_pool_forward_impl(const void* feat, int dtype, int layout, const float* rois, const float* roi_scale, int R,
                                 int N, int C, int H, int W, int ph, int pw, float spatial_scale, void* out, int out_dtype,
                                 int* argmax, void* out_hi, void* workspace, long long workspace_bytes, bool m2_ready,
                                 wsovod_stream_t stream) {
  int rc = check_common("wsovod_roi_pool_forward", feat, dtype, layout, rois, R, N, C, H, W, ph, pw, out);
  WS_CHECK_ARG(!out_hi || ((out_dtype == WSOVOD_BF16X2 || out_dtype == WSOVOD_F16MX) && ((uintptr_t)out_hi & 15) == 0),
               "wsovod_roi_pool_forward_x2hi: the bf16 copy goes with a bf16x2 / f16mx output");
  if (rc) return rc;
  WS_CHECK_ARG(out_dtype == WSOVOD_F32 || out_dtype == WSOVOD_BF16 || out_dtype == WSOVOD_BF16X2 || out_dtype == WSOVOD_BF16X2P ||
                   out_dtype == WSOVOD_F16MX, "wsovod_roi_pool_forward: bad out_dtype");
  const bool planar_out = out_dtype == WSOVOD_BF16X2P;  // same constraints as the interleaved form; lo plane behind the hi plane
  if (planar_out) {
    WS_CHECK_ARG(((long long)R * C * ph * pw) % 8 == 0, "wsovod_roi_pool_forward: planar bf16x2 output needs 16-byte aligned planes");
    out_hi = (char*)out + (long long)R * C * ph * pw * 2;
  }
  // bf16x2 output: the wavefront-per-pooled-row kernel only (NHWC, 7 bins wide), whole 128-channel groups, so that every
  // workgroup's run of outputs is whole 32-value groups
  WS_CHECK_ARG((out_dtype != WSOVOD_BF16X2 && out_dtype != WSOVOD_F16MX && !planar_out) || (layout == WSOVOD_NHWC && pw == 7 && ph <= 16 && C % 256 == 0 &&
                                              (C * ph * pw) % 32 == 0 && ((uintptr_t)feat & 7) == 0 && ((uintptr_t)out & 15) == 0),
               "wsovod_roi_pool_forward: bf16x2 output needs NHWC, pw = 7, C a multiple of 256");
  if (R == 0) return WSOVOD_OK;
  hipStream_t s = (hipStream_t)stream;
  const int esz = dtype == WSOVOD_BF16 ? 2 : 4, osz = out_dtype == WSOVOD_BF16 ? 2 : 4;
  const double out_elems = (double)R * C * ph * pw;
  // algorithmic bytes: every feature element once + outputs (+argmax) + rois
  const double bytes = (double)N * C * H * W * esz + out_elems * (osz + (argmax ? 4 : 0)) + 20.0 * R;
  if (layout == WSOVOD_NHWC) {
    static int slot = wsovod::prof_slot("roi_pool_fwd_nhwc");
    const int cgroups = ceil_div(C, 64);
    const long long items = (long long)R * cgroups;
    const int grid = (int)ceil_div_ll(items, 4);
    const int lds = 4 * 64 * ph * pw * 4 * (argmax ? 2 : 1);
    wsovod::ProfScope prof(slot, s, 0.0, bytes);
    if (lds > 160 * 1024 && !(pw == 7 && ph <= 16 && (C & 1) == 0)) {
      // pooled tile does not fit the per-wavefront LDS transpose: thread-per-bin fallback
      const long long total = (long long)R * C * ph * pw;
      const int g = (int)std::min<long long>(ceil_div_ll(total, 256), 256 * 32);
      if (dtype == WSOVOD_BF16)
        hipLaunchKernelGGL(roi_pool_fwd_nchw<bf16_t>, dim3(g), dim3(256), 0, s, (const bf16_t*)feat, rois, roi_scale,
                           total, C, H, W, ph, pw, spatial_scale, out, out_dtype, argmax, 1);
      else
        hipLaunchKernelGGL(roi_pool_fwd_nchw<float>, dim3(g), dim3(256), 0, s, (const float*)feat, rois, roi_scale,
                           total, C, H, W, ph, pw, spatial_scale, out, out_dtype, argmax, 1);
      WS_CHECK_LAUNCH("wsovod_roi_pool_forward");
      return WSOVOD_OK;
    }
#define LAUNCH_POOL(T, AM)                                                                                        \
  do {                                                                                                            \
    auto k = roi_pool_fwd_nhwc<T, AM>;                                                                            \
    if (lds > 64 * 1024) WS_CHECK_HIP(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds), \
                                      "wsovod_roi_pool_forward: LDS opt-in");                                      \
    hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds, s, (const T*)feat, rois, roi_scale, R, C, H, W, ph, pw,      \
                       spatial_scale, out, out_dtype, argmax, cgroups);                                           \
  } while (0)
    if (pw == 7 && ph <= 16 && (C & 1) == 0 && (((uintptr_t)feat) & 7) == 0) {
      // fast path: workgroup per (roi, 64*CPL-channel group), wavefront per pooled row, one 8- or 16-byte load per lane
      // and cell; bf16 maps whose channel count is not a multiple of 4 keep 2 channels per lane
      const long long need = wsovod_roi_pool_workspace_bytes(dtype, layout, R, N, C, H, W, ph, pw, argmax != nullptr);
      const bool use_m2 = workspace && need > 0 && workspace_bytes >= need && (((uintptr_t)feat | (uintptr_t)workspace) & 15) == 0;
      if (use_m2 && !m2_ready) {  // one pass over the map: its stride-1 2x2 maxima (the rows kernel then reads a quarter of the cells)
        static int slot2 = wsovod::prof_slot("roi_pool_max2x2_map");
        const int v = dtype == WSOVOD_BF16 ? 8 : 4;
        const long long total_vec = (long long)N * H * W * (C / v);
        const int g2 = (int)std::min<long long>(ceil_div_ll(total_vec, 256), 256 * 64);
        wsovod::ProfScope prof2(slot2, s, 0.0, 2.0 * (double)N * C * H * W * esz);
        // strips of 8 rows per thread (WSOVOD_ROIPOOL_M2_STRIP=0: one output per thread; A/B runs)
        const char* ms = getenv("WSOVOD_ROIPOOL_M2_STRIP");
        constexpr int RH = 8;
        const long long total_strip = (long long)N * ceil_div(H, RH) * W * (C / v);
        const int g3 = (int)std::min<long long>(ceil_div_ll(total_strip, 256), 256 * 64);
        if (!(ms && ms[0] == '0') && dtype == WSOVOD_BF16)
          hipLaunchKernelGGL((max2x2_s1_strip_nhwc<bf16_t, 8, RH>), dim3(g3), dim3(256), 0, s, (const bf16_t*)feat,
                             (bf16_t*)workspace, H, W, C, total_strip);
        else if (!(ms && ms[0] == '0'))
          hipLaunchKernelGGL((max2x2_s1_strip_nhwc<float, 4, RH>), dim3(g3), dim3(256), 0, s, (const float*)feat,
                             (float*)workspace, H, W, C, total_strip);
        else if (dtype == WSOVOD_BF16)
          hipLaunchKernelGGL((max2x2_s1_nhwc<bf16_t, 8>), dim3(g2), dim3(256), 0, s, (const bf16_t*)feat, (bf16_t*)workspace,
                             H, W, C, total_vec);
        else
          hipLaunchKernelGGL((max2x2_s1_nhwc<float, 4>), dim3(g2), dim3(256), 0, s, (const float*)feat, (float*)workspace, H,
                             W, C, total_vec);
        WS_CHECK_LAUNCH("wsovod_roi_pool_forward (2x2-max map)");
      }
      const bool wide = dtype == WSOVOD_BF16 && (C & 3) == 0 && !argmax;  // measured: 4 channels per lane pay off only without argmax registers (C = 2048: 1.83 -> 1.66 ms; with argmax 0.70 -> 1.07)
      // fp32 maps (the "parity" precision's res5): 4 channels = one 16-byte load per lane and cell (WSOVOD_ROIPOOL_F32_WIDE=0: 2)
      // (WSOVOD_ROIPOOL_F32_CPL = 1 / 2 / 4 selects the channels per lane for A/B runs: 64 / 128 / 256 channels per workgroup)
      const char* wf = getenv("WSOVOD_ROIPOOL_F32_CPL");
      const int fcpl = wf ? atoi(wf) : (use_m2 ? 4 : 2);  // cell scan, 32 x 512 boxes, bf16x2 + bf16 out: 4 -> 1.665, 2 -> 1.530, 1 -> 1.673 ms
      const bool f32na = dtype == WSOVOD_F32 && !argmax;
      const bool widef = f32na && fcpl == 4 && (C & 3) == 0 && (((uintptr_t)feat) & 15) == 0;
      const bool narrowf = f32na && fcpl == 1;  // one XCD per 64-channel group: its slice of an fp32 map (1.9 MB) stays in L2
      // bf16 maps through the 2x2-max map: WSOVOD_ROIPOOL_BF16_CPL = 8 selects 8 channels = one 16-byte load per lane
      // (A/B runs: 0.560 vs 0.575 ms at 32 x 512 boxes, but 123 VGPRs against 68: the 4-channel form stays the default)
      const char* wb = getenv("WSOVOD_ROIPOOL_BF16_CPL");
      const bool wide8 = wide && use_m2 && out_dtype == WSOVOD_BF16 && (C & 7) == 0 && wb && atoi(wb) == 8;
      const int cg = wide8 ? 512 : (wide || widef) ? 256 : narrowf ? 64 : 128;
      const int cgroups = ceil_div(C, cg);
      const bool obf = wide && out_dtype == WSOVOD_BF16;  // bf16 transpose tile: 25 instead of 50 KiB per workgroup (0.833 -> 0.817 ms)
      const int lds7 = obf ? cg * ph * pw * 2 : cg * ph * pw * 4 * (argmax ? 2 : 1);
      // WSOVOD_ROIPOOL_XCD=1: channel group g on 8 / cgroups XCDs only (A/B switch; see the kernel)
      const char* wx = getenv("WSOVOD_ROIPOOL_XCD");
      const int xpg = (wx && atoi(wx) == 1 && (cgroups == 2 || cgroups == 4 || cgroups == 8)) ? 8 / cgroups : 0;
      const int grid7 = xpg ? ceil_div(R, xpg) * 8 : R * cgroups;
      const void* m2p = use_m2 ? workspace : nullptr;
      WS_CHECK_ARG(lds7 <= 160 * 1024, "wsovod_roi_pool_forward: pooled tile too large for LDS");
#define LAUNCH_ROWS(T, AM, CPL, OBF)                                                                               \
  do {                                                                                                             \
    auto k = roi_pool_fwd_nhwc_rows<T, AM, 7, CPL, OBF>;                                                           \
    if (lds7 > 64 * 1024) WS_CHECK_HIP(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds7), \
                                       "wsovod_roi_pool_forward: LDS opt-in");                                     \
    hipLaunchKernelGGL(k, dim3(grid7), dim3(64 * ph), lds7, s, (const T*)feat, rois, roi_scale, C, H, W, ph,        \
                       spatial_scale, out, out_dtype, argmax, cgroups, out_hi, (const T*)m2p, R, xpg);             \
  } while (0)
      if (wide8) {
        LAUNCH_ROWS(bf16_t, false, 8, true);
      } else if (obf) {
        LAUNCH_ROWS(bf16_t, false, 4, true);
      } else if (wide) {
        LAUNCH_ROWS(bf16_t, false, 4, false);
      } else if (widef) {
        LAUNCH_ROWS(float, false, 4, false);
      } else if (narrowf) {
        LAUNCH_ROWS(float, false, 1, false);
      } else if (dtype == WSOVOD_BF16) {
        if (argmax) LAUNCH_ROWS(bf16_t, true, 2, false); else LAUNCH_ROWS(bf16_t, false, 2, false);
      } else {
        if (argmax) LAUNCH_ROWS(float, true, 2, false); else LAUNCH_ROWS(float, false, 2, false);
      }
#undef LAUNCH_ROWS
    } else if (dtype == WSOVOD_BF16) {
      if (argmax) LAUNCH_POOL(bf16_t, true); else LAUNCH_POOL(bf16_t, false);
    } else {
      if (argmax) LAUNCH_POOL(float, true); else LAUNCH_POOL(float, false);
    }
#undef LAUNCH_POOL
  } else {
    static int slot = wsovod::prof_slot("roi_pool_fwd_nchw");
    const long long total = (long long)R * C * ph * pw;
    const int grid = (int)std::min<long long>(ceil_div_ll(total, 256), 256 * 32);
    wsovod::ProfScope prof(slot, s, 0.0, bytes);
    if (dtype == WSOVOD_BF16)
      hipLaunchKernelGGL(roi_pool_fwd_nchw<bf16_t>, dim3(grid), dim3(256), 0, s, (const bf16_t*)feat, rois, roi_scale,
                         total, C, H, W, ph, pw, spatial_scale, out, out_dtype, argmax, 0);
    else
      hipLaunchKernelGGL(roi_pool_fwd_nchw<float>, dim3(grid), dim3(256), 0, s, (const float*)feat, rois, roi_scale,
                         total, C, H, W, ph, pw, spatial_scale, out, out_dtype, argmax, 0);
  }
  WS_CHECK_LAUNCH("wsovod_roi_pool_forward");
  return WSOVOD_OK;
}

int wsovod_roi_loop_pool_forward(const void* feat, int dtype, int layout, const float* rois, int R, int N, int C, int H,
                                 int W, int ph, int pw, float spatial_scale, float context_ratio, float* out,
                                 int* argmax, wsovod_stream_t stream) {
  WS_CHECK_ARG(layout == WSOVOD_NCHW || layout == WSOVOD_NHWC, "wsovod_roi_loop_pool_forward: bad layout");
  WS_CHECK_ARG(dtype == WSOVOD_F32 || dtype == WSOVOD_BF16, "wsovod_roi_loop_pool_forward: bad dtype");
  WS_CHECK_ARG(R >= 0 && N >= 0 && C > 0 && H > 0 && W > 0 && ph > 0 && pw > 0 && context_ratio > 0.f,
               "wsovod_roi_loop_pool_forward: bad shape");
  WS_CHECK_ARG((long long)H * W < (1ll << 31), "wsovod_roi_loop_pool_forward: H*W overflows int32 argmax");
  if (R == 0) return WSOVOD_OK;
  WS_CHECK_ARG(feat && rois && out && argmax, "wsovod_roi_loop_pool_forward: null pointer");
  if (layout != WSOVOD_NHWC) {
    // channels per lane need the channels innermost: the HIP backbone's layout.  The reference's NCHW tensors go through
    // `.contiguous(memory_format=torch.channels_last)` in the Python fronts (wsovod_amd/_C.py, layers/hip_ops.py).
    wsovod::set_error("wsovod_roi_loop_pool_forward: the feature map must be NHWC (channels_last)");
    return WSOVOD_ERR_UNSUPPORTED;
  }
  static int slot = wsovod::prof_slot("roi_loop_pool_fwd");
  hipStream_t s = (hipStream_t)stream;
  const int nbins = ph * pw;
  // two (channels, bins) value + argmax tiles per workgroup; 2 channels per lane when both fit next to a second workgroup
  const bool two = (C & 1) == 0 && (((uintptr_t)feat) & 7) == 0 && C >= 128 && 2 * 128 * nbins * 8 <= 80 * 1024;
  const int cpl = two ? 2 : 1, cg = 64 * cpl;
  const int share = 2 * cg * nbins * 8 <= 160 * 1024 ? 1 : 0;  // region and frame from one scan (two tiles) or from two
  const int lds = (share ? 2 : 1) * cg * nbins * 8;
  if (lds > 160 * 1024) {
    wsovod::set_error("wsovod_roi_loop_pool_forward: a %d x %d pooled tile of 64 channels does not fit the LDS", ph, pw);
    return WSOVOD_ERR_UNSUPPORTED;
  }
  const int cgroups = ceil_div(C, cg);
  const int nwaves = std::min(ph, 8);
  const double cells = (double)R * C * nbins;
  wsovod::ProfScope prof(slot, s, 0.0, cells * 24.0 + (double)N * C * H * W * (dtype == WSOVOD_BF16 ? 2 : 4));
#define LAUNCH_LOOP(T, CPL)                                                                                          \
  do {                                                                                                               \
    auto k = roi_loop_pool_fwd_nhwc_rows<T, CPL>;                                                                    \
    if (lds > 64 * 1024) WS_CHECK_HIP(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds), \
                                      "wsovod_roi_loop_pool_forward: LDS opt-in");                                   \
    hipLaunchKernelGGL(k, dim3(R * cgroups), dim3(64 * nwaves), lds, s, (const T*)feat, rois, R, C, H, W, ph, pw,      \
                       spatial_scale, context_ratio, out, argmax, cgroups, share);                                   \
  } while (0)
  if (dtype == WSOVOD_BF16) {
    if (two) LAUNCH_LOOP(bf16_t, 2); else LAUNCH_LOOP(bf16_t, 1);
  } else {
    if (two) LAUNCH_LOOP(float, 2); else LAUNCH_LOOP(float, 1);
  }
#undef LAUNCH_LOOP
  WS_CHECK_LAUNCH("wsovod_roi_loop_pool_forward");
  return WSOVOD_OK;
}

int wsovod_roi_pool_backward(const float* grad_out, const float* rois, const float* roi_scale, const int* argmax, int R,
                             int N, int C, int H, int W, int ph, int pw, int layout, float* grad_in,
                             wsovod_stream_t stream) {
  WS_CHECK_ARG(layout == WSOVOD_NCHW || layout == WSOVOD_NHWC, "wsovod_roi_pool_backward: bad layout");
  WS_CHECK_ARG(R >= 0 && N >= 0 && C > 0 && H > 0 && W > 0 && ph > 0 && pw > 0, "wsovod_roi_pool_backward: bad shape");
  if (R == 0) return WSOVOD_OK;
  WS_CHECK_ARG(grad_out && rois && argmax && grad_in, "wsovod_roi_pool_backward: null pointer");
  static int slot = wsovod::prof_slot("roi_pool_bwd");
  hipStream_t s = (hipStream_t)stream;
  const long long total = (long long)R * C * ph * pw;
  const int grid = (int)std::min<long long>(ceil_div_ll(total, 256), 256 * 32);
  wsovod::ProfScope prof(slot, s, 0.0, (double)total * 12.0);
  hipLaunchKernelGGL(roi_pool_bwd, dim3(grid), dim3(256), 0, s, grad_out, rois, roi_scale, argmax, total, C, H, W, ph,
                     pw, layout == WSOVOD_NHWC ? 1 : 0, grad_in);
  WS_CHECK_LAUNCH("wsovod_roi_pool_backward");
  return WSOVOD_OK;
}

int wsovod_roi_align_forward(const void* feat, int dtype, int layout, const float* rois, const float* roi_scale, int R,
                             int N, int C, int H, int W, int ph, int pw, float spatial_scale, int sampling_ratio,
                             int aligned, void* out, int out_dtype, wsovod_stream_t stream) {
  return wsovod_roi_align_forward_x2hi(feat, dtype, layout, rois, roi_scale, R, N, C, H, W, ph, pw, spatial_scale,
                                       sampling_ratio, aligned, out, out_dtype, nullptr, stream);
}

int wsovod_roi_align_forward_x2hi(const void* feat, int dtype, int layout, const float* rois, const float* roi_scale, int R,
                                  int N, int C, int H, int W, int ph, int pw, float spatial_scale, int sampling_ratio,
                                  int aligned, void* out, int out_dtype, void* out_hi, wsovod_stream_t stream) {
  int rc = check_common("wsovod_roi_align_forward", feat, dtype, layout, rois, R, N, C, H, W, ph, pw, out);
  WS_CHECK_ARG(!out_hi || ((out_dtype == WSOVOD_BF16X2 || out_dtype == WSOVOD_F16MX) && ((uintptr_t)out_hi & 15) == 0),
               "wsovod_roi_align_forward_x2hi: the bf16 copy goes with a bf16x2 / f16mx output");
  if (rc) return rc;
  WS_CHECK_ARG(out_dtype == WSOVOD_F32 || out_dtype == WSOVOD_BF16 || out_dtype == WSOVOD_BF16X2 || out_dtype == WSOVOD_BF16X2P ||
                   out_dtype == WSOVOD_F16MX, "wsovod_roi_align_forward: bad out_dtype");
  const bool planar_out = out_dtype == WSOVOD_BF16X2P;
  if (planar_out) {
    WS_CHECK_ARG(((long long)R * C * ph * pw) % 8 == 0, "wsovod_roi_align_forward: planar bf16x2 output needs 16-byte aligned planes");
    out_hi = (char*)out + (long long)R * C * ph * pw * 2;
  }
  WS_CHECK_ARG((out_dtype != WSOVOD_BF16X2 && out_dtype != WSOVOD_F16MX && !planar_out) || (layout == WSOVOD_NHWC && pw == 7 && ph >= 7 && ph <= 8 && C % 256 == 0 &&
                                              (C * ph * pw) % 32 == 0 && ((uintptr_t)feat & 7) == 0 && ((uintptr_t)out & 15) == 0),
               "wsovod_roi_align_forward: bf16x2 output needs NHWC, 7x7 / 8x7 bins, C a multiple of 256");
  if (R == 0) return WSOVOD_OK;
  hipStream_t s = (hipStream_t)stream;
  const int esz = dtype == WSOVOD_BF16 ? 2 : 4, osz = out_dtype == WSOVOD_BF16 ? 2 : 4;
  const double bytes = (double)N * C * H * W * esz + (double)R * C * ph * pw * osz + 20.0 * R;
  if (layout == WSOVOD_NHWC) {
    static int slot = wsovod::prof_slot("roi_align_fwd_nhwc");
    const int cgroups = ceil_div(C, 64);
    const int grid = (int)ceil_div_ll((long long)R * cgroups, 4);
    const int lds = 4 * 64 * ph * pw * 4;
    WS_CHECK_ARG(lds <= 160 * 1024, "wsovod_roi_align_forward: pooled size too large for LDS tile");
    wsovod::ProfScope prof(slot, s, 0.0, bytes);
    if (pw == 7 && ph >= 7 && ph <= 8 && ((uintptr_t)feat & 7) == 0 && C % (dtype == WSOVOD_BF16 ? 4 : 2) == 0) {
      // fast path: workgroup per (roi, 64*CPL channels), wavefront per pooled row, 8-byte loads, 28 taps in flight
      // bf16 maps of 512 channels and more (res5 of both depths) with bf16 output: 8 channels = 16 bytes per lane and
      // load, half the load instructions and L2 requests of the 4-channel form (2.15 -> 1.39 ms at the bench shape)
      const bool wide = dtype == WSOVOD_BF16 && out_dtype == WSOVOD_BF16 && C % 512 == 0 && ((uintptr_t)feat & 15) == 0;
      // (LDS of a workgroup: output transpose tile + sample-column table + separable weights)
      // Occupancy decides this kernel (it waits on L2 gathers): ONE column of every bin per iteration (7 loads in flight
      // per wavefront) and at most 128 registers, so that two 7-wavefront workgroups share a CU -- measured at the bench
      // shape against two columns per iteration at 136 - 230 registers (one workgroup per CU): bf16 1.39 -> 0.91 ms, fp32
      // map -> bf16x2 + bf16 3.71 -> 1.69 ms (with 4 instead of 2 fp32 channels per lane: 16-byte loads, two channel
      // groups per roi instead of four).
#define WS_ALIGN_ROWS(TV, CPLV, OBFV, HI)                                                                              \
  {                                                                                                                    \
    auto k = roi_align_fwd_nhwc_rows<TV, 7, CPLV, OBFV, 1, 4>;                                                         \
    const int cgv = 64 * CPLV, groupsv = ceil_div(C, cgv);                                                             \
    const int ldsv = ((cgv * ph * pw * (OBFV ? 2 : 4) + 15) & ~15) + kAlignTabBytes;                                   \
    if (ldsv > 64 * 1024)                                                                                              \
      WS_CHECK_HIP(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, ldsv),              \
                   "wsovod_roi_align_forward: LDS opt-in");                                                            \
    hipLaunchKernelGGL(k, dim3(R * groupsv), dim3(64 * ph), ldsv, s, (const TV*)feat, rois, roi_scale, C, H, W, ph,    \
                       spatial_scale, sampling_ratio, aligned, out, out_dtype, groupsv, HI);                           \
  }
      if (wide) WS_ALIGN_ROWS(bf16_t, 8, true, (void*)nullptr)
      else if (dtype == WSOVOD_BF16) WS_ALIGN_ROWS(bf16_t, 4, false, out_hi)
      else if (C % 256 == 0 && ((uintptr_t)feat & 15) == 0) WS_ALIGN_ROWS(float, 4, false, out_hi)
      else WS_ALIGN_ROWS(float, 2, false, out_hi)
#undef WS_ALIGN_ROWS
    } else if (dtype == WSOVOD_BF16) {
      auto k = roi_align_fwd_nhwc<bf16_t>;
      if (lds > 64 * 1024) WS_CHECK_HIP(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds), "wsovod_roi_align_forward: LDS opt-in");
      hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds, s, (const bf16_t*)feat, rois, roi_scale, R, C, H, W, ph, pw,
                         spatial_scale, sampling_ratio, aligned, out, out_dtype, cgroups);
    } else {
      auto k = roi_align_fwd_nhwc<float>;
      if (lds > 64 * 1024) WS_CHECK_HIP(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds), "wsovod_roi_align_forward: LDS opt-in");
      hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds, s, (const float*)feat, rois, roi_scale, R, C, H, W, ph, pw,
                         spatial_scale, sampling_ratio, aligned, out, out_dtype, cgroups);
    }
  } else {
    static int slot = wsovod::prof_slot("roi_align_fwd_nchw");
    const long long total = (long long)R * C * ph * pw;
    const int grid = (int)std::min<long long>(ceil_div_ll(total, 256), 256 * 32);
    wsovod::ProfScope prof(slot, s, 0.0, bytes);
    if (dtype == WSOVOD_BF16)
      hipLaunchKernelGGL(roi_align_fwd_nchw<bf16_t>, dim3(grid), dim3(256), 0, s, (const bf16_t*)feat, rois,
                         roi_scale, total, C, H, W, ph, pw, spatial_scale, sampling_ratio, aligned, out, out_dtype);
    else
      hipLaunchKernelGGL(roi_align_fwd_nchw<float>, dim3(grid), dim3(256), 0, s, (const float*)feat, rois, roi_scale,
                         total, C, H, W, ph, pw, spatial_scale, sampling_ratio, aligned, out, out_dtype);
  }
  WS_CHECK_LAUNCH("wsovod_roi_align_forward");
  return WSOVOD_OK;
}

int wsovod_roi_align_backward(const float* grad_out, const float* rois, const float* roi_scale, int R, int N, int C,
                              int H, int W, int ph, int pw, float spatial_scale, int sampling_ratio, int aligned,
                              int layout, float* grad_in, wsovod_stream_t stream) {
  WS_CHECK_ARG(layout == WSOVOD_NCHW || layout == WSOVOD_NHWC, "wsovod_roi_align_backward: bad layout");
  WS_CHECK_ARG(R >= 0 && N >= 0 && C > 0 && H > 0 && W > 0 && ph > 0 && pw > 0, "wsovod_roi_align_backward: bad shape");
  if (R == 0) return WSOVOD_OK;
  WS_CHECK_ARG(grad_out && rois && grad_in, "wsovod_roi_align_backward: null pointer");
  static int slot = wsovod::prof_slot("roi_align_bwd");
  hipStream_t s = (hipStream_t)stream;
  const long long total = (long long)R * C * ph * pw;
  const int grid = (int)std::min<long long>(ceil_div_ll(total, 256), 256 * 32);
  wsovod::ProfScope prof(slot, s, 0.0, (double)total * 20.0);
  hipLaunchKernelGGL(roi_align_bwd, dim3(grid), dim3(256), 0, s, grad_out, rois, roi_scale, total, C, H, W, ph, pw,
                     spatial_scale, sampling_ratio, aligned, layout == WSOVOD_NHWC ? 1 : 0, grad_in);
  WS_CHECK_LAUNCH("wsovod_roi_align_backward");
  return WSOVOD_OK;
}

}  // extern "C"

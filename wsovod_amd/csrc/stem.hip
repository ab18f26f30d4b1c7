// Stem conv1 of the WSL ResNet (3x3, stride 2, pad 1, 3 -> 64 channels, folded FrozenBN + ReLU; resnet_wsl.py:375-383,
// 410-413) straight from the uint8 image batch: normalisation ((x - mean) / std, rcnn_wsovod.py:321-328), the 27-tap
// gather and the contraction in one kernel -- the (pixels x 32) bf16 im2col operand (123 MB per 16 images) is never
// written.  HBM-bound: 23 MB of uint8 in, 245 MB of bf16 NHWC out per 16 images.
//
// A workgroup owns an 8 x 32 tile of output pixels.  Its 17 x 65 x 3 input patch is normalised ONCE into LDS as bf16
// (zero outside the image: padding is zero AFTER normalisation, and the batch canvas is zero outside each image's own
// size), every byte of the image is read and converted once instead of 2.25 times, and the gather that builds the MFMA
// A fragments is 8 LDS reads per lane and 16-pixel group with no bounds test left in it (the first version gathered
// from global memory with per-tap tests and was VALU-bound at 4x the HBM time).  K = 32 per MFMA: k = (r*3+q)*3+c,
// 27 real taps + 5 zeros -- the layout of wsovod_stem_im2col, so the folded weights are shared and the operand is
// bit-identical.  B fragments (the 64 x 32 weight) live in registers.  The MFMA takes the weight fragment first, so a
// lane ends up with 16 consecutive channels of its pixel: a pixel's 128-byte row leaves as 4 lanes x 32 bytes.
#include "common.h"

namespace {

constexpr int S_TH = 8, S_TW = 32;                    // output tile
constexpr int S_PR = 2 * S_TH + 1, S_PC = 2 * S_TW + 1;  // input patch rows / columns
constexpr int S_PCP = S_PC + 1;                       // padded row length (elements)

// Normalised patch staging shared by the two kernels.  (x - mean) / std takes 256 values per channel: a 768-entry table
// is built once per workgroup with the expression of wsovod_stem_im2col (bit-identical operand) and the 3 x 17 x 65
// patch elements become byte load -> table read -> LDS write; a wavefront walks whole patch rows (row, channel and the
// row test are scalar), a lane owns one column.  (The first form converted, subtracted and divided per element, with two
// constant divisions for its index: ~40 VALU instructions per element, 13 elements per thread -- as long as the tile's
// HBM time.)
template <bool X2>
__device__ __forceinline__ void stem_stage_patch(const uint8_t* __restrict__ plane, int Hp, int Wp, int hi, int wi,
                                                 int y0, int x0, float m0, float m1, float m2, float s0, float s1,
                                                 float s2, bf16_t* lut_hi, bf16_t* lut_lo, bf16_t* patch_hi,
                                                 bf16_t* patch_lo) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int e = tid; e < 768; e += 256) {
    const int c = e >> 8;
    const float mean = c == 0 ? m0 : (c == 1 ? m1 : m2), sd = c == 0 ? s0 : (c == 1 ? s1 : s2);
    const float v = ((float)(e & 255) - mean) / sd;  // = wsovod_stem_im2col (fp32)
    const bf16_t vh = (bf16_t)v;
    lut_hi[e] = vh;
    if (X2) lut_lo[e] = (bf16_t)(v - (float)vh);
  }
  __syncthreads();
  const bf16_t zero = (bf16_t)0.f;
  // all byte loads of the thread first (clamped addresses, no branch around them), then the table reads: the 13 + 1
  // requests of a lane travel together instead of one global-memory latency after the other
  constexpr int RPW = (3 * S_PR + 3) / 4;  // patch rows per wavefront
  const int w = 2 * x0 - 1 + lane, wc = min(max(w, 0), Wp - 1);
  const bool wok = w >= 0 && w < wi;
  const int w64 = 2 * x0 - 1 + 64, w64c = min(w64, Wp - 1);
  int byte[RPW + 1];
#pragma unroll
  for (int k = 0; k < RPW; ++k) {
    const int cr = min(wave + 4 * k, 3 * S_PR - 1);  // cr = c * S_PR + row: one patch row of one channel
    const int c = cr / S_PR, row = cr - c * S_PR;
    const int h = min(max(2 * y0 - 1 + row, 0), Hp - 1);
    byte[k] = plane[((long long)c * Hp + h) * Wp + wc];
  }
  {
    const int cr = min(tid, 3 * S_PR - 1), c = cr / S_PR, row = cr - c * S_PR;  // the 65th column: threads 0 .. 50
    const int h = min(max(2 * y0 - 1 + row, 0), Hp - 1);
    byte[RPW] = plane[((long long)c * Hp + h) * Wp + w64c];
  }
#pragma unroll
  for (int k = 0; k < RPW; ++k) {
    const int cr = wave + 4 * k;
    if (cr < 3 * S_PR) {
      const int c = cr / S_PR, row = cr - c * S_PR;
      const int h = 2 * y0 - 1 + row;
      const bool ok = h >= 0 && h < hi && wok;
      const int b = c * 256 + byte[k];
      patch_hi[cr * S_PCP + lane] = ok ? lut_hi[b] : zero;
      if (X2) patch_lo[cr * S_PCP + lane] = ok ? lut_lo[b] : zero;
    }
  }
  if (tid < 3 * S_PR) {
    const int cr = tid, c = cr / S_PR, row = cr - c * S_PR;
    const int h = 2 * y0 - 1 + row;
    const bool ok = h >= 0 && h < hi && w64 < wi;
    const int b = c * 256 + byte[RPW];
    patch_hi[cr * S_PCP + 64] = ok ? lut_hi[b] : zero;
    if (X2) patch_lo[cr * S_PCP + 64] = ok ? lut_lo[b] : zero;
  }
}

__global__ __launch_bounds__(256) void stem_conv1_kernel(const uint8_t* __restrict__ img, const int* __restrict__ sizes,
                                                         float m0, float m1, float m2, float s0, float s1, float s2,
                                                         int N, int Hp, int Wp, int Ho, int Wo, int tiles_x, int tiles_y,
                                                         const bf16_t* __restrict__ w32, const float* __restrict__ bias,
                                                         bf16_t* __restrict__ out) {
  __shared__ bf16_t patch[3 * S_PR * S_PCP];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int frow = lane & 15, g = lane >> 4;
  const int tpi = tiles_x * tiles_y;
  const int n = blockIdx.x / tpi;
  const int t_in = blockIdx.x - n * tpi;
  const int ty = t_in / tiles_x, tx = t_in - ty * tiles_x;
  const int y0 = ty * S_TH, x0 = tx * S_TW;
  const int hi = sizes[2 * n], wi = sizes[2 * n + 1];
  // ---- stage the normalised patch: element (c, row, col) = image pixel (2*y0 - 1 + row, 2*x0 - 1 + col) of channel c
  const uint8_t* plane = img + (long long)n * 3 * Hp * Wp;
  __shared__ bf16_t lut[768];
  stem_stage_patch<false>(plane, Hp, Wp, hi, wi, y0, x0, m0, m1, m2, s0, s1, s2, lut, nullptr, patch, nullptr);
  // weight fragments.  Row rho = lane & 15 of tile j is output channel 16*(rho>>2) + 4*j + (rho&3): after the MFMA
  // (weights first) lane (pixel = lane&15, g) then owns channels 16g + 4j + r -- 16 CONSECUTIVE channels.
  bf16x8 bw[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) bw[j] = *(const bf16x8*)(w32 + (16 * (frow >> 2) + 4 * j + (frow & 3)) * 32 + g * 8);
  // this lane's 8 taps: k = 8g + t -> (r, q, c) -> patch offset relative to the pixel's top-left patch element
  int toff[8];
  bool real[8];
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    const int k = 8 * g + t;
    const int tap = k / 3, c = k - 3 * tap, r = tap / 3, q = tap - 3 * r;
    real[t] = k < 27;
    toff[t] = real[t] ? (c * S_PR + r) * S_PCP + q : 0;
  }
  __syncthreads();
  const bf16_t zero = (bf16_t)0.f;
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int ly = wave * 2 + (i >> 1), lx = (i & 1) * 16 + frow;
    const bf16_t* pb = patch + (2 * ly) * S_PCP + 2 * lx;
    bf16x8 a;
#pragma unroll
    for (int t = 0; t < 8; ++t) a[t] = real[t] ? pb[toff[t]] : zero;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bw[j], a, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
  }
  // epilogue: acc[i][j][r] = pixel (row wave*2 + (i>>1), column (i&1)*16 + frow) of the tile, channel 16g + 4j + r
  f32x4 b4[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) b4[j] = *(const f32x4*)(bias + 16 * g + 4 * j);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int y = y0 + wave * 2 + (i >> 1), x = x0 + (i & 1) * 16 + frow;
    if (y >= Ho || x >= Wo) continue;
    bf16x8 lo, hi8;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      lo[r] = (bf16_t)fmaxf(acc[i][0][r] + b4[0][r], 0.f);
      lo[4 + r] = (bf16_t)fmaxf(acc[i][1][r] + b4[1][r], 0.f);
      hi8[r] = (bf16_t)fmaxf(acc[i][2][r] + b4[2][r], 0.f);
      hi8[4 + r] = (bf16_t)fmaxf(acc[i][3][r] + b4[3][r], 0.f);
    }
    bf16x8* dst = (bf16x8*)(out + (((long long)n * Ho + y) * Wo + x) * 64 + 16 * g);
    dst[0] = lo;  // (streaming stores: 0.171 -> 0.144 ms alone, nothing in the step -- the next kernel reads the map)
    dst[1] = hi8;
  }
}

// The same layer for MODEL.HIP.PRECISION = "parity": operands and output in bf16x2 (include/wsovod_hip.h).  The
// normalised patch is kept twice in LDS -- hi = bf16(v) and lo = bf16(v - hi) -- the folded weight arrives as the bf16x2
// encoding of its (64, 32) matrix ([hi 32 | lo 32] per output channel), and every (pixel group, channel tile) takes
// three MFMAs: w_hi*a_hi + w_lo*a_hi + w_hi*a_lo (fp32 accumulation; ~2^-16 relative per product instead of 2^-8).
// Output: NHWC with 64 values = 128 bf16 slots per pixel, [hi 0-31 | lo 0-31 | hi 32-63 | lo 32-63].
__global__ __launch_bounds__(256) void stem_conv1_x2_kernel(const uint8_t* __restrict__ img, const int* __restrict__ sizes,
                                                            float m0, float m1, float m2, float s0, float s1, float s2,
                                                            int N, int Hp, int Wp, int Ho, int Wo, int tiles_x, int tiles_y,
                                                            const bf16_t* __restrict__ w32x2, const float* __restrict__ bias,
                                                            bf16_t* __restrict__ out) {
  __shared__ bf16_t patch_hi[3 * S_PR * S_PCP];
  __shared__ bf16_t patch_lo[3 * S_PR * S_PCP];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int frow = lane & 15, g = lane >> 4;
  const int tpi = tiles_x * tiles_y;
  const int n = blockIdx.x / tpi;
  const int t_in = blockIdx.x - n * tpi;
  const int ty = t_in / tiles_x, tx = t_in - ty * tiles_x;
  const int y0 = ty * S_TH, x0 = tx * S_TW;
  const int hi = sizes[2 * n], wi = sizes[2 * n + 1];
  const uint8_t* plane = img + (long long)n * 3 * Hp * Wp;
  __shared__ bf16_t lut_hi[768], lut_lo[768];
  stem_stage_patch<true>(plane, Hp, Wp, hi, wi, y0, x0, m0, m1, m2, s0, s1, s2, lut_hi, lut_lo, patch_hi, patch_lo);
  bf16x8 bwh[4], bwl[4];  // weight rows permuted as in the bf16 kernel: a lane ends up with 16 consecutive channels
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const bf16_t* wr = w32x2 + (16 * (frow >> 2) + 4 * j + (frow & 3)) * 64 + g * 8;
    bwh[j] = *(const bf16x8*)wr;
    bwl[j] = *(const bf16x8*)(wr + 32);
  }
  int toff[8];
  bool real[8];
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    const int k = 8 * g + t;
    const int tap = k / 3, c = k - 3 * tap, r = tap / 3, q = tap - 3 * r;
    real[t] = k < 27;
    toff[t] = real[t] ? (c * S_PR + r) * S_PCP + q : 0;
  }
  __syncthreads();
  const bf16_t zero = (bf16_t)0.f;
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int ly = wave * 2 + (i >> 1), lx = (i & 1) * 16 + frow;
    const int po = (2 * ly) * S_PCP + 2 * lx;
    bf16x8 ah, al;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      ah[t] = real[t] ? patch_hi[po + toff[t]] : zero;
      al[t] = real[t] ? patch_lo[po + toff[t]] : zero;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      f32x4 c4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bwh[j], ah, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
      c4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bwl[j], ah, c4, 0, 0, 0);
      acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bwh[j], al, c4, 0, 0, 0);
    }
  }
  f32x4 b4[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) b4[j] = *(const f32x4*)(bias + 16 * g + 4 * j);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int y = y0 + wave * 2 + (i >> 1), x = x0 + (i & 1) * 16 + frow;
    if (y >= Ho || x >= Wo) continue;
    bf16x8 h0, h1, l0, l1;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float v0 = fmaxf(acc[i][0][r] + b4[0][r], 0.f), v1 = fmaxf(acc[i][1][r] + b4[1][r], 0.f);
      const float v2 = fmaxf(acc[i][2][r] + b4[2][r], 0.f), v3 = fmaxf(acc[i][3][r] + b4[3][r], 0.f);
      h0[r] = (bf16_t)v0; h0[4 + r] = (bf16_t)v1; h1[r] = (bf16_t)v2; h1[4 + r] = (bf16_t)v3;
      l0[r] = (bf16_t)(v0 - (float)h0[r]); l0[4 + r] = (bf16_t)(v1 - (float)h0[4 + r]);
      l1[r] = (bf16_t)(v2 - (float)h1[r]); l1[4 + r] = (bf16_t)(v3 - (float)h1[4 + r]);
    }
    // channels 16g .. 16g+15 of the pixel's 128 slots: group g>>1, hi at 16*(g&1), lo 32 slots further
    bf16_t* dst = out + (((long long)n * Ho + y) * Wo + x) * 128 + 64 * (g >> 1) + 16 * (g & 1);
    *(bf16x8*)dst = h0;
    *(bf16x8*)(dst + 8) = h1;
    *(bf16x8*)(dst + 32) = l0;
    *(bf16x8*)(dst + 40) = l1;
  }
}

}  // namespace

extern "C" int wsovod_stem_conv1(const unsigned char* img, const int* sizes, const float* mean_host,
                                 const float* std_host, int N, int Hp, int Wp, const void* w32, const float* bias,
                                 void* out, wsovod_stream_t stream) {
  WS_CHECK_ARG(N >= 0 && Hp > 0 && Wp > 0, "wsovod_stem_conv1: bad shape");
  if (N == 0) return WSOVOD_OK;
  WS_CHECK_ARG(img && sizes && mean_host && std_host && w32 && bias && out, "wsovod_stem_conv1: null pointer");
  WS_CHECK_ARG((((uintptr_t)w32 | (uintptr_t)bias | (uintptr_t)out) & 15) == 0,
               "wsovod_stem_conv1: weights / bias / output must be 16-byte aligned");
  const int Ho = (Hp - 1) / 2 + 1, Wo = (Wp - 1) / 2 + 1;
  const long long total = (long long)N * Ho * Wo;
  const int tiles_x = (Wo + S_TW - 1) / S_TW, tiles_y = (Ho + S_TH - 1) / S_TH;
  WS_CHECK_ARG((long long)N * tiles_x * tiles_y < (1ll << 31), "wsovod_stem_conv1: too many tiles for one launch");
  static int slot = wsovod::prof_slot("stem_conv1_fused");
  hipStream_t s = (hipStream_t)stream;
  wsovod::ProfScope prof(slot, s, 2.0 * total * 64 * 27, (double)N * 3 * Hp * Wp + (double)total * 128);
  hipLaunchKernelGGL(stem_conv1_kernel, dim3((unsigned)(N * tiles_x * tiles_y)), dim3(256), 0, s, img, sizes,
                     mean_host[0], mean_host[1], mean_host[2], std_host[0], std_host[1], std_host[2], N, Hp, Wp, Ho, Wo,
                     tiles_x, tiles_y, (const bf16_t*)w32, bias, (bf16_t*)out);
  WS_CHECK_LAUNCH("wsovod_stem_conv1");
  return WSOVOD_OK;
}

extern "C" int wsovod_stem_conv1_x2(const unsigned char* img, const int* sizes, const float* mean_host,
                                    const float* std_host, int N, int Hp, int Wp, const void* w32x2, const float* bias,
                                    void* out, wsovod_stream_t stream) {
  WS_CHECK_ARG(N >= 0 && Hp > 0 && Wp > 0, "wsovod_stem_conv1_x2: bad shape");
  if (N == 0) return WSOVOD_OK;
  WS_CHECK_ARG(img && sizes && mean_host && std_host && w32x2 && bias && out, "wsovod_stem_conv1_x2: null pointer");
  WS_CHECK_ARG((((uintptr_t)w32x2 | (uintptr_t)bias | (uintptr_t)out) & 15) == 0,
               "wsovod_stem_conv1_x2: weights / bias / output must be 16-byte aligned");
  const int Ho = (Hp - 1) / 2 + 1, Wo = (Wp - 1) / 2 + 1;
  const long long total = (long long)N * Ho * Wo;
  const int tiles_x = (Wo + S_TW - 1) / S_TW, tiles_y = (Ho + S_TH - 1) / S_TH;
  WS_CHECK_ARG((long long)N * tiles_x * tiles_y < (1ll << 31), "wsovod_stem_conv1_x2: too many tiles for one launch");
  static int slot = wsovod::prof_slot("stem_conv1_fused_bf16x2");
  hipStream_t s = (hipStream_t)stream;
  wsovod::ProfScope prof(slot, s, 6.0 * total * 64 * 27, (double)N * 3 * Hp * Wp + (double)total * 256);
  hipLaunchKernelGGL(stem_conv1_x2_kernel, dim3((unsigned)(N * tiles_x * tiles_y)), dim3(256), 0, s, img, sizes,
                     mean_host[0], mean_host[1], mean_host[2], std_host[0], std_host[1], std_host[2], N, Hp, Wp, Ho, Wo,
                     tiles_x, tiles_y, (const bf16_t*)w32x2, bias, (bf16_t*)out);
  WS_CHECK_LAUNCH("wsovod_stem_conv1_x2");
  return WSOVOD_OK;
}

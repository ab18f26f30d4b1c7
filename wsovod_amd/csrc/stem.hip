// Stem conv1 of the WSL ResNet (3x3, stride 2, pad 1, 3 -> 64 channels, folded FrozenBN + ReLU; resnet_wsl.py:375-383,
// 410-413) straight from the uint8 image batch: normalisation ((x - mean) / std, rcnn_wsovod.py:321-328), the 27-tap
// gather and the contraction in one kernel -- the (pixels x 32) bf16 im2col operand (123 MB per 16 images) is never
// written.  HBM-bound: 23 MB of uint8 in, 245 MB of bf16 NHWC out per 16 images.
//
// A wavefront owns 64 consecutive output pixels x 64 channels = 4 x 4 MFMA tiles of one K = 32 step (k = (r*3+q)*3+c,
// 27 real taps + 5 zeros: the layout of wsovod_stem_im2col, so the folded weights are shared).  A fragments are built
// in registers: lane (pixel = lane & 15 of the tile, k-group g = lane >> 4) gathers its 8 taps as byte loads --
// consecutive lanes read every second byte of an image row -- normalises them with the same expression as the
// im2col kernel (bit-identical operand) and packs them to bf16.  B fragments (the 64 x 32 weight) live in registers.
// The MFMA takes the weight fragment first, so a lane ends up with 4 consecutive channels of its pixel: one 8-byte
// store per tile.
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void stem_conv1_kernel(const uint8_t* __restrict__ img, const int* __restrict__ sizes,
                                                         float m0, float m1, float m2, float s0, float s1, float s2,
                                                         int N, int Hp, int Wp, int Ho, int Wo,
                                                         const bf16_t* __restrict__ w32, const float* __restrict__ bias,
                                                         bf16_t* __restrict__ out) {
  // normalised value of every (channel, byte) as bf16: 768 table entries replace a float divide per gathered tap
  __shared__ bf16_t lut[3 * 256];
  for (int e = threadIdx.x; e < 768; e += 256) {
    const int c = e >> 8;
    const float mean = c == 0 ? m0 : (c == 1 ? m1 : m2), sd = c == 0 ? s0 : (c == 1 ? s1 : s2);
    lut[e] = (bf16_t)(((float)(e & 255) - mean) / sd);
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int frow = lane & 15, g = lane >> 4;
  const long long total = (long long)N * Ho * Wo;
  const long long m_base = ((long long)blockIdx.x * 4 + wave) * 64;
  if (m_base >= total) return;
  // weight fragments.  Row rho = lane & 15 of tile j is output channel 16*(rho>>2) + 4*j + (rho&3): after the MFMA
  // (weights first) lane (pixel = lane&15, g) then owns channels 16g + 4j + r -- 16 CONSECUTIVE channels, so a pixel's
  // 128-byte row is written as 4 lanes x 32 contiguous bytes.
  bf16x8 bw[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) bw[j] = *(const bf16x8*)(w32 + (16 * (frow >> 2) + 4 * j + (frow & 3)) * 32 + g * 8);
  // this lane's 8 taps: k = 8g + t -> (r, q, c); image offset relative to (c = 0, h = 2*ho, w = 2*wo)
  int dr[8], dq[8], ch[8];
  long long doff[8];
  bool real[8];
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    const int k = 8 * g + t;
    const int tap = k / 3, c = k - 3 * tap, r = tap / 3, q = tap - 3 * r;
    real[t] = k < 27;
    dr[t] = r - 1;
    dq[t] = q - 1;
    ch[t] = c << 8;
    doff[t] = ((long long)c * Hp + (r - 1)) * Wp + (q - 1);
  }
  const bf16_t zero = (bf16_t)0.f;
  f32x4 acc[4][4];
  // (n, ho, wo) of the wavefront's first pixel: one wave-uniform division (the launcher guarantees total < 2^31);
  // the lanes' pixels follow by carries -- this kernel is VALU-bound and per-lane 64-bit divisions were 70 % of it.
  const unsigned mb = (unsigned)__builtin_amdgcn_readfirstlane((int)m_base);
  const unsigned row0 = mb / (unsigned)Wo;
  const int wo0 = (int)(mb - row0 * (unsigned)Wo);
  const int n0 = (int)(row0 / (unsigned)Ho);
  const int ho0 = (int)(row0 - (unsigned)n0 * (unsigned)Ho);
  const int last = (int)(total - 1 - m_base);  // >= 0
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int o = min(i * 16 + frow, last);  // clamp: the tail lanes recompute the last pixel
    int wo = wo0 + o, ho = ho0, n = n0;
    while (wo >= Wo) {  // at most ceil(64 / Wo) rounds
      wo -= Wo;
      if (++ho == Ho) {
        ho = 0;
        ++n;
      }
    }
    const int hi = sizes[2 * n], wi = sizes[2 * n + 1];
    const uint8_t* base = img + ((long long)n * 3 * Hp + 2 * ho) * Wp + 2 * wo;
    bf16x8 a;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int h = 2 * ho + dr[t], w = 2 * wo + dq[t];
      const bool ok = real[t] && h >= 0 && w >= 0 && h < hi && w < wi;
      a[t] = ok ? lut[ch[t] + base[doff[t]]] : zero;  // padding is zero AFTER normalisation
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
      acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bw[j], a, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
  }
  // epilogue: acc[i][j][r] = pixel (i*16 + frow), channel 16g + 4j + r
  f32x4 b4[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) b4[j] = *(const f32x4*)(bias + 16 * g + 4 * j);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const long long m = m_base + i * 16 + frow;
    if (m >= total) continue;
    bf16x8 lo, hi8;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      lo[r] = (bf16_t)fmaxf(acc[i][0][r] + b4[0][r], 0.f);
      lo[4 + r] = (bf16_t)fmaxf(acc[i][1][r] + b4[1][r], 0.f);
      hi8[r] = (bf16_t)fmaxf(acc[i][2][r] + b4[2][r], 0.f);
      hi8[4 + r] = (bf16_t)fmaxf(acc[i][3][r] + b4[3][r], 0.f);
    }
    bf16x8* dst = (bf16x8*)(out + m * 64 + 16 * g);
    dst[0] = lo;
    dst[1] = hi8;
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
  WS_CHECK_ARG(total < (1ll << 31) - 64, "wsovod_stem_conv1: more than 2^31 output pixels in one launch");
  static int slot = wsovod::prof_slot("stem_conv1_fused");
  hipStream_t s = (hipStream_t)stream;
  wsovod::ProfScope prof(slot, s, 2.0 * total * 64 * 27, (double)N * 3 * Hp * Wp + (double)total * 128);
  hipLaunchKernelGGL(stem_conv1_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, img, sizes, mean_host[0],
                     mean_host[1], mean_host[2], std_host[0], std_host[1], std_host[2], N, Hp, Wp, Ho, Wo,
                     (const bf16_t*)w32, bias, (bf16_t*)out);
  WS_CHECK_LAUNCH("wsovod_stem_conv1");
  return WSOVOD_OK;
}

// Round 6 (MODEL.HIP.PRECISION = "parity_mx"): the forward contractions of the big layers -- the res4 / res5 implicit-GEMM
// convolutions and the two FC layers of the box head -- on gfx950's block-scaled matrix instruction: the two cross terms of the
// three-product forward (hi*lo + lo*hi, 2^-12 of a product when hi is an fp16 rounding) as ONE
// v_mfma_scale_f32_32x32x64_f8f6f4 on e4m3 planes, hi*hi as two v_mfma_f32_32x32x16_f16: per 32x32 output tile and 32 values of
// K 128 matrix-pipe cycles where the bf16x2 form (gemm8.hip) spends 192.
//
// Operand format "f16mx" (f16mx.h; the same 4 bytes per value and row stride as bf16x2): a row is groups of 32 values = 128
// bytes [32 x fp16 hi | 32 x e4m3 q | 32 x e4m3 ql], q = e4m3(x 2^-s), ql = e4m3((x - hi) 2^-(s - 11)) -- |x - hi| is at most
// half an fp16 ulp, so the lo plane's scale is TIED 11 binades below the q plane's and one exponent serves both.  Scales are
// LOOP CONSTANTS of the kernel: activations (A) carry none (s = 0, written by the producing epilogue without a row maximum),
// weights (B) one E8M0 byte per row (or per row segment) in a side array.  e4m3's own exponent carries the dynamic range
// inside a row.  Numerics gate: profiles/r06_mx_gate.md (tools/mx_emulation.py, "THE BUILD": logits 2.1e-4 from the oracle).
//
// The kernel is the 8-wavefront 256x256 two-phase staggered tile of gemm8.hip with
//   * 32-row MFMA tiles: lane (r = lane & 31, h = lane >> 5) reads, per tile and K-step, 32 bytes of the hi plane (chunks 2h,
//     2h+1: fp16 values 16h .. 16h+15, the two 32x32x16 products) and 32 bytes of an fp8 plane -- A: q for h = 0, ql for h = 1;
//     B: ql for h = 0, q for h = 1 -- so the scaled MFMA's K halves are q_a*ql_b and ql_a*q_b.  Same 128-byte LDS rows, same
//     DMA and XOR swizzle as the bf16x2 tile.  The two 16-byte pieces of an fp8 operand are read into FIXED adjacent physical
//     registers (the instruction takes 8 consecutive VGPRs; through allocator-chosen registers hipcc assembled them with
//     v_movs, kept a second copy alive and spilled);
//   * a RING of three K-steps for A and B refilled right after its only reads: every LDS-DMA piece is requested two K-steps
//     (three to four phases) before its use, four pieces in each phase; one counted wait per K-step.  With the bf16x2 tile's
//     schedule (one K-step ahead, 6 + 2 pieces) the 512-cycle phases of this format waited on memory: 663 -> 714 TFLOP/s on
//     fc1, matrix pipe 78 -> 90 % busy (tools/mx_phases.py, tools/mx_abl_pmc.sh); what is left is the clock under the power
//     limit (1.45 GHz with fragment reads + DMA, 2.2 GHz on the bare MFMA loop);
//   * B rows permuted at staging so that a lane owns 16 CONSECUTIVE output columns per tile (32x32 accumulator layout:
//     row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)): the epilogue writes 64 bytes per lane and row, also as f16mx;
//   * CONV: the implicit-GEMM gather of gemm8.hip's lean form (per-lane pixel offset + tap validity bits, the tap's
//     displacement as the scalar offset, the fused 1x1 projection shortcut as extra K-steps on a second input), the offsets of
//     the K-step requested next computed under the products of phase B.
// Replaces (opt-in): roi_heads/box_head.py:60-75 (fc1 / fc2 forward, F.linear) and backbone/resnet_wsl.py:94-110 (the conv +
// FrozenBN + shortcut + ReLU of the res4 / res5 BasicBlocks) for the "parity_mx" precision.
#include "gemm_common.h"
#include "f16mx.h"

#include <stdlib.h>

#include <algorithm>
#include <type_traits>
#include <vector>

namespace wsovod_gemm {
namespace {

using namespace wsovod_mx;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef int i32x8 __attribute__((ext_vector_type(8)));

struct MxArgs {
  GemmArgs g;               // (K, lda, ldb, Cin, Cin2 in 2-byte slots: a value is two slots, a K-step 64 slots = 32 values)
  const unsigned char* sa;  // [M][nseg_a] E8M0 bytes of the A rows; NULL = unit scale (activations)
  const unsigned char* sb;  // [N][nseg_b]
  int nseg_a, nseg_b;       // equal K segments per row
  void* c_bf16;             // optional plain bf16 copy of C (the operand of the NEXT layer's weight gradient, the mask source)
  long long ld_cb;
  float* dbg;               // (-DMX_STAMPS builds, tools/mx_phases.py) 2 x 16 tick sums
};

template <int OFF>
__device__ __forceinline__ void mx_read(u32x4& dst, unsigned addr) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF));
#endif
}

__device__ __forceinline__ i32x8 mx_cat(const u32x4 a, const u32x4 b) {
  return i32x8{(int)a[0], (int)a[1], (int)a[2], (int)a[3], (int)b[0], (int)b[1], (int)b[2], (int)b[3]};
}

// EPI: the epilogue's output / residual formats as compile-time facts (0 = read from the arguments: every combination, ~100 KB
// of code behind runtime branches that each tile streamed through the instruction cache; the hot combinations are built
// apart): 1 = f16mx out, no residual; 2 = f16mx out + f16mx residual; 3 = fp32 out + f16mx residual; 4 = bf16x2 out, no
// residual -- each with N a multiple of 64 (the staged row-image stores).
template <bool CONV, int EPI = 0>
__global__ __launch_bounds__(512) void gemm256_mx_kernel(const MxArgs q) {
  const GemmArgs& p = q.g;
  [[maybe_unused]] constexpr int BM = 256, BN = 256, BKE = 64, EPC = 8, esz = 2, LR = 64;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  [[maybe_unused]] char* sA = smem;                 // 3 x [256][128 B]: a ring of three K-steps
  [[maybe_unused]] char* sB = smem + 3 * BM * 128;  // 2 x [256][128 B]: refilled right after its only reads (phase A)

  const int nwg = p.tiles_m * p.tiles_n;
  int wg;
  {
    const int bid = p.ksplit > 1 ? (int)(blockIdx.x % (unsigned)nwg) : (int)blockIdx.x;  // split-K: slice-major copies of the grid
    const int qq = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    wg = (xcd < r ? xcd * (qq + 1) : r * (qq + 1) + (xcd - r) * qq) + (bid >> 3);
  }
  const int group_size = p.group_m * p.tiles_n;
  const int group_id = wg / group_size;
  const int first_m = group_id * p.group_m;
  const int gm = min(p.tiles_m - first_m, p.group_m);
  const int in_group = wg - group_id * group_size;
  const int tile_m = first_m + in_group % gm;
  const int tile_n = in_group / gm;
  [[maybe_unused]] const int m0 = p.m_base + tile_m * BM, n0 = tile_n * BN;  // (m_base: a launch may cover rows [m_base, M) only)
  // split-K (the last, partly filled round of tiles of a launch: wsovod_gemm_f16mx): slice z of the grid reduces K-steps
  // [z * slice_steps, (z + 1) * slice_steps) and stores its raw fp32 sums; mx_splitk_finalize_kernel adds them
  const int kslice = p.ksplit > 1 ? (int)(blockIdx.x / (unsigned)nwg) : 0;
  [[maybe_unused]] const int kt_base = kslice * p.slice_steps;

#if defined(MX_STAMPS)
  const unsigned long long st_begin = __builtin_amdgcn_s_memtime();  // (tile level: prologue / first data / loop / epilogue / drain)
#endif
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  [[maybe_unused]] const int wr = wave >> 2, wc = wave & 3;
  const int lrow = tid >> 3;
  [[maybe_unused]] const int lchunk = (tid & 7) ^ ((lrow >> 1) & 7);  // swizzle on the DMA source
  [[maybe_unused]] const int r32 = lane & 31, hh = lane >> 5;

#if defined(__HIP_DEVICE_COMPILE__)
  constexpr unsigned OOB = 0x80000000u;
  __amdgpu_buffer_rsrc_t rsrcA, rsrcB;
  [[maybe_unused]] __amdgpu_buffer_rsrc_t rsrcA2;
  // conv: a pixel offset with the filter at its top-left tap is negative along the image's top / left border, and the range
  // check adds voffset + soffset without wrapping: the resource starts `cbias` bytes in front of the map and every per-lane
  // offset carries +cbias (the bytes in front are never addressed: their taps are the invalid ones)
  [[maybe_unused]] const int cbias = CONV ? (p.pad * p.W + p.pad) * p.Cin * esz : 0;
  if constexpr (CONV) {
    rsrcA = __builtin_amdgcn_make_buffer_rsrc((void*)(p.A - cbias), 0, (int)min(p.a_bytes + (long long)cbias, (long long)0x7fffffff),
                                              0x00020000);
    rsrcA2 = __builtin_amdgcn_make_buffer_rsrc((void*)(p.A2 ? p.A2 : p.A), 0, (int)(p.A2 ? p.a2_bytes : 0), 0x00020000);
  } else {
    const long long rows = min(BM, p.M - m0);
    rsrcA = __builtin_amdgcn_make_buffer_rsrc((void*)(p.A + (long long)m0 * p.lda * esz), 0, (int)(rows * p.lda * esz), 0x00020000);
  }
  {
    const long long rows = min(BN, p.N - n0);
    rsrcB = __builtin_amdgcn_make_buffer_rsrc((void*)(p.B + (long long)n0 * p.ldb * esz), 0, (int)(rows * p.ldb * esz), 0x00020000);
  }
  // plain GEMM: one per-lane constant per operand; the pass (64 rows further) is a scalar offset.  Rows beyond the matrix
  // fall outside the tile's buffer resource (its size is the tile's valid rows): the hardware range check returns zeros
  [[maybe_unused]] const unsigned va0 = (unsigned)(((long long)lrow * p.lda + lchunk * EPC) * esz);
  // B pass i = the 64 columns of the wavefronts with wc == i; LDS row (tile u = lrow >> 5, tile row f = lrow & 31) is fed from
  // source column 32 u + 16 ((f >> 2) & 1) + 4 (f >> 3) + (f & 3): accumulator register 4 g + r of lane half h then holds
  // column 32 u + 16 h + 4 g + r, i.e. a lane owns 16 consecutive columns of the tile
  const int fB = lrow & 31;
  const unsigned vb0 = (unsigned)(((long long)(32 * (lrow >> 5) + 16 * ((fB >> 2) & 1) + 4 * (fB >> 3) + (fB & 3)) * p.ldb + lchunk * EPC) * esz);
  [[maybe_unused]] const int passA = (int)(LR * p.lda * esz);
  const int passB = (int)(LR * p.ldb * esz);  // (scalar; < 2^31: launcher)
  typedef __attribute__((address_space(3))) void lds_void;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const int nk = p.ksplit > 1 ? max(0, min(p.K / BKE - kt_base, p.slice_steps)) : p.K / BKE;  // (launcher: K = whole K-steps)

  // ---- conv: the four rows a lane stages (pass i: tile row lrow + 64 i) as pixels -- offset of the pixel with the filter at
  // its top-left tap, one validity bit per tap, the output pixel in the shortcut's input
  struct Tap { int r, q, c0; };  // c0 >= Cin: the K-steps of the fused 1x1 shortcut (second input at channel c0 - Cin)
  [[maybe_unused]] unsigned pixb[4], pix2v[4], vmask[4], va[4];
  // (channel chunk, tap) order with the tap innermost: the taps of a chunk re-read the same input pixels while they are still
  // in L2.  Branch-free on purpose (scalar selects): as `if`s hipcc turned the tap state into a web of scalar branches
  // through the K loop, one of them in the middle of the products of phase B (tools/mx_phases.py: +450 cycles per K-step)
  auto tap_next = [&](const Tap t) {
    const bool sec = t.c0 >= p.Cin;
    const int q1 = t.q + 1;
    const bool wq = q1 >= p.KW;
    const int r1 = t.r + (wq ? 1 : 0);
    const bool wrp = r1 >= p.KH;
    Tap n;
    n.q = sec ? t.q : (wq ? 0 : q1);
    n.r = sec ? t.r : (wrp ? 0 : r1);
    n.c0 = t.c0 + ((sec || (wq && wrp)) ? BKE : 0);
    return n;
  };
  // per-lane offsets of tap t for this lane's four rows -- the pixel offset where the tap lies inside the image, out of
  // range where it does not; the tap's own displacement is the scalar offset.  Branch-free
  [[maybe_unused]] auto conv_va = [&](const Tap t) {
    const bool sec = t.c0 >= p.Cin;
    const unsigned secm = sec ? 0xffffffffu : 0u;                    // (scalar masks instead of a uniform branch)
    const unsigned tapbit = (1u << (t.r * p.KW + t.q)) & ~secm;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const unsigned alt = (pix2v[i] & secm) | (OOB & ~secm);
      va[i] = (vmask[i] & tapbit) ? pixb[i] : alt;
    }
  };
  // scalar byte offsets of a K-step: into the A operand (conv: the tap's displacement + channel chunk) and the B rows
  auto soff_a = [&](int kt, const Tap t) -> int {
    if (!CONV) return (kt + kt_base) * (BKE * esz);
    const int main_off = (((t.r * p.W + t.q) * p.dil) * p.Cin + t.c0) * esz, sec_off = (t.c0 - p.Cin) * esz;
    return t.c0 >= p.Cin ? sec_off : main_off;
  };
  auto soff_b = [&](int kt, const Tap t) -> int {
    if (!CONV) return (kt + kt_base) * (BKE * esz);
    const int main_k = (t.r * p.KW + t.q) * p.Cin + t.c0, sec_k = p.KH * p.KW * p.Cin + (t.c0 - p.Cin);
    return (t.c0 >= p.Cin ? sec_k : main_k) * esz;
  };
  auto dma_a = [&](int stage_off, int i, int so, bool second) {  // stage_off: byte offset of the ring stage (scalar)
    char* dA = sA + stage_off + wave_u * 1024 + LR * i * 128;
    if constexpr (CONV) {
      if (second) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcA2, (lds_void*)dA, 16, (int)va[i], so, 0, 0);
      else __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcA, (lds_void*)dA, 16, (int)va[i], so, 0, 0);
    } else {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcA, (lds_void*)dA, 16, (int)va0, so + i * passA, 0, 0);
    }
  };
  auto dma_b = [&](int buf_off, int i, int so) {
    char* dB = sB + buf_off + wave_u * 1024 + LR * i * 128;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcB, (lds_void*)dB, 16, (int)vb0, so + i * passB, 0, 0);
  };
  // K-steps 0 and 1 (the loop keeps two K-steps in flight: the pieces of K-step kt + 2 are requested during K-step kt).  Their
  // B rows first: they are in flight while the conv form decodes its pixels (4 integer divisions per lane, the tap masks)
  Tap t2{0, 0, 0};  // (conv) the tap of the K-step requested next
  if (CONV && kt_base > 0) {  // split-K slice of a conv: the (filter tap, channel chunk) of its first K-step
    const int taps = p.KH * p.KW, nk_main = taps * (p.Cin / BKE);
    if (kt_base >= nk_main) {
      t2.c0 = p.Cin + (kt_base - nk_main) * BKE;
    } else {
      const int chunk = kt_base / taps, tap = kt_base - chunk * taps;
      t2.r = tap / p.KW;
      t2.q = tap - t2.r * p.KW;
      t2.c0 = chunk * BKE;
    }
  }
  const Tap t1 = CONV ? tap_next(t2) : t2;
  {
    constexpr int ST = BM * 128;
    const int sb0 = soff_b(0, t2), sb1 = soff_b(1, t1);
    dma_b(0, 0, sb0); dma_b(0, 1, sb0); dma_b(0, 2, sb0); dma_b(0, 3, sb0);
    if (nk > 1) { dma_b(ST, 0, sb1); dma_b(ST, 1, sb1); dma_b(ST, 2, sb1); dma_b(ST, 3, sb1); }
  }
  if constexpr (CONV) {
    const int hw = p.Ho * p.Wo;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = m0 + lrow + LR * i;
      const bool ok = m < p.M;
      const int mm = ok ? m : 0;
      const int img = mm / hw, rem = mm - img * hw;
      const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
      const int hi0 = ok ? ho * p.stride - p.pad : -(1 << 28), wi0 = wo * p.stride - p.pad;
      const int a_img = (img * p.H * p.W * p.Cin + lchunk * EPC) * esz;
      pixb[i] = ok ? (unsigned)(a_img + ((hi0 * p.W + wi0) * p.Cin) * esz + cbias) : 0u;
      pix2v[i] = (ok && p.A2) ? (unsigned)((((img * p.Ho + ho) * p.Wo + wo) * p.Cin2 + lchunk * EPC) * esz) : OOB;
      // branch-free, KH + KW steps: valid filter rows x valid filter columns (a row past M: hi0 = -2^28)
      unsigned rowm = 0, colm = 0;
      for (int r = 0; r < p.KH; ++r) rowm |= (unsigned)((unsigned)(hi0 + r * p.dil) < (unsigned)p.H) << r;
      for (int c = 0; c < p.KW; ++c) colm |= (unsigned)((unsigned)(wi0 + c * p.dil) < (unsigned)p.W) << c;
      unsigned mk = 0;
      for (int r = 0; r < p.KH; ++r) mk |= ((rowm >> r) & 1u) ? (colm << (r * p.KW)) : 0u;
      vmask[i] = mk;
    }
  }
  {
    if constexpr (CONV) conv_va(t2);
    const int sa0 = soff_a(0, t2);
    const bool sec = CONV && t2.c0 >= p.Cin;
    dma_a(0, 0, sa0, sec); dma_a(0, 2, sa0, sec); dma_a(0, 1, sa0, sec); dma_a(0, 3, sa0, sec);
  }
  if constexpr (CONV) t2 = t1;
  if (nk > 1) {
    constexpr int ST = BM * 128;
    if constexpr (CONV) conv_va(t2);
    const int sa1 = soff_a(1, t2);
    const bool sec = CONV && t2.c0 >= p.Cin;
    dma_a(ST, 0, sa1, sec); dma_a(ST, 2, sa1, sec); dma_a(ST, 1, sa1, sec); dma_a(ST, 3, sa1, sec);
  }
  if constexpr (CONV) {
    t2 = tap_next(t2);
    conv_va(t2);
  }

  // ---- block scales: per lane the rows of its 4 A tiles and 2 B tiles; loop constants inside a row segment.  The ql lanes
  // carry scale - 11
  const unsigned subA = hh ? 11u : 0u;   // A fragments: h = 0 reads q (scale s), h = 1 reads ql (s - 11)
  const unsigned subB = hh ? 0u : 11u;   // B fragments: h = 0 reads ql, h = 1 reads q
  unsigned sa_c[4], sb_c[2];
  auto load_scales_a = [&](int seg) {
#pragma unroll
    for (int t = 0; t < 4; ++t)
      sa_c[t] = (q.sa ? (unsigned)q.sa[(long long)min(m0 + wr * 128 + 32 * t + r32, p.M - 1) * q.nseg_a + seg] : 127u) - subA;
  };
  auto load_scales_b = [&](int seg) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int col = n0 + wc * 64 + 32 * u + 16 * ((r32 >> 2) & 1) + 4 * (r32 >> 3) + (r32 & 3);
      sb_c[u] = (unsigned)q.sb[(long long)min(col, p.N - 1) * q.nseg_b + seg] - subB;
    }
  };
  load_scales_a(0);
  load_scales_b(0);
  const int seg_a = nk / q.nseg_a, seg_b = nk / q.nseg_b;  // K-steps per segment

  f32x16 acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;

  // per-lane LDS addresses of the four 16-byte pieces a lane reads of its fragment row (the row base inside a 32-row tile is
  // r32: the swizzle term is a per-lane constant)
  typedef __attribute__((address_space(3))) const char lds_cchar;
  const unsigned ldsA = (unsigned)(size_t)(lds_cchar*)sA, ldsB = (unsigned)(size_t)(lds_cchar*)sB;
  const unsigned sw = (unsigned)((r32 >> 1) & 7);
  const unsigned rowA = ldsA + (unsigned)((wr * 128 + r32) * 128), rowB = ldsB + (unsigned)((wc * 64 + r32) * 128);
  const unsigned cA0 = rowA + (((unsigned)(2 * hh) ^ sw) << 4), cA1 = rowA + (((unsigned)(2 * hh + 1) ^ sw) << 4);
  const unsigned cA2 = rowA + (((unsigned)(4 + 2 * hh) ^ sw) << 4), cA3 = rowA + (((unsigned)(5 + 2 * hh) ^ sw) << 4);
  const unsigned cB0 = rowB + (((unsigned)(2 * hh) ^ sw) << 4), cB1 = rowB + (((unsigned)(2 * hh + 1) ^ sw) << 4);
  const unsigned cB2 = rowB + (((unsigned)(6 - 2 * hh) ^ sw) << 4), cB3 = rowB + (((unsigned)(7 - 2 * hh) ^ sw) << 4);

  // fragments: the fp16 pieces in allocator-chosen registers; the two 16-byte pieces of an fp8 operand in FIXED adjacent
  // physical registers (the scaled MFMA takes 8 consecutive VGPRs: no v_mov assembly, no second copy alive)
  u32x4 af[2][2], bf[2][2];      // [tile of the phase][fp16 values 0-7 | 8-15 of the lane's half]
  u32x4 a8l[2], a8h[2], b8l[2], b8h[2];
#define MX_RA0L "v[224:227]"
#define MX_RA0H "v[228:231]"
#define MX_RA1L "v[232:235]"
#define MX_RA1H "v[236:239]"
#define MX_RB0L "v[240:243]"
#define MX_RB0H "v[244:247]"
#define MX_RB1L "v[248:251]"
#define MX_RB1H "v[252:255]"
#define MX_READ_P(REG, VAR, OFF, ADDR) asm volatile("ds_read_b128 %0, %1 offset:%2" : "={" REG "}"(VAR) : "v"(ADDR), "n"(OFF))
#define MX_VMCNT(N) asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory")
#define MX_LGKM_ALL()                                                                                                       \
  asm volatile("s_waitcnt lgkmcnt(0)"                                                                                       \
               : "+v"(af[0][0]), "+v"(af[0][1]), "+v"(af[1][0]), "+v"(af[1][1]), "+v"(bf[0][0]), "+v"(bf[0][1]),           \
                 "+v"(bf[1][0]), "+v"(bf[1][1]), "+{" MX_RA0L "}"(a8l[0]), "+{" MX_RA0H "}"(a8h[0]), "+{" MX_RA1L "}"(a8l[1]), \
                 "+{" MX_RA1H "}"(a8h[1]), "+{" MX_RB0L "}"(b8l[0]), "+{" MX_RB0H "}"(b8h[0]), "+{" MX_RB1L "}"(b8l[1]),    \
                 "+{" MX_RB1H "}"(b8h[1]))
#define MX_LGKM_A()                                                                                                         \
  asm volatile("s_waitcnt lgkmcnt(0)"                                                                                       \
               : "+v"(af[0][0]), "+v"(af[0][1]), "+v"(af[1][0]), "+v"(af[1][1]), "+{" MX_RA0L "}"(a8l[0]),                 \
                 "+{" MX_RA0H "}"(a8h[0]), "+{" MX_RA1L "}"(a8l[1]), "+{" MX_RA1H "}"(a8h[1]))
  // one 32x32 tile: acc[T] += hi_b x hi_a (two fp16 steps) + [ql_b | q_b] x [q_a | ql_a] (one block-scaled step); a phase
  // issues the three steps tile-interleaved (four independent accumulators between two products into the same one)
#define MX_H0(T, TA, U) \
  acc[T] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, bf[U][0]), __builtin_bit_cast(f16x8, af[TA][0]), acc[T], 0, 0, 0)
#define MX_H1(T, TA, U) \
  acc[T] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, bf[U][1]), __builtin_bit_cast(f16x8, af[TA][1]), acc[T], 0, 0, 0)
#define MX_SC(T, TA, U, SA, SB)                                                                                             \
  acc[T] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(mx_cat(b8l[U], b8h[U]), mx_cat(a8l[TA], a8h[TA]), acc[T], 0, 0,  \
                                                           0, (int)(SB), 0, (int)(SA))
#define MX_PHASE(T0, S0, S1)                                                                          \
  MX_H0(T0, 0, 0); MX_H0(T0 + 1, 0, 1); MX_H0(T0 + 2, 1, 0); MX_H0(T0 + 3, 1, 1);                     \
  MX_H1(T0, 0, 0); MX_H1(T0 + 1, 0, 1); MX_H1(T0 + 2, 1, 0); MX_H1(T0 + 3, 1, 1);                     \
  MX_SC(T0, 0, 0, S0, sb_c[0]); MX_SC(T0 + 1, 0, 1, S0, sb_c[1]); MX_SC(T0 + 2, 1, 0, S1, sb_c[0]);   \
  MX_SC(T0 + 3, 1, 1, S1, sb_c[1])

#if defined(MX_STAMPS)
  const unsigned long long st_setup = __builtin_amdgcn_s_memtime();
#endif
  MX_VMCNT(0);  // (K-steps 0 and 1, and the scale bytes requested behind them)
  __builtin_amdgcn_s_barrier();
  if (wr == 1) __builtin_amdgcn_s_barrier();  // stagger: the second M-half runs one barrier behind
#if defined(MX_STAMPS)
  const unsigned long long st_loop0 = __builtin_amdgcn_s_memtime();
#endif

#if defined(MX_STAMPS)
  // instrumented builds only (tools/mx_phases.py): s_memtime ticks per section of the two-phase K-step
  unsigned long long st_t = 0, st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define MX_STAMP0() st_t = __builtin_amdgcn_s_memtime()
#define MX_STAMP(k)                                              \
  {                                                              \
    const unsigned long long now = __builtin_amdgcn_s_memtime(); \
    st_acc[k] += now - st_t;                                     \
    st_t = now;                                                  \
  }
#else
#define MX_STAMP0() (void)0
#define MX_STAMP(k) (void)0
#endif
  // ---- one K-step.  A rows of K-step kt sit in ring stage SA = kt % 3, B rows in buffer SB = kt % 2.  Requests run TWO
  // K-steps ahead (a piece has three to four phases to land; with one K-step ahead the loop waited ~300 of 2600 cycles per
  // K-step on pieces requested a phase earlier -- tools/mx_phases.py):
  //   phase A (kt): the four A passes of kt + 2 -> ring stage (SA + 2) % 3, last read in K-step kt - 1 (its second half one
  //                 phase ago: every fragment read is WAITED FOR in front of the barrier that ends its read section, so
  //                 whoever passes that barrier -- also the staggered group -- may overwrite the rows);
  //   phase B (kt): the four B passes of kt + 2 -> buffer SB, whose only reads (phase A of kt, fragments kept in registers
  //                 through phase B) ended one phase ago; then vmcnt(8) -> everything of K-step kt + 1 has landed.
  // The ring stage / buffer of a K-step is RUNTIME state (scalar byte offsets added to the eight per-lane read addresses at
  // the top of the K-step, 8 VALU instructions): with the stages as template constants the loop was six K-steps long (+ five
  // for the tail) and, in the conv form, ~100 KB of code.
  int offA = 0, offB = 0;  // ring stage of K-step kt (A: 0 / 32 KiB / 64 KiB), buffer (B: 0 / 32 KiB)
  auto kstep = [&](int kt) {
    constexpr int IA = 0, IB = 0, ST = BM * 128;
    const int NA = offA >= ST ? offA - ST : offA + 2 * ST;  // stage of K-step kt + 2 = (stage + 2) % 3
    const int SB = offB;
    const unsigned a0 = cA0 + (unsigned)offA, a1 = cA1 + (unsigned)offA, a2 = cA2 + (unsigned)offA, a3 = cA3 + (unsigned)offA;
    const unsigned b0 = cB0 + (unsigned)offB, b1 = cB1 + (unsigned)offB, b2 = cB2 + (unsigned)offB, b3 = cB3 + (unsigned)offB;
    const bool more2 = kt + 2 < nk;
    const int soa = soff_a(kt + 2, t2), sob = soff_b(kt + 2, t2);  // (scalar)
    const bool sec2 = CONV && t2.c0 >= p.Cin;
    // ---- phase A: A rows 0-63 (tiles 0, 1) x all 64 columns (tiles 0, 1) of this wavefront: 16 fragment reads
    MX_STAMP0();
#if !defined(MX_ABL_NOREAD)
    mx_read<IB + 0 * 4096>(bf[0][0], b0); mx_read<IB + 0 * 4096>(bf[0][1], b1);
    MX_READ_P(MX_RB0L, b8l[0], IB + 0 * 4096, b2); MX_READ_P(MX_RB0H, b8h[0], IB + 0 * 4096, b3);
    mx_read<IA + 0 * 4096>(af[0][0], a0); mx_read<IA + 0 * 4096>(af[0][1], a1);
    MX_READ_P(MX_RA0L, a8l[0], IA + 0 * 4096, a2); MX_READ_P(MX_RA0H, a8h[0], IA + 0 * 4096, a3);
    mx_read<IA + 1 * 4096>(af[1][0], a0); mx_read<IA + 1 * 4096>(af[1][1], a1);
    MX_READ_P(MX_RA1L, a8l[1], IA + 1 * 4096, a2); MX_READ_P(MX_RA1H, a8h[1], IA + 1 * 4096, a3);
    mx_read<IB + 1 * 4096>(bf[1][0], b0); mx_read<IB + 1 * 4096>(bf[1][1], b1);
    MX_READ_P(MX_RB1L, b8l[1], IB + 1 * 4096, b2); MX_READ_P(MX_RB1H, b8h[1], IB + 1 * 4096, b3);
#endif
    MX_STAMP(0);
#if !defined(MX_ABL_NODMA)
    if (more2) { dma_a(NA, 0, soa, sec2); dma_a(NA, 2, soa, sec2); dma_a(NA, 1, soa, sec2); dma_a(NA, 3, soa, sec2); }
#endif
    MX_STAMP(1);
    MX_LGKM_ALL();
    MX_STAMP(2);
    __builtin_amdgcn_s_barrier();
    MX_STAMP(3);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
    MX_PHASE(0, sa_c[0], sa_c[1]);
    // (the empty asm pins the products HERE: they are pure register operations for every pass before the scheduler, which
    // otherwise sinks them past the barriers into the next phase and keeps copies of their operands alive)
    asm volatile("" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]));
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    MX_STAMP(4);
    __builtin_amdgcn_s_barrier();
    MX_STAMP(5);
    // ---- phase B: A rows 64-127 (tiles 2, 3, into the same registers) x the same B fragments
#if !defined(MX_ABL_NOREAD)
    mx_read<IA + 2 * 4096>(af[0][0], a0); mx_read<IA + 2 * 4096>(af[0][1], a1);
    MX_READ_P(MX_RA0L, a8l[0], IA + 2 * 4096, a2); MX_READ_P(MX_RA0H, a8h[0], IA + 2 * 4096, a3);
    mx_read<IA + 3 * 4096>(af[1][0], a0); mx_read<IA + 3 * 4096>(af[1][1], a1);
    MX_READ_P(MX_RA1L, a8l[1], IA + 3 * 4096, a2); MX_READ_P(MX_RA1H, a8h[1], IA + 3 * 4096, a3);
#endif
    if (more2) {
#if !defined(MX_ABL_NODMA)
      dma_b(SB, 0, sob); dma_b(SB, 1, sob); dma_b(SB, 2, sob); dma_b(SB, 3, sob);
#endif
      if constexpr (CONV) {
        // the tap after next and its per-lane offsets (branch-free, ~20 VALU + ~25 scalar instructions), HERE in the read
        // section behind the DMA issue: interleaved with the products of this phase (gemm8.hip's place for them) they made
        // the 512-cycle product block of this format ~100 cycles longer (tools/mx_conv_ab.py: 7.80 -> 7.59 ms for the eight
        // convs); the empty asm keeps hipcc from sinking the selects to their use in front of the next K-step's DMA issue
        t2 = tap_next(t2);
        conv_va(t2);
        asm volatile("" : "+v"(va[0]), "+v"(va[1]), "+v"(va[2]), "+v"(va[3]));
      }
      MX_VMCNT(8);  // younger: the eight pieces of K-step kt + 2 -> K-step kt + 1 has landed
    } else {
      MX_VMCNT(0);
    }
    MX_LGKM_A();
    MX_STAMP(6);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
    MX_PHASE(4, sa_c[2], sa_c[3]);
    asm volatile("" : "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]), "+v"(acc[7]));
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    MX_STAMP(7);
    offA = offA >= 2 * ST ? 0 : offA + ST;
    offB ^= ST;
  };
  // (the block scales of an operand change at segment boundaries only: launcher -- whole rows, or multiples of 6 K-steps)
  int next_a = seg_a, next_b = seg_b;
#pragma unroll 1
  for (int kt = 0; kt < nk; ++kt) {
    if (kt == next_a) {  // (the empty asm: the bytes are waited for HERE, not in front of the first product)
      load_scales_a(kt / seg_a); next_a += seg_a;
      asm volatile("" : "+v"(sa_c[0]), "+v"(sa_c[1]), "+v"(sa_c[2]), "+v"(sa_c[3]));
    }
    if (kt == next_b) {
      load_scales_b(kt / seg_b); next_b += seg_b;
      asm volatile("" : "+v"(sb_c[0]), "+v"(sb_c[1]));
    }
    kstep(kt);
  }
  if (wr == 0) __builtin_amdgcn_s_barrier();  // balance the stagger barrier
#if defined(MX_STAMPS)
  const unsigned long long st_loop1 = __builtin_amdgcn_s_memtime();
#endif
#undef MX_PHASE
#undef MX_H0
#undef MX_H1
#undef MX_SC
#undef MX_READ_P
#undef MX_LGKM_ALL
#undef MX_LGKM_A
#undef MX_VMCNT

  // ---- epilogue: tile T = 2 t + u holds row m0 + wr*128 + 32 t + r32 and, in registers 4 g .. 4 g + 3, the columns
  // n0 + wc*64 + 32 u + 16 h + 4 g ..  (the B-row permutation above): 16 consecutive columns per lane, tile and row
  if (p.ksplit > 1) {  // split-K: raw partial sums of this K slice (rows counted from m_base)
    float* part = p.partial + ((long long)kslice * (p.M - p.m_base) - p.m_base) * p.partial_ld;
    auto part_tile = [&](const f32x16& a, const int T) {
      const int m = m0 + wr * 128 + 32 * (T >> 1) + r32;
      const int nb0 = n0 + wc * 64 + 32 * (T & 1) + 16 * hh;
      if (m >= p.M) return;
#pragma unroll
      for (int g = 0; g < 4; ++g)
        if (nb0 + 4 * g < p.N)
          *(f32x4*)(part + (long long)m * p.partial_ld + nb0 + 4 * g) = f32x4{a[4 * g], a[4 * g + 1], a[4 * g + 2], a[4 * g + 3]};
    };
    part_tile(acc[0], 0); part_tile(acc[1], 1); part_tile(acc[2], 2); part_tile(acc[3], 3);
    part_tile(acc[4], 4); part_tile(acc[5], 5); part_tile(acc[6], 6); part_tile(acc[7], 7);
    return;
  }
  const float keep_scale = p.dropout_p > 0.f ? 1.0f / (1.0f - p.dropout_p) : 1.0f;
  const unsigned long long dseed = p.dropout_p > 0.f ? WS_DROPOUT_SEED(p) : 0ull;
  const unsigned dthr = dropout_threshold(p.dropout_p);
  const float lo = p.relu ? 0.f : -__builtin_inff();
  // (launcher: alpha / bias / residual / ReLU / dropout epilogue, vector-aligned rows, N a multiple of 4 -- of 16 for f16mx)
  // Loads FIRST: the output may alias anything as far as hipcc knows, so a load behind a store waits for nothing but is never
  // hoisted above it -- eight tiles of [bias load, residual load, compute, store] were eight memory round trips in a row
  // (~30 us of a 120-us res4 tile).  The bias quads of this lane's 2 x 16 columns are loaded once, an f16mx residual (the conv
  // chain's) for four tiles at a time, before the first store of those tiles.
#if defined(MX_ABL_NOEPI)
  if (p.M > 0) return;  // (timing ablation: the tile without its epilogue)
#endif
  asm volatile("" ::: "memory");  // (the epilogue's loads stay behind the K loop: hoisted above it they would live through it)
  f32x4 bv[2][4];
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int nb = n0 + wc * 64 + 32 * u + 16 * hh + 4 * g;
      bv[u][g] = (p.bias && nb < p.N) ? *(const f32x4*)(p.bias + nb) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  // (the specialised conv forms: no dropout, no bf16 copy -- the launcher sends anything else to EPI = 0)
  const bool drop = (CONV && EPI != 0) ? false : p.dropout_p > 0.f;
  void* const c_bf16 = (CONV && EPI != 0) ? nullptr : q.c_bf16;
  const int dtype_c = EPI == 1 || EPI == 2 ? (int)WSOVOD_F16MX : EPI == 3 ? (int)WSOVOD_F32 : EPI == 4 ? (int)WSOVOD_BF16X2 : p.dtype_c;
  const bool has_res = EPI == 1 || EPI == 4 ? false : EPI == 2 || EPI == 3 ? true : p.residual != nullptr;
  const bool res_mx = EPI == 2 || EPI == 3 ? true : EPI == 0 ? (p.residual && p.dtype_r == WSOVOD_F16MX) : false;
  // 4-byte-per-value outputs of whole 64-column wavefront blocks leave through a row image in LDS (free behind the K loop:
  // every fragment read was waited for in front of the last barrier each wavefront passed)
  constexpr int IMG_LD = 256 + 16;  // (+16: the 8 rows of a ds_write_b128 lane group fall on different banks)
  const bool via_lds = EPI != 0 || (p.N % 64 == 0 && (dtype_c == WSOVOD_F16MX || dtype_c == WSOVOD_BF16X2 || dtype_c == WSOVOD_F32));
  char* img = smem + wave_u * (32 * IMG_LD);
  auto emit_tile = [&](const f32x16& a, auto T_c, const f16x8 rh0, const f16x8 rh1, const i32x4 rl) {
    constexpr int T = decltype(T_c)::value;  // (a compile-time tile index: a runtime one would index bv[] in scratch)
    const int m = m0 + wr * 128 + 32 * (T >> 1) + r32;
    const int nb0 = n0 + wc * 64 + 32 * (T & 1) + 16 * hh;
    if (!via_lds && (m >= p.M || nb0 >= p.N)) return;
    const int mc = min(m, p.M - 1);  // (staged form: every lane fills its slot of the row image; rows past M are not stored)
    f32x4 y[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int nb = nb0 + 4 * g;
      f32x4 x = f32x4{a[4 * g], a[4 * g + 1], a[4 * g + 2], a[4 * g + 3]} * p.alpha;
      if (nb < p.N) {
        x += bv[T & 1][g];
        if (res_mx) {
          const f16x8 h = g < 2 ? rh0 : rh1;
          const int o = 4 * (g & 1);
          x += mx_dec4_unit(f16x4{h[o], h[o + 1], h[o + 2], h[o + 3]}, rl[g]);
        } else if (has_res) {
          x += load4_as_f32(p.residual, mc, p.ldr, nb, p.dtype_r);
        }
        x = f32x4{fmaxf(x[0], lo), fmaxf(x[1], lo), fmaxf(x[2], lo), fmaxf(x[3], lo)};
        if (drop) {
          const unsigned long long dz = dropout_quad(dseed, mc, p.N, nb);
#pragma unroll
          for (int r = 0; r < 4; ++r) x[r] = dropout_keep(dz, r, dthr) ? x[r] * keep_scale : 0.f;
        }
      }
      y[g] = x;
    }
    if (via_lds) {
      // the lane's 16 values as the bytes of the output format, into the wavefront's row image [32 rows][256 B + 16]: the
      // stores then leave as whole 256-byte row segments (emit_pair) instead of 64 separate 16-byte pieces per instruction
      char* slot = img + r32 * IMG_LD + (T & 1) * 128;
      if (dtype_c == WSOVOD_F16MX) {
        f16x4 h[4];
        int qv[4], lv[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) mx_enc4_unit(y[g], h[g], qv[g], lv[g]);
        *(f16x8*)(slot + 32 * hh) = f16x8{h[0][0], h[0][1], h[0][2], h[0][3], h[1][0], h[1][1], h[1][2], h[1][3]};
        *(f16x8*)(slot + 32 * hh + 16) = f16x8{h[2][0], h[2][1], h[2][2], h[2][3], h[3][0], h[3][1], h[3][2], h[3][3]};
        *(i32x4*)(slot + 64 + 16 * hh) = i32x4{qv[0], qv[1], qv[2], qv[3]};
        *(i32x4*)(slot + 96 + 16 * hh) = i32x4{lv[0], lv[1], lv[2], lv[3]};
      } else if (dtype_c == WSOVOD_BF16X2) {
        bf16x8 hi0, hi1, lo0, lo1;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float v0 = y[e >> 2][e & 3], v1 = y[2 + (e >> 2)][e & 3];
          hi0[e] = (bf16_t)v0; hi1[e] = (bf16_t)v1;
          lo0[e] = x2_lo(v0, hi0[e]); lo1[e] = x2_lo(v1, hi1[e]);
        }
        *(bf16x8*)(slot + 32 * hh) = hi0; *(bf16x8*)(slot + 32 * hh + 16) = hi1;
        *(bf16x8*)(slot + 64 + 32 * hh) = lo0; *(bf16x8*)(slot + 64 + 32 * hh + 16) = lo1;
      } else {
#pragma unroll
        for (int g = 0; g < 4; ++g) *(f32x4*)(slot + 64 * hh + 16 * g) = y[g];
      }
    } else if (dtype_c == WSOVOD_F16MX) {  // 16 values = half a group: 32 B of hi, 16 B of q, 16 B of ql
      f16x4 h[4];
      int qv[4], lv[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) mx_enc4_unit(y[g], h[g], qv[g], lv[g]);
      char* grp = (char*)p.C + (long long)m * p.ldc * 4 + mx_group(nb0);
      const int w = nb0 & 31;
      *(f16x8*)(grp + 2 * w) = f16x8{h[0][0], h[0][1], h[0][2], h[0][3], h[1][0], h[1][1], h[1][2], h[1][3]};
      *(f16x8*)(grp + 2 * w + 16) = f16x8{h[2][0], h[2][1], h[2][2], h[2][3], h[3][0], h[3][1], h[3][2], h[3][3]};
      *(i32x4*)(grp + 64 + w) = i32x4{qv[0], qv[1], qv[2], qv[3]};
      *(i32x4*)(grp + 96 + w) = i32x4{lv[0], lv[1], lv[2], lv[3]};
    } else {
#pragma unroll
      for (int g = 0; g < 4; ++g)
        if (nb0 + 4 * g < p.N) store4_from_f32(p.C, m, p.ldc, nb0 + 4 * g, dtype_c, y[g]);
    }
    if (c_bf16 && m < p.M && nb0 < p.N) {
      bf16_t* cb = (bf16_t*)c_bf16 + (long long)m * q.ld_cb + nb0;
      if (nb0 + 16 <= p.N) {
        *(bf16x8*)cb = bf16x8{(bf16_t)y[0][0], (bf16_t)y[0][1], (bf16_t)y[0][2], (bf16_t)y[0][3],
                              (bf16_t)y[1][0], (bf16_t)y[1][1], (bf16_t)y[1][2], (bf16_t)y[1][3]};
        *(bf16x8*)(cb + 8) = bf16x8{(bf16_t)y[2][0], (bf16_t)y[2][1], (bf16_t)y[2][2], (bf16_t)y[2][3],
                                    (bf16_t)y[3][0], (bf16_t)y[3][1], (bf16_t)y[3][2], (bf16_t)y[3][3]};
      } else {
#pragma unroll
        for (int g = 0; g < 4; ++g)
          if (nb0 + 4 * g < p.N) *(bf16x4*)(cb + 4 * g) = bf16x4{(bf16_t)y[g][0], (bf16_t)y[g][1], (bf16_t)y[g][2], (bf16_t)y[g][3]};
      }
    }
  };
  // the f16mx residual pieces of TWO tiles at a time (one row, both column halves: 24 registers next to the accumulators)
  auto emit_pair = [&](auto T0_c, const f32x16& a0, const f32x16& a1) {
    constexpr int T0 = decltype(T0_c)::value;
    f16x8 rh0[2], rh1[2];
    i32x4 rl[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      rh0[k] = rh1[k] = f16x8{0, 0, 0, 0, 0, 0, 0, 0};
      rl[k] = i32x4{0, 0, 0, 0};
      const int T = T0 + k;
      const int m = m0 + wr * 128 + 32 * (T >> 1) + r32;
      const int nb0 = n0 + wc * 64 + 32 * (T & 1) + 16 * hh;
      if (res_mx && m < p.M && nb0 < p.N) {  // (launcher: N a multiple of 16 with an f16mx residual)
        const char* grp = (const char*)p.residual + (long long)m * p.ldr * 4 + mx_group(nb0);
        const int w = nb0 & 31;
        rh0[k] = *(const f16x8*)(grp + 2 * w);
        rh1[k] = *(const f16x8*)(grp + 2 * w + 16);
        rl[k] = *(const i32x4*)(grp + 96 + w);
      }
    }
    emit_tile(a0, std::integral_constant<int, T0>{}, rh0[0], rh1[0], rl[0]);
    emit_tile(a1, std::integral_constant<int, T0 + 1>{}, rh0[1], rh1[1], rl[1]);
    if (via_lds) {  // 32 rows x 256 B: an instruction stores four whole row segments (16 lanes x 16 B each)
      const int mrow = m0 + wr * 128 + 32 * (T0 >> 1);
      char* cbase = (char*)p.C + ((long long)(n0 + wc * 64) << 2) + (lane & 15) * 16;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int row = 4 * j + (lane >> 4);
        const u32x4 v = *(const u32x4*)(img + row * IMG_LD + (lane & 15) * 16);
        if (mrow + row < p.M && n0 + wc * 64 < p.N) *(u32x4*)(cbase + (long long)(mrow + row) * p.ldc * 4) = v;
      }
    }
  };
  emit_pair(std::integral_constant<int, 0>{}, acc[0], acc[1]);
  emit_pair(std::integral_constant<int, 2>{}, acc[2], acc[3]);
  emit_pair(std::integral_constant<int, 4>{}, acc[4], acc[5]);
  emit_pair(std::integral_constant<int, 6>{}, acc[6], acc[7]);
#if defined(MX_STAMPS)
  {
    const unsigned long long st_e0 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long st_e1 = __builtin_amdgcn_s_memtime();
    if (q.dbg && lane == 0) {  // (all atomics behind the last stamp: 512 workgroups adding to the same floats take ~50 us)
#pragma unroll
      for (int k = 0; k < 8; ++k) atomicAdd(q.dbg + wr * 16 + k, (float)st_acc[k]);
      atomicAdd(q.dbg + wr * 16 + 8, (float)nk);
    }
    if (q.dbg && lane == 0 && (wave == 0 || wave == 4)) {
      atomicAdd(q.dbg + wr * 16 + 9, (float)(st_setup - st_begin));
      atomicAdd(q.dbg + wr * 16 + 12, (float)(st_loop0 - st_setup));
      atomicAdd(q.dbg + wr * 16 + 10, (float)(st_loop1 - st_loop0));
      atomicAdd(q.dbg + wr * 16 + 13, (float)(st_e0 - st_loop1));
      atomicAdd(q.dbg + wr * 16 + 14, (float)(st_e1 - st_e0));
      atomicAdd(q.dbg + wr * 16 + 11, 1.0f);
    }
  }
#endif
#endif
}

// split-K finalize: C[m][n] = epilogue(sum over the K slices) for 4 consecutive columns per thread -- the same chain and the
// same output formats as the tile kernel's epilogue
__global__ __launch_bounds__(256) void mx_splitk_finalize_kernel(const MxArgs q) {
#if defined(__HIP_DEVICE_COMPILE__)
  const GemmArgs& p = q.g;
  const int n4 = p.N >> 2;  // (launcher: N a multiple of 4)
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  const int rows = p.M - p.m_base;
  if (idx >= (long long)rows * n4) return;
  const int mr = (int)(idx / n4), nb = (int)(idx - (long long)mr * n4) * 4, m = p.m_base + mr;
  f32x4 x = {0.f, 0.f, 0.f, 0.f};
  for (int z = 0; z < p.ksplit; ++z) x += *(const f32x4*)(p.partial + ((long long)z * rows + mr) * p.partial_ld + nb);
  x = x * p.alpha;
  if (p.bias) x += *(const f32x4*)(p.bias + nb);
  if (p.residual) {
    if (p.dtype_r == WSOVOD_F16MX) x += mx_load4_unit((const char*)p.residual + (long long)m * p.ldr * 4, nb);
    else x += load4_as_f32(p.residual, m, p.ldr, nb, p.dtype_r);
  }
  const float lo = p.relu ? 0.f : -__builtin_inff();
  x = f32x4{fmaxf(x[0], lo), fmaxf(x[1], lo), fmaxf(x[2], lo), fmaxf(x[3], lo)};
  if (p.dropout_p > 0.f) {
    const float keep_scale = 1.0f / (1.0f - p.dropout_p);
    const unsigned long long dz = dropout_quad(WS_DROPOUT_SEED(p), m, p.N, nb);
    const unsigned dthr = dropout_threshold(p.dropout_p);
#pragma unroll
    for (int r = 0; r < 4; ++r) x[r] = dropout_keep(dz, r, dthr) ? x[r] * keep_scale : 0.f;
  }
  if (p.dtype_c == WSOVOD_F16MX) mx_store4_unit((char*)p.C + (long long)m * p.ldc * 4, nb, x);
  else store4_from_f32(p.C, m, p.ldc, nb, p.dtype_c, x);
  if (q.c_bf16)
    *(bf16x4*)((bf16_t*)q.c_bf16 + (long long)m * q.ld_cb + nb) = bf16x4{(bf16_t)x[0], (bf16_t)x[1], (bf16_t)x[2], (bf16_t)x[3]};
#endif
}

// ---- plane conversions between the two parity formats (the maps that cross from the bf16x2 layers to the f16mx ones)
__global__ __launch_bounds__(256) void mx_from_x2_kernel(const bf16_t* __restrict__ src, char* __restrict__ dst, long long ngroups) {
#if defined(__HIP_DEVICE_COMPILE__)
  // a thread per 8 values of a 32-value group: bf16x2 group = [32 hi | 32 lo] bf16, same 128 bytes as the f16mx group
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long grp = t >> 2;
  if (grp >= ngroups) return;
  const int w = (int)(t & 3) * 8;
  const bf16_t* s = src + grp * 64 + w;
  const bf16x8 h = *(const bf16x8*)s, l = *(const bf16x8*)(s + 32);
  f16x4 h0, h1;
  int q0, q1, l0, l1;
  mx_enc4_unit(f32x4{(float)h[0] + (float)l[0], (float)h[1] + (float)l[1], (float)h[2] + (float)l[2], (float)h[3] + (float)l[3]}, h0, q0, l0);
  mx_enc4_unit(f32x4{(float)h[4] + (float)l[4], (float)h[5] + (float)l[5], (float)h[6] + (float)l[6], (float)h[7] + (float)l[7]}, h1, q1, l1);
  char* d = dst + grp * 128;
  *(f16x8*)(d + 2 * w) = f16x8{h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
  *(i32x2*)(d + 64 + w) = i32x2{q0, q1};
  *(i32x2*)(d + 96 + w) = i32x2{l0, l1};
#endif
}

// ---- encoder: fp32 (rows, cols) -> f16mx carrier + one scale byte per (row, segment); a workgroup per (row, segment):
// pass 1 the segment's largest |fp16(x)| (exponent field), pass 2 the planes.  scales == NULL: the unit-scale form
__global__ __launch_bounds__(256) void mx_encode_kernel(const float* __restrict__ src, long long ld_src, int rows, int cols,
                                                        int nseg, unsigned char* __restrict__ dst, long long ld_dst_bytes,
                                                        unsigned char* __restrict__ scales,
                                                        const unsigned char* __restrict__ tscale = nullptr) {
#if defined(__HIP_DEVICE_COMPILE__)
  __shared__ unsigned red[4];
  const int r = blockIdx.x / nseg, sg = blockIdx.x - r * nseg;
  const int seg_cols = cols / nseg, c0 = sg * seg_cols;
  const float* s = src + (long long)r * ld_src + c0;
  int sexp = 0;  // q scale 2^sexp; ql scale 2^(sexp - 11)
  if (tscale) {  // one scale for the whole tensor, chosen by the caller
    sexp = (int)*tscale - 127;
    if (threadIdx.x == 0) scales[(long long)r * nseg + sg] = *tscale;
  } else if (scales) {
    unsigned mx = 0;
    for (int e = threadIdx.x * 4; e < seg_cols; e += 256 * 4) {
      const f32x4 t = *(const f32x4*)(s + e);
#pragma unroll
      for (int j = 0; j < 4; ++j) mx = max(mx, (unsigned)(__builtin_bit_cast(unsigned short, (_Float16)t[j]) & 0x7fffu));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = max(mx, (unsigned)__shfl_xor((int)mx, o));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
    __syncthreads();
    mx = max(max(red[0], red[1]), max(red[2], red[3]));
    const int ef = min((int)(mx >> 10), 30);  // fp16 exponent field of the largest |hi| (31 = inf / nan: treated as 30)
    sexp = (ef == 0 ? -14 : ef - 15) - 7;
    if (threadIdx.x == 0) scales[(long long)r * nseg + sg] = (unsigned char)(sexp + 127);
  }
  const float inv_q = __builtin_ldexpf(1.0f, -sexp), inv_l = __builtin_ldexpf(1.0f, -(sexp - 11));
  // one thread per 8 values: 16 B of hi, 8 B of q, 8 B of ql
  unsigned char* drow = dst + (long long)r * ld_dst_bytes;
  for (int e = threadIdx.x * 8; e < seg_cols; e += 256 * 8) {
    const int c = c0 + e;
    f16x4 h0, h1;
    int q0, q1, l0, l1;
    mx_enc4(*(const f32x4*)(s + e), inv_q, inv_l, h0, q0, l0);
    mx_enc4(*(const f32x4*)(s + e + 4), inv_q, inv_l, h1, q1, l1);
    unsigned char* d = drow + mx_group(c);
    const int w = c & 31;  // position inside the group of 32 (a multiple of 8)
    *(f16x8*)(d + 2 * w) = f16x8{h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
    *(i32x2*)(d + 64 + w) = i32x2{q0, q1};
    *(i32x2*)(d + 96 + w) = i32x2{l0, l1};
  }
#endif
}

}  // namespace
}  // namespace wsovod_gemm

using namespace wsovod_gemm;

extern "C" int wsovod_f16mx_encode(const float* src, long long ld_src, int rows, int cols, int nseg, void* dst, long long ld_dst,
                                   unsigned char* scales, wsovod_stream_t stream) {
  WS_CHECK_ARG(rows >= 0 && cols >= 0 && nseg >= 1 && cols % (32 * nseg) == 0,
               "wsovod_f16mx_encode: cols=%d must be a multiple of 32 * nseg (%d)", cols, nseg);
  if (rows == 0 || cols == 0) return WSOVOD_OK;
  WS_CHECK_ARG(src && dst && ld_src >= cols && ld_src % 4 == 0 && ld_dst >= cols && ld_dst % 4 == 0 &&
                   (((uintptr_t)dst | (uintptr_t)src) & 15) == 0,
               "wsovod_f16mx_encode: bad pointer / leading dimension");
  static int slot = wsovod::prof_slot("f16mx_encode");
  hipStream_t s = (hipStream_t)stream;
  wsovod::ProfScope prof(slot, s, 0.0, (double)rows * cols * 8.0);
  hipLaunchKernelGGL(mx_encode_kernel, dim3((unsigned)((long long)rows * nseg)), dim3(256), 0, s, src, ld_src, rows, cols, nseg,
                     (unsigned char*)dst, ld_dst * 4, scales, (const unsigned char*)nullptr);
  WS_CHECK_LAUNCH("wsovod_f16mx_encode");
  return WSOVOD_OK;
}

extern "C" int wsovod_f16mx_encode_with(const float* src, long long ld_src, int rows, int cols, void* dst, long long ld_dst,
                                        unsigned char* scales, const unsigned char* tensor_scale, wsovod_stream_t stream) {
  WS_CHECK_ARG(rows >= 0 && cols >= 0 && cols % 32 == 0, "wsovod_f16mx_encode_with: cols=%d must be a multiple of 32", cols);
  if (rows == 0 || cols == 0) return WSOVOD_OK;
  WS_CHECK_ARG(src && dst && scales && tensor_scale && ld_src >= cols && ld_src % 4 == 0 && ld_dst >= cols && ld_dst % 4 == 0 &&
                   (((uintptr_t)dst | (uintptr_t)src) & 15) == 0,
               "wsovod_f16mx_encode_with: bad pointer / leading dimension");
  static int slot = wsovod::prof_slot("f16mx_encode");
  hipStream_t s = (hipStream_t)stream;
  wsovod::ProfScope prof(slot, s, 0.0, (double)rows * cols * 8.0);
  hipLaunchKernelGGL(mx_encode_kernel, dim3((unsigned)rows), dim3(256), 0, s, src, ld_src, rows, cols, 1, (unsigned char*)dst,
                     ld_dst * 4, scales, tensor_scale);
  WS_CHECK_LAUNCH("wsovod_f16mx_encode_with");
  return WSOVOD_OK;
}

extern "C" int wsovod_f16mx_from_bf16x2(const void* src, void* dst, long long n, wsovod_stream_t stream) {
  WS_CHECK_ARG(n >= 0 && n % 32 == 0, "wsovod_f16mx_from_bf16x2: n=%lld must be whole 32-value groups", n);
  if (n == 0) return WSOVOD_OK;
  WS_CHECK_ARG(src && dst && (((uintptr_t)dst | (uintptr_t)src) & 15) == 0, "wsovod_f16mx_from_bf16x2: bad pointer");
  static int slot = wsovod::prof_slot("f16mx_from_bf16x2");
  hipStream_t s = (hipStream_t)stream;
  wsovod::ProfScope prof(slot, s, 0.0, (double)n * 8.0);
  const long long threads = n / 8;
  hipLaunchKernelGGL(mx_from_x2_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, (const bf16_t*)src, (char*)dst, n / 32);
  WS_CHECK_LAUNCH("wsovod_f16mx_from_bf16x2");
  return WSOVOD_OK;
}

extern "C" int wsovod_gemm_f16mx(const wsovod_gemm_desc* d, const unsigned char* a_scale, int a_segments,
                                 const unsigned char* b_scale, int b_segments, void* c_bf16, long long ld_c_bf16,
                                 wsovod_stream_t stream) {
  WS_CHECK_ARG(d && b_scale, "wsovod_gemm_f16mx: null descriptor / weight scale array");
  WS_CHECK_ARG(d->a_plane_bytes == 0 && (d->conv || !d->A2), "wsovod_gemm_f16mx: interleaved operands; A2 is the conv form's shortcut input");
  WS_CHECK_ARG(d->M >= 0 && d->N >= 0 && d->K > 0 && d->K % 32 == 0 && a_segments >= 1 && b_segments >= 1 &&
                   (d->K / 32) % a_segments == 0 && (d->K / 32) % b_segments == 0,
               "wsovod_gemm_f16mx: K=%d must be whole 32-value groups, split evenly into the operands' scale segments", d->K);
  {
    const int sa = d->K / 32 / a_segments, sb = d->K / 32 / b_segments;
    WS_CHECK_ARG((a_segments == 1 || sa % 6 == 0) && (b_segments == 1 || sb % 6 == 0),
                 "wsovod_gemm_f16mx: a scale segment (%d / %d groups of 32) must be the whole row or a multiple of 6 groups", sa, sb);
  }
  if (d->M == 0 || d->N == 0) return WSOVOD_OK;
  WS_CHECK_ARG(d->A && d->B, "wsovod_gemm_f16mx: null pointer");
  WS_CHECK_ARG(d->ldb % 4 == 0 && d->ldb >= d->K && (((uintptr_t)d->A | (uintptr_t)d->B) & 15) == 0 &&
                   320ll * d->ldb * 4 < (1ll << 31),
               "wsovod_gemm_f16mx: operands must be 16-byte aligned f16mx rows (a tile of weight rows below 2 GiB)");
  WS_CHECK_ARG(d->dropout_p >= 0.f && d->dropout_p < 1.f, "wsovod_gemm_f16mx: dropout_p must be in [0,1)");
  WS_CHECK_ARG(d->C && !d->Ct && !d->row_scale && !d->group_add && !d->mask_src && !d->accumulate,
               "wsovod_gemm_f16mx: the epilogue is alpha / bias / residual / ReLU / dropout");
  WS_CHECK_ARG(d->N % 4 == 0 && (!d->bias || ((uintptr_t)d->bias & 15) == 0) &&
                   (d->dtype_c == WSOVOD_BF16X2 || d->dtype_c == WSOVOD_F16MX
                        ? (d->ldc % 32 == 0 && ((uintptr_t)d->C & 15) == 0 && (d->dtype_c == WSOVOD_BF16X2 || d->N % 16 == 0))
                    : d->dtype_c == WSOVOD_BF16 ? (d->ldc % 4 == 0 && ((uintptr_t)d->C & 7) == 0)
                                                : (d->dtype_c == WSOVOD_F32 && d->ldc % 4 == 0 && ((uintptr_t)d->C & 15) == 0)),
               "wsovod_gemm_f16mx: N must be a multiple of 4 (16 for an f16mx output) and the output rows vector-aligned "
               "(bf16x2 / f16mx: whole 32-value groups)");
  WS_CHECK_ARG(!c_bf16 || (ld_c_bf16 % 8 == 0 && ld_c_bf16 >= d->N && ((uintptr_t)c_bf16 & 15) == 0),
               "wsovod_gemm_f16mx: the bf16 copy needs 16-byte aligned rows");
  WS_CHECK_ARG(!d->residual || (d->dtype_r == WSOVOD_F16MX || d->dtype_r == WSOVOD_BF16X2
                                    ? (d->ldr % 32 == 0 && ((uintptr_t)d->residual & 15) == 0)
                                : d->dtype_r == WSOVOD_BF16 ? (d->ldr % 4 == 0 && ((uintptr_t)d->residual & 7) == 0)
                                                            : (d->dtype_r == WSOVOD_F32 && d->ldr % 4 == 0 && ((uintptr_t)d->residual & 15) == 0)),
               "wsovod_gemm_f16mx: residual rows must be vector-aligned");
  WS_CHECK_ARG(!d->residual || d->dtype_r != WSOVOD_F16MX || d->N % 16 == 0, "wsovod_gemm_f16mx: an f16mx residual needs N a multiple of 16");

  MxArgs q;
  memset(&q, 0, sizeof(q));
  GemmArgs& a = q.g;
  a.A = (const char*)d->A;
  a.B = (const char*)d->B;
  a.lda = d->lda * 2;  // counted in 2-byte slots, as the bf16x2 form
  a.ldb = d->ldb * 2;
  a.M = d->M;
  a.N = d->N;
  a.K = d->K * 2;
  a.C = d->C;
  a.ldc = d->ldc;
  a.dtype_c = d->dtype_c;
  a.alpha = d->alpha;
  a.bias = d->bias;
  a.residual = d->residual;
  a.ldr = d->ldr;
  a.dtype_r = d->dtype_r;
  a.relu = d->relu;
  a.dropout_p = d->dropout_p;
  a.seed = d->dropout_seed;
  a.seed_add = d->dropout_seed_add;
  double bytes;
  if (d->conv) {
    const wsovod_conv_geom& g = d->geom;
    WS_CHECK_ARG(!a_scale, "wsovod_gemm_f16mx(conv): the input map is a unit-scale f16mx tensor (a_scale = NULL)");
    WS_CHECK_ARG(g.Cin > 0 && g.Cin % 32 == 0 && !g.pool, "wsovod_gemm_f16mx(conv): Cin=%d must be a multiple of 32 (no fused pool)", g.Cin);
    WS_CHECK_ARG(!d->A2 || (d->Cin2 > 0 && d->Cin2 % 32 == 0 && ((uintptr_t)d->A2 & 15) == 0),
                 "wsovod_gemm_f16mx(conv): the fused shortcut input needs Cin2 (%d) a multiple of 32 and 16-byte alignment", d->Cin2);
    WS_CHECK_ARG(d->K == g.KH * g.KW * g.Cin + (d->A2 ? d->Cin2 : 0), "wsovod_gemm_f16mx(conv): K=%d != KH*KW*Cin (+ Cin2)", d->K);
    WS_CHECK_ARG(g.KH * g.KW <= 32, "wsovod_gemm_f16mx(conv): filters of more than 32 taps are not supported (per-tap validity mask)");
    WS_CHECK_ARG((long long)d->M == (long long)g.n_img * g.Ho * g.Wo, "wsovod_gemm_f16mx(conv): M=%d != n_img*Ho*Wo", d->M);
    WS_CHECK_ARG(g.stride >= 1 && g.dil >= 1 && g.pad >= 0, "wsovod_gemm_f16mx(conv): bad stride/dil/pad");
    a.H = g.H; a.W = g.W; a.Cin = g.Cin * 2; a.Ho = g.Ho; a.Wo = g.Wo;
    a.KH = g.KH; a.KW = g.KW; a.stride = g.stride; a.pad = g.pad; a.dil = g.dil;
    a.a_bytes = (long long)g.n_img * g.H * g.W * g.Cin * 4;
    if (d->A2) {
      a.A2 = (const char*)d->A2;
      a.Cin2 = d->Cin2 * 2;
      a.a2_bytes = (long long)g.n_img * g.Ho * g.Wo * d->Cin2 * 4;
      WS_CHECK_ARG(a.a2_bytes < (1ll << 31), "wsovod_gemm_f16mx(conv): fused shortcut input exceeds the 2 GiB buffer-addressing limit");
    }
    WS_CHECK_ARG(a.a_bytes + (long long)(g.pad * g.W + g.pad) * g.Cin * 4 < (1ll << 31),
                 "wsovod_gemm_f16mx(conv): input of %lld bytes exceeds the 2 GiB buffer-addressing limit", a.a_bytes);
    bytes = ((double)g.n_img * g.H * g.W * g.Cin + (double)d->N * d->K + (d->A2 ? (double)d->M * d->Cin2 : 0.0)) * 4.0;
  } else {
    WS_CHECK_ARG(a_segments == 1 || a_scale, "wsovod_gemm_f16mx: scale segments without a scale array");
    WS_CHECK_ARG(d->lda % 4 == 0 && d->lda >= d->K && 320ll * d->lda * 4 < (1ll << 31),
                 "wsovod_gemm_f16mx: lda=%lld: f16mx rows of at least K values, a tile of rows below 2 GiB", d->lda);
    bytes = ((double)d->M + d->N) * d->K * 4.0;
  }
  bytes += (double)d->M * d->N * ((d->dtype_c == WSOVOD_BF16 ? 2 : 4) + (c_bf16 ? 2 : 0) + (d->residual ? (d->dtype_r == WSOVOD_BF16 ? 2 : 4) : 0));
  a.tiles_m = (d->M + 255) / 256;
  a.tiles_n = (d->N + 255) / 256;
  {
    const int run = std::max(1, a.tiles_m * a.tiles_n / 8);
    int g = 1;
    while ((g + 1) * (g + 1) <= run) ++g;
    a.group_m = std::max(1, std::min(std::min(g, 4), a.tiles_m));
  }
  q.sa = d->conv ? nullptr : a_scale;
  q.sb = b_scale;
  q.nseg_a = a_scale ? a_segments : 1;
  q.nseg_b = b_segments;
  q.c_bf16 = c_bf16;
  q.ld_cb = ld_c_bf16;
#if defined(MX_STAMPS)
  if (const char* e = getenv("WSOVOD_MX_DEBUG_PTR")) q.dbg = (float*)strtoull(e, nullptr, 16);
#endif
  static int slot_g = wsovod::prof_slot("gemm_nt_f16mx_256x256_8ph"), slot_c = wsovod::prof_slot("conv_igemm_f16mx_256x256_8ph");
  static bool attr_set = false;
  constexpr int lds_bytes = (3 * 256 + 2 * 256) * 128;  // 160 KiB: the whole CU
  if (!attr_set) {
#define MX_OPT_IN(C, E)                                                                                                  \
  WS_CHECK_HIP(hipFuncSetAttribute((const void*)gemm256_mx_kernel<C, E>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes), \
               "wsovod_gemm_f16mx: LDS opt-in")
    MX_OPT_IN(false, 0); MX_OPT_IN(false, 1); MX_OPT_IN(false, 4);
    MX_OPT_IN(true, 0); MX_OPT_IN(true, 1); MX_OPT_IN(true, 2); MX_OPT_IN(true, 3);
#undef MX_OPT_IN
    attr_set = true;
  }
  hipStream_t s = (hipStream_t)stream;
  // ---- the last, partly filled round of tiles.  One 256 x 256 tile per CU and launch round: a launch of T tiles costs
  // ceil(T / CUs) tile times -- res5 of 32 images is 1876 tiles = 7.33 rounds.  The rows of the last round go to a SECOND
  // launch of the same kernel whose grid is S copies of those tiles, copy z reducing a slice of the K-steps into an fp32
  // workspace, and a finalize pass adds the slices and applies the epilogue: S is chosen so that the copies fill whole rounds
  // again (res5: 84 tiles x 3 = 252).  Measured (tools/mx_conv_ab.py, 32 images): res5 1.833 -> 1.780 ms per conv; a last
  // round that is more than half full (res4: 170 tiles) is left alone -- its slices' workspace traffic and shorter K loops cost
  // more than the idle CUs, which the chip gives back as clock under its power limit (0.509 -> 0.560 ms with 170 x 3).
  static int n_cu = 0;
  if (!n_cu) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) n_cu = 256;
    else n_cu = std::max(1, prop.multiProcessorCount);
    (void)hipGetLastError();
  }
  const int ntiles = a.tiles_m * a.tiles_n, nk_all = a.K / 64;
  int tail_mt = 0, S = 1;
  {
    const char* ts = getenv("WSOVOD_MX_TAIL");  // (A/B runs: 0 = one launch)
    const bool on = !(ts && ts[0] == '0') && q.nseg_a == 1 && q.nseg_b == 1 && n_cu % a.tiles_n == 0 && nk_all >= 24;
    const int rem = ntiles % n_cu;
    if (on && ntiles > n_cu && rem != 0 && rem % a.tiles_n == 0 && 2 * rem <= n_cu) {
      double best = 1.0 + 0.02;
      for (int c = 2; c <= 8 && nk_all / c >= 12; ++c) {
        const double cost = (double)((rem * c + n_cu - 1) / n_cu) / c + 0.02 * c;
        if (cost < best - 1e-9) best = cost, S = c;
      }
      if (S > 1) tail_mt = rem / a.tiles_n;
    }
  }
  if (tail_mt > 0) {
    // workspace of this process (single-stream use, as the rest of the library): never freed once handed out -- a captured
    // HIP graph keeps the pointer --, growing under stream capture is refused (the policy of gemm8.hip's split-K workspace)
    static float* ws = nullptr;
    static size_t ws_bytes = 0;
    static std::vector<float*> retired;
    const int m_base = (a.tiles_m - tail_mt) * 256;
    const long long ldp = ((long long)d->N + 3) / 4 * 4;
    const size_t need = (size_t)S * (d->M - m_base) * ldp * sizeof(float);
    if (need > ws_bytes) {
      hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
      const bool capturing = s && hipStreamIsCapturing(s, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone;
      float* fresh = nullptr;
      const size_t want = std::max(need, 2 * ws_bytes);
      if (capturing || hipMalloc((void**)&fresh, want) != hipSuccess) {
        (void)hipGetLastError();
        tail_mt = 0;  // (no workspace: the single launch)
      } else {
        if (ws) retired.push_back(ws);
        ws = fresh;
        ws_bytes = want;
      }
    }
    if (tail_mt > 0) {
      a.partial = ws;
      a.partial_ld = ldp;
    }
  }
  wsovod::ProfScope prof(d->conv ? slot_c : slot_g, s, 2.0 * d->M * (double)d->N * d->K, bytes);
  auto launch = [&](const MxArgs& qq, int grid) {
    // the epilogue form (the kernel's EPI): the hot combinations have their own builds
    const bool mxr = d->residual && d->dtype_r == WSOVOD_F16MX;
    const int epi = (d->N % 64 != 0 || qq.g.ksplit > 1 || (d->conv && (d->dropout_p > 0.f || c_bf16))) ? 0
                    : (d->dtype_c == WSOVOD_F16MX && !d->residual) ? 1
                    : (d->dtype_c == WSOVOD_F16MX && mxr && d->conv) ? 2
                    : (d->dtype_c == WSOVOD_F32 && mxr && d->conv) ? 3
                    : (d->dtype_c == WSOVOD_BF16X2 && !d->residual && !d->conv) ? 4 : 0;
#define MX_LAUNCH(C, E) hipLaunchKernelGGL((gemm256_mx_kernel<C, E>), dim3(grid), dim3(512), lds_bytes, s, qq)
    if (d->conv) {
      if (epi == 1) MX_LAUNCH(true, 1);
      else if (epi == 2) MX_LAUNCH(true, 2);
      else if (epi == 3) MX_LAUNCH(true, 3);
      else MX_LAUNCH(true, 0);
    } else {
      if (epi == 1) MX_LAUNCH(false, 1);
      else if (epi == 4) MX_LAUNCH(false, 4);
      else MX_LAUNCH(false, 0);
    }
#undef MX_LAUNCH
  };
  if (tail_mt == 0) {
    launch(q, ntiles);
  } else {
    MxArgs qm = q;  // whole rounds: rows [0, m_base)
    const int m_base = (a.tiles_m - tail_mt) * 256;
    qm.g.M = m_base;
    qm.g.tiles_m = a.tiles_m - tail_mt;
    qm.g.partial = nullptr;
    launch(qm, qm.g.tiles_m * a.tiles_n);
    MxArgs qt = q;  // the last round's rows [m_base, M), S slices of K
    qt.g.m_base = m_base;
    qt.g.tiles_m = tail_mt;
    qt.g.ksplit = S;
    qt.g.slice_steps = (nk_all + S - 1) / S;
    {
      const int run = std::max(1, tail_mt * a.tiles_n / 8);
      int g = 1;
      while ((g + 1) * (g + 1) <= run) ++g;
      qt.g.group_m = std::max(1, std::min(std::min(g, 4), tail_mt));
    }
    launch(qt, tail_mt * a.tiles_n * S);
    const long long quads = (long long)(d->M - m_base) * (d->N / 4);
    hipLaunchKernelGGL(mx_splitk_finalize_kernel, dim3((unsigned)((quads + 255) / 256)), dim3(256), 0, s, qt);
  }
  WS_CHECK_LAUNCH("wsovod_gemm_f16mx");
  return WSOVOD_OK;
}

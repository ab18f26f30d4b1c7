// bf16 256x256x64 GEMM / implicit-GEMM tile in the "8-phase" shape (cdna_hip_programming.md section 5, "The 256^2
// 8-phase template"), written for this library's operand layout and epilogue.
//
//   * 512 threads = 8 wavefronts as 2 (M) x 4 (N); a wavefront owns 128 x 64 of the tile = 8 x 4 MFMA tiles
//     (128 accumulator VGPRs), so a K-step reads (128 + 64) rows x 128 B of fragments per wavefront: 192 KiB per
//     workgroup against 256 KiB for 16 wavefronts of 64 x 64 -- the LDS array is no longer half as busy as the MFMA.
//   * A K-step is four PHASES, one 64 x 32 quadrant of the wavefront's output each (16 MFMAs):
//         phase 1: A rows 0-63, B cols 0-31  (12 ds_read_b128)   quadrant (lo, lo)
//         phase 2:              B cols 32-63 ( 4)                 quadrant (lo, hi)
//         phase 3: A rows 64-127             ( 8, reuse A regs)   quadrant (hi, hi)
//         phase 4: nothing to read                                quadrant (hi, lo)
//     every phase is  [fragment reads, LDS-DMA issue]  s_barrier  [16 MFMA at raised priority]  s_barrier.
//   * The two M-halves of the workgroup (wavefronts 0-3 / 4-7, one of each per SIMD) run STAGGERED by one barrier:
//     while one group is in its MFMA section the other is in its read/stage section, so the matrix pipe and the
//     LDS array are both busy all the time instead of alternating.
//   * Staging is LDS-direct (buffer_load ... lds), two K-step buffers (128 KiB), two DMA instructions per phase,
//     issued as early as the LDS rows they overwrite are free (>= 2 phases after their last read) and retired by
//     COUNTED s_waitcnt vmcnt(N) that leave the younger ones in flight -- the prefetch runs 2-5 phases ahead of its
//     use and crosses the barriers.  Every wait sits before a barrier that each reader passes before it reads
//     those rows, also the staggered group (the "one barrier more" rule of the guide).
//   * Fragment reads are inline asm: hipcc would otherwise put s_waitcnt vmcnt(0) in front of every ds_read it can
//     see while an LDS-DMA is outstanding and drain the prefetch at once.
// Same XOR swizzle (chunk ^ ((row >> 1) & 7), applied to the DMA source address and to the read address), XCD-aware
// grouped tile order and epilogue as gemm.hip.
#include "gemm_common.h"
#include <vector>

namespace wsovod_gemm {

namespace {

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

// X3 (bf16x2 operands, include/wsovod_hip.h): the 128-byte LDS row of a K-step holds the hi halves of 32 values in its
// chunks 0-3 and their lo halves in chunks 4-7, so the SAME two fragment reads per tile (k-half 0 / 1 of the bf16 form)
// deliver (a_hi, a_lo) and (b_hi, b_lo), and a quadrant phase issues 24 MFMAs -- bh*ah, bl*ah, bh*al -- on them instead of
// 16: staging, LDS traffic and barriers of a bf16 GEMM over 2K elements, 1.5x its MFMA work (= 3x the bf16 GEMM over K).
// PH = 2 ("merged" form): a K-step is TWO phases -- (A rows 0-63 x all 64 columns) and (A rows 64-127 x all 64 columns) --
// each [fragment reads, LDS-DMA issue, counted wait] barrier [32 (bf16) / 48 (X3) MFMAs] barrier: half the barriers and
// wait points per MFMA of the four-phase form.  Every group measured ~450 cycles per phase in which its SIMD's matrix
// pipe idles whatever the phase's MFMA count (bf16: 16 MFMAs = 256 cycles at 55 % busy; X3: 24 MFMAs = 384 cycles at
// 61 %), so doubling the MFMAs per phase raises the busy share.  DMA schedule: phase A stages B and A_lo of K-step kt+1
// (their LDS rows were last read two phases earlier) and waits for A_hi(kt); phase B stages A_hi(kt+1) and waits for the
// six instructions of phase A.  Same products in the same order as PH = 4: bit-identical results.
// LEAN (round 5, two-phase form only; K a whole number of K-steps): the half of a phase that the other group's MFMAs have
// to cover -- fragment reads + DMA issue + the counted wait -- without its address arithmetic (s_memtime stamps,
// tools/g8_phases.py: the read section of phase A took 737 ticks on a plain GEMM against 586 (bf16) / 840 (bf16x2) of MFMAs,
// and 1130 on the implicit-GEMM conv, whose tap decode / validity selects sat in front of every DMA instruction):
//  * fragment reads address four per-lane constants; the K-step buffer and the 16-row block are the instruction's
//    immediate offset (K loop unrolled by two);
//  * a DMA instruction = a per-lane CONSTANT offset (rows / columns outside the matrix: 2^31, beyond the resource's range)
//    + a scalar K offset in `soffset` (it takes part in the hardware range check on gfx950: tools/soffset_probe.hip);
//  * conv: the per-lane offsets of the NEXT K-step's tap (the pixel offset where the tap lies inside the image, 2^31
//    where it does not) are computed under the MFMAs of phase B, not in front of the DMA.
template <int OFF>
__device__ __forceinline__ void g8_read_imm(__attribute__((ext_vector_type(4))) unsigned int& dst, unsigned addr) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF));
#endif
}

template <bool CONV, bool X3 = false, int PH = 4, bool LEAN = false>
__global__ __launch_bounds__(512) void gemm256_8ph_kernel(const GemmArgs p) {
  static_assert(!LEAN || PH == 2, "the lean form is a two-phase K-step");
  constexpr int BM = 256, BN = 256, BKE = 64, EPC = 8, esz = 2;
  constexpr int LR = 64;  // rows staged per DMA pass (512 threads x 16 B = 64 rows x 128 B)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  [[maybe_unused]] char* sA = smem;                  // 2 x [256][128 B]
  [[maybe_unused]] char* sB = smem + 2 * BM * 128;   // 2 x [256][128 B]

  const int nwg = p.tiles_m * p.tiles_n;
  int wg;
  {
    const int bid = p.ksplit > 1 ? (int)(blockIdx.x % (unsigned)nwg) : (int)blockIdx.x;  // split-K: slice-major copies of the grid
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int group_size = p.group_m * p.tiles_n;
  const int group_id = wg / group_size;
  const int first_m = group_id * p.group_m;
  const int gm = min(p.tiles_m - first_m, p.group_m);
  const int in_group = wg - group_id * group_size;
  const int tile_m = first_m + in_group % gm;
  const int tile_n = in_group / gm;
  const int m0 = p.m_base + tile_m * BM, n0 = tile_n * BN;  // (m_base: a launch may cover rows [m_base, M) only)

#if defined(G8_STAMPS) && defined(__HIP_DEVICE_COMPILE__)
  const unsigned long long st_begin = __builtin_amdgcn_s_memtime();  // (workgroup start: prologue / loop / rest of the tile)
#endif
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 2, wc = wave & 3;
  const int lrow = tid >> 3;
  const int lchunk = (tid & 7) ^ ((lrow >> 1) & 7);  // swizzle on the DMA source

  [[maybe_unused]] __amdgpu_buffer_rsrc_t rsrcA, rsrcB, rsrcA2;
  int a_off[4], hi0[4], wi0[4], b_off[4];
  [[maybe_unused]] int pix2_off[4];  // conv + fused shortcut: this lane's chunk of its output pixel in A2, <0 = row past M
  if (CONV) {
    rsrcA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, (int)p.a_bytes, 0x00020000);
    rsrcA2 = __builtin_amdgcn_make_buffer_rsrc((void*)(p.A2 ? p.A2 : p.A), 0, (int)(p.A2 ? p.a2_bytes : 0), 0x00020000);
  } else {
    // (planar bf16x2 A: the resource spans the tile's rows in the hi plane up to the same rows in the lo plane)
    const long long rows = min(BM, p.M - m0);
    rsrcA = __builtin_amdgcn_make_buffer_rsrc((void*)(p.A + (long long)m0 * p.lda * esz), 0,
                                              (int)(p.a_plane + rows * p.lda * esz), 0x00020000);
  }
  {
    const long long rows = min(BN, p.N - n0);
    rsrcB = __builtin_amdgcn_make_buffer_rsrc((void*)(p.B + (long long)n0 * p.ldb * esz), 0, (int)(rows * p.ldb * esz),
                                              0x00020000);
  }
  // ---- B rows first: their DMA for K-step 0 is in flight while the A rows' (conv: pixel decode, tap masks) setup runs
  // (round 5: the conv tile's prologue was ~3x the GEMM's; tools/tile_fixed_cost.py)
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    // B pass i = the 64 columns of the wavefronts with wc == i; LDS row (tile j = lrow >> 4, tile row f = lrow & 15) is
    // fed from B row 16*(f>>2) + 4*j + (f&3), so that after the MFMAs a lane owns 16 CONSECUTIVE output columns
    const int src = LR * i + 16 * ((lrow & 15) >> 2) + 4 * (lrow >> 4) + (lrow & 3);
    b_off[i] = n0 + src < p.N ? (int)(((long long)src * p.ldb + lchunk * EPC) * esz) : -1;
  }
  const int kslice = p.ksplit > 1 ? (int)(blockIdx.x / (unsigned)nwg) : 0;
  const int kt_base = kslice * p.slice_steps;  // first K-step of this block (0 unless split-K)
  const int nk = p.ksplit > 1 ? max(0, min((p.K + BKE - 1) / BKE - kt_base, p.slice_steps)) : (p.K + BKE - 1) / BKE;
  typedef __attribute__((address_space(3))) void lds_void [[maybe_unused]];
  [[maybe_unused]] const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  // conv: (filter row, filter column, first channel) of a K-step, advanced incrementally (scalar adds instead of the
  // two integer divisions per staged K-step)
  struct Tap { int r, q, c0; };  // c0 >= Cin: the K-steps of the fused 1x1 shortcut (second input A2 at channel c0 - Cin)
  Tap t0{0, 0, 0};
  if (CONV && kt_base > 0) {  // split-K slice of a conv: the (filter tap, channel chunk) of its first K-step
    const int taps = p.KH * p.KW, nk_main = taps * (p.Cin / BKE);
    if (kt_base >= nk_main) {
      t0.c0 = p.Cin + (kt_base - nk_main) * BKE;
    } else {
      const int chunk = kt_base / taps, tap = kt_base - chunk * taps;
      t0.r = tap / p.KW;
      t0.q = tap - t0.r * p.KW;
      t0.c0 = chunk * BKE;
    }
  }
  auto stage_B = [&](int kt, int buf, int i, const Tap t) {
#if defined(__HIP_DEVICE_COMPILE__)
    // conv: weight rows are [kh][kw][Cin] (+ [Cin2] of the fused shortcut behind them)
    const int kbase = CONV ? (t.c0 >= p.Cin ? p.KH * p.KW * p.Cin + (t.c0 - p.Cin) : (t.r * p.KW + t.q) * p.Cin + t.c0)
                           : (kt + kt_base) * BKE;
    const bool k_ok = kbase + lchunk * EPC < p.K;
    char* dB = sB + buf * BN * 128 + wave_u * 1024 + LR * i * 128;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcB, (lds_void*)dB, 16,
                                             (k_ok && b_off[i] >= 0) ? b_off[i] + kbase * esz : -1, 0, 0, 0);
#endif
  };
  stage_B(0, 0, 0, t0); stage_B(0, 0, 1, t0); stage_B(0, 0, 2, t0); stage_B(0, 0, 3, t0);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + lrow + LR * i;
    const bool ok = m < p.M;
    if (CONV) {
      const int hw = p.Ho * p.Wo;
      const int mm = ok ? m : 0;
      const int img = mm / hw;
      const int rem = mm - img * hw;
      const int ho = rem / p.Wo;
      const int wo = rem - ho * p.Wo;
      hi0[i] = ok ? ho * p.stride - p.pad : -(1 << 28);
      wi0[i] = wo * p.stride - p.pad;
      a_off[i] = (img * p.H * p.W * p.Cin + lchunk * EPC) * esz;
      pix2_off[i] = ok ? (((img * p.Ho + ho) * p.Wo + wo) * p.Cin2 + lchunk * EPC) * esz : -1;
    } else {
      hi0[i] = wi0[i] = 0;
      // planar bf16x2 A (LEAN, X3): chunks 0-3 of a K-step's 128 bytes are 64 bytes of the hi plane's row, chunks 4-7 the
      // same 64 bytes of the lo plane's row
      a_off[i] = !ok ? -1
                 : p.a_plane ? (int)((long long)(lrow + LR * i) * p.lda * esz + (lchunk & 3) * 16 + (lchunk >> 2) * p.a_plane)
                             : (int)(((long long)(lrow + LR * i) * p.lda + lchunk * EPC) * esz);
    }
  }
  // conv: per-lane pixel offset (filter at its top-left tap) and one validity bit per tap, hoisted out of the K loop
  // exactly as in gemm.hip
  [[maybe_unused]] int pix_off[4];
  [[maybe_unused]] unsigned vmask[4];
  if (CONV) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      pix_off[i] = hi0[i] > -(1 << 27) ? a_off[i] + ((hi0[i] * p.W + wi0[i]) * p.Cin) * esz : 0;
  }
  {  // A rows of K-step 0 (passes 0, 2, 1, 3); conv: the tap's validity tested directly -- the masks are built behind the DMA
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
    for (int ii = 0; ii < 4; ++ii) {
      const int i = ((ii & 1) << 1) | (ii >> 1);
      char* dA = sA + wave_u * 1024 + LR * i * 128;
      int off;
      if (CONV) {
        if (t0.c0 >= p.Cin) {
          off = pix2_off[i] >= 0 ? pix2_off[i] + (t0.c0 - p.Cin) * esz : -1;
        } else {
          const bool in = (unsigned)(hi0[i] + t0.r * p.dil) < (unsigned)p.H && (unsigned)(wi0[i] + t0.q * p.dil) < (unsigned)p.W;
          off = in ? pix_off[i] + (((t0.r * p.W + t0.q) * p.dil) * p.Cin + t0.c0) * esz : -1;
        }
        if (t0.c0 >= p.Cin) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcA2, (lds_void*)dA, 16, off, 0, 0, 0);
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcA, (lds_void*)dA, 16, off, 0, 0, 0);
      } else {
        const int kbase = kt_base * BKE;
        const bool k_ok = kbase + lchunk * EPC < p.K;
        off = (k_ok && a_off[i] >= 0) ? a_off[i] + ((kbase * esz) >> (p.a_plane ? 1 : 0)) : -1;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcA, (lds_void*)dA, 16, off, 0, 0, 0);
      }
    }
#endif
  }
  if (CONV) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      // branch-free, KH + KW steps (round 5: the KH x KW double loop with a per-lane `if` was ~1500 instructions of exec
      // masking and scalar branches per tile): valid filter rows x valid filter columns (a row past M: hi0 = -2^28)
      unsigned rowm = 0, colm = 0;
      for (int r = 0; r < p.KH; ++r) rowm |= (unsigned)((unsigned)(hi0[i] + r * p.dil) < (unsigned)p.H) << r;
      for (int q = 0; q < p.KW; ++q) colm |= (unsigned)((unsigned)(wi0[i] + q * p.dil) < (unsigned)p.W) << q;
      unsigned mk = 0;
      for (int r = 0; r < p.KH; ++r) mk |= ((rowm >> r) & 1u) ? (colm << (r * p.KW)) : 0u;
      vmask[i] = mk;
    }
  }

  // one DMA pass = 64 tile rows x 128 B (8 rows per wavefront instruction).  A passes 0 / 2 hold the rows the two
  // wavefront groups read in phase 1 ("A_lo"), passes 1 / 3 the rows they read in phase 3 ("A_hi"); B pass i holds
  // the 64 columns of the wavefronts with wc == i.
  auto tap_next = [&](Tap t) {  // (channel chunk, tap) order with the tap innermost, as gemm.hip: the taps of a chunk
    if (t.c0 >= p.Cin) { t.c0 += BKE; return t; }
    if (++t.q >= p.KW) {        // re-read the same input pixels while they are still in L2
      t.q = 0;
      if (++t.r >= p.KH) { t.r = 0; t.c0 += BKE; }
    }
    return t;
  };
  auto stage_A = [&](int kt, int buf, int i, const Tap t) {
#if defined(__HIP_DEVICE_COMPILE__)
    const int kbase = (kt + kt_base) * BKE;
    char* dA = sA + buf * BM * 128 + wave_u * 1024 + LR * i * 128;
    if (CONV) {
      if (t.c0 >= p.Cin) {  // fused shortcut: the second input at the output pixel (wave-uniform branch)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcA2, (lds_void*)dA, 16,
                                                 pix2_off[i] >= 0 ? pix2_off[i] + (t.c0 - p.Cin) * esz : -1, 0, 0, 0);
      } else {
        const int tap = t.r * p.KW + t.q;
        const int delta = (((t.r * p.W + t.q) * p.dil) * p.Cin + t.c0) * esz;  // wave-uniform
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcA, (lds_void*)dA, 16,
                                                 ((vmask[i] >> tap) & 1u) ? pix_off[i] + delta : -1, 0, 0, 0);
      }
    } else {
      const bool k_ok = kbase + lchunk * EPC < p.K;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcA, (lds_void*)dA, 16,
                                               (k_ok && a_off[i] >= 0) ? a_off[i] + ((kbase * esz) >> (p.a_plane ? 1 : 0)) : -1,
                                               0, 0, 0);
    }
#endif
  };

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int frow = lane & 15, fq = lane >> 4;
  // per-lane LDS byte offsets of its fragment rows (row bases are multiples of 16: the swizzle term is per lane)
  const int sw = (frow >> 1) & 7;
  [[maybe_unused]] const unsigned offA = (unsigned)((wr * 128 + frow) * 128);
  [[maybe_unused]] const unsigned offB = (unsigned)((wc * 64 + frow) * 128);
  [[maybe_unused]] const unsigned c0 = (unsigned)(((fq) ^ sw) << 4), c1 = (unsigned)(((fq + 4) ^ sw) << 4);

  u32x4 af[4][2], bl[2][2], bh[2][2];
#if defined(__HIP_DEVICE_COMPILE__)
  typedef __attribute__((address_space(3))) const char lds_cchar;
  const unsigned ldsA = (unsigned)(size_t)(lds_cchar*)sA, ldsB = (unsigned)(size_t)(lds_cchar*)sB;
#define WS_DS_READ(dst, addr) asm volatile("ds_read_b128 %0, %1" : "=v"(dst) : "v"(addr))
#define WS_LGKM0_12() \
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(af[0][0]), "+v"(af[0][1]), "+v"(af[1][0]), "+v"(af[1][1]), "+v"(af[2][0]), \
               "+v"(af[2][1]), "+v"(af[3][0]), "+v"(af[3][1]), "+v"(bl[0][0]), "+v"(bl[0][1]), "+v"(bl[1][0]), "+v"(bl[1][1]))
#define WS_LGKM0_16() \
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(af[0][0]), "+v"(af[0][1]), "+v"(af[1][0]), "+v"(af[1][1]), "+v"(af[2][0]), \
               "+v"(af[2][1]), "+v"(af[3][0]), "+v"(af[3][1]), "+v"(bl[0][0]), "+v"(bl[0][1]), "+v"(bl[1][0]), "+v"(bl[1][1]), \
               "+v"(bh[0][0]), "+v"(bh[0][1]), "+v"(bh[1][0]), "+v"(bh[1][1]))
#define WS_LGKM0_A() \
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(af[0][0]), "+v"(af[0][1]), "+v"(af[1][0]), "+v"(af[1][1]), "+v"(af[2][0]), \
               "+v"(af[2][1]), "+v"(af[3][0]), "+v"(af[3][1]))
#define WS_LGKM0_BH() \
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bh[0][0]), "+v"(bh[0][1]), "+v"(bh[1][0]), "+v"(bh[1][1]))
#else
#define WS_DS_READ(dst, addr) (void)0
#define WS_LGKM0_16() (void)0
#define WS_LGKM0_12() (void)0
#define WS_LGKM0_A() (void)0
#define WS_LGKM0_BH() (void)0
  const unsigned ldsA = 0, ldsB = 0;
#endif

#if defined(G8_MXPROBE) && defined(__HIP_DEVICE_COMPILE__)
  // TIMING PROBE ONLY (round 6, tools/mx_rate_probe.py; never part of the product library): the MFMA block of a plain-bf16
  // phase replaced by the instruction mix a block-scaled cross-term format would issue on the SAME fragments / LDS image /
  // DMA schedule -- per 32x32 output tile two fp16 32x32x16 products (hi x hi) and ONE v_mfma_scale_f32_32x32x64_f8f6f4
  // (both cross terms as 64 MX-e4m3 values): 12 MFMAs = 512 matrix-pipe cycles per phase instead of 32 x 16.  The bits in
  // the fragments are whatever the bf16 operands hold: RESULTS ARE MEANINGLESS, the time is the point.
  typedef float f32x16 __attribute__((ext_vector_type(16)));
  typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
  typedef int i32x8 __attribute__((ext_vector_type(8)));
  f32x16 pacc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) pacc[i][e] = 0.f;
  const int mx_scale = 0x7f7f7f7f;  // E8M0 2^0 in every byte
  auto mx_cat = [](const u32x4 a, const u32x4 b) {
    return i32x8{(int)a[0], (int)a[1], (int)a[2], (int)a[3], (int)b[0], (int)b[1], (int)b[2], (int)b[3]};
  };
#define WS_MFMA_QUAD(I0, BREG, J0)                                                                                    \
  _Pragma("unroll") for (int t = 0; t < 2; ++t) {                                                                     \
    f32x16& c_ = pacc[(I0) + 2 * t + ((J0) >> 1)];                                                                    \
    c_ = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, BREG[0][0]),                                \
                                                __builtin_bit_cast(f16x8, af[2 * t][0]), c_, 0, 0, 0);                \
    c_ = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, BREG[0][1]),                                \
                                                __builtin_bit_cast(f16x8, af[2 * t][1]), c_, 0, 0, 0);                \
    c_ = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(mx_cat(BREG[1][0], BREG[1][1]),                              \
                                                         mx_cat(af[2 * t + 1][0], af[2 * t + 1][1]), c_, 0, 0, 0,     \
                                                         mx_scale, 0, mx_scale);                                      \
  }
#else
  // bf16: k-halves (0,0), (1,1).  X3: (b_lo, a_hi), (b_hi, a_hi), (b_hi, a_lo) -- the order of gemm.hip's X3 tiles (bit-
  // identical results); the lo*lo term (2^-16 of a product) is dropped
#define WS_MFMA_QUAD(I0, BREG, J0)                                                                                   \
  _Pragma("unroll") for (int ks = 0; ks < (X3 ? 3 : 2); ++ks) _Pragma("unroll") for (int i = 0; i < 4; ++i)           \
      _Pragma("unroll") for (int j = 0; j < 2; ++j) acc[(I0) + i][(J0) + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16( \
          __builtin_bit_cast(bf16x8, BREG[j][X3 ? (ks == 0) : ks]), __builtin_bit_cast(bf16x8, af[i][X3 ? (ks == 2) : ks]), \
          acc[(I0) + i][(J0) + j], 0, 0, 0)
#endif

  // ---- DMA schedule (two instructions per phase; the LDS rows a pass overwrites were last read >= 2 phases ago):
  //   phase 1 (kt): B passes 0,1 of kt+1      phase 2 (kt): B passes 2,3 of kt+1, then vmcnt -> A_hi(kt) landed
  //   phase 3 (kt): A_hi of kt+1              phase 4 (kt): A_lo of kt+2,          then vmcnt -> A_lo, B of kt+1 landed
  // Each wait sits in the read section of its phase, i.e. before a barrier that every reader (also the staggered
  // group) passes before the phase in which it reads those rows.  Counts = DMA instructions issued after the ones
  // waited for; at the tail, where fewer are issued, the waits fall back to vmcnt(0).
#if defined(__HIP_DEVICE_COMPILE__)
#define WS_VMCNT(N) asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory")
#else
#define WS_VMCNT(N) (void)0
#endif
#if defined(G8_STAMPS) && (G8_STAMPS == 1) && defined(__HIP_DEVICE_COMPILE__)
  // instrumented builds only (tools/g8_phases.py): s_memtime ticks per section of the two-phase K-step
  // (-DG8_STAMPS=2: the tile-level stamps only -- setup / first wait / loop / epilogue / store drain -- the loop undisturbed)
  unsigned long long st_t = 0, st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define G8_STAMP0() st_t = __builtin_amdgcn_s_memtime()
#define G8_STAMP(k)                                              \
  {                                                              \
    const unsigned long long now = __builtin_amdgcn_s_memtime(); \
    st_acc[k] += now - st_t;                                     \
    st_t = now;                                                  \
  }
#else
#define G8_STAMP0() (void)0
#define G8_STAMP(k) (void)0
#endif
  Tap t1 = tap_next(t0);   // K-step kt + 1
  Tap t2 = tap_next(t1);   // K-step kt + 2
  // (K-step 0 was requested at the top of the kernel: B rows, then A rows, ahead of the rest of the setup)
  if (PH == 4 && nk > 1) { stage_A(1, 1, 0, t1); stage_A(1, 1, 2, t1); }
#if defined(G8_STAMPS) && defined(__HIP_DEVICE_COMPILE__)
  const unsigned long long st_setup = __builtin_amdgcn_s_memtime();
#endif
  WS_VMCNT(0);
  __builtin_amdgcn_s_barrier();
  if (wr == 1) __builtin_amdgcn_s_barrier();  // stagger: the second M-half runs one barrier behind
#if defined(G8_STAMPS) && defined(__HIP_DEVICE_COMPILE__)
  const unsigned long long st_loop0 = __builtin_amdgcn_s_memtime();
#endif

  if constexpr (LEAN) {
#if defined(__HIP_DEVICE_COMPILE__)
    // ---- per-lane constants of the lean form
    const unsigned rA0 = ldsA + offA + c0, rA1 = ldsA + offA + c1, rB0 = ldsB + offB + c0, rB1 = ldsB + offB + c1;
    constexpr unsigned OOB = 0x80000000u;
    // conv: a pixel offset with the filter at its top-left tap is negative along the image's top / left border, and the
    // range check adds voffset + soffset without wrapping: the resource starts `bias` bytes in front of the map and every
    // per-lane offset carries +bias (the bytes in front are never addressed: their taps are the invalid ones)
    const int bias = CONV ? (p.pad * p.W + p.pad) * p.Cin * esz : 0;
    [[maybe_unused]] const __amdgpu_buffer_rsrc_t rsrcAl = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(p.A - bias), 0, (int)min(p.a_bytes + (long long)bias, (long long)0x7fffffff), 0x00020000);
    unsigned vb[4], va[4];  // DMA source offsets: B rows (loop constants); A rows of the K-step staged next
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      vb[i] = b_off[i] >= 0 ? (unsigned)b_off[i] : OOB;
      va[i] = CONV ? 0u : (a_off[i] >= 0 ? (unsigned)a_off[i] : OOB);
    }
    // conv: offsets of tap t for this lane's four rows -- the pixel offset (filter at its top-left tap) where the tap
    // lies inside the image, out of range where it does not; the tap's own displacement is the scalar `soffset`.
    // Branch-free (16 VALU instructions on loop constants): it is issued BETWEEN the products of phase B
    unsigned pixb[4], pix2v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      pixb[i] = CONV ? (unsigned)(pix_off[i] + bias) : 0u;
      pix2v[i] = (CONV && pix2_off[i] >= 0) ? (unsigned)pix2_off[i] : OOB;
    }
    auto conv_va = [&](const Tap t) {
      const bool sec = t.c0 >= p.Cin;                                    // the fused 1x1 shortcut's K-steps (scalar)
      const unsigned tapbit = sec ? 0u : (1u << (t.r * p.KW + t.q));     // scalar
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const unsigned alt = sec ? pix2v[i] : OOB;
        va[i] = (vmask[i] & tapbit) ? pixb[i] : alt;
      }
    };
    // scalar byte offsets of a K-step: into the A operand (conv: the tap's displacement + channel chunk) and the B rows
    auto soff_a = [&](int kt, const Tap t) -> int {
      if (!CONV) return ((kt + kt_base) * (BKE * esz)) >> (p.a_plane ? 1 : 0);  // planar A: 64 bytes of each plane per K-step
      if (t.c0 >= p.Cin) return (t.c0 - p.Cin) * esz;
      return (((t.r * p.W + t.q) * p.dil) * p.Cin + t.c0) * esz;
    };
    auto soff_b = [&](int kt, const Tap t) -> int {
      if (!CONV) return (kt + kt_base) * (BKE * esz);
      return (t.c0 >= p.Cin ? p.KH * p.KW * p.Cin + (t.c0 - p.Cin) : (t.r * p.KW + t.q) * p.Cin + t.c0) * esz;
    };
    auto dma_a = [&](int buf, int i, int so, bool second) {
      char* dA = sA + buf * BM * 128 + wave_u * 1024 + LR * i * 128;
      if (CONV && second)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcA2, (lds_void*)dA, 16, (int)va[i], so, 0, 0);
      else if (CONV)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcAl, (lds_void*)dA, 16, (int)va[i], so, 0, 0);
      else
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcA, (lds_void*)dA, 16, (int)va[i], so, 0, 0);
    };
    auto dma_b = [&](int buf, int i, int so) {
      char* dB = sB + buf * BN * 128 + wave_u * 1024 + LR * i * 128;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcB, (lds_void*)dB, 16, (int)vb[i], so, 0, 0);
    };
    // one K-step on buffer CUR; `more`: K-step kt + 1 exists (its tap is t1, its A offsets are in va)
    auto kstep = [&](auto cur_c, int kt) {
      constexpr int CUR = decltype(cur_c)::value;
      constexpr int IA = CUR * (BM * 128), IB = CUR * (BN * 128);
      const bool more = kt + 1 < nk;
      const bool second = CONV && t1.c0 >= p.Cin;
      const int sa = soff_a(kt + 1, t1), sb = soff_b(kt + 1, t1);
      // ---- phase A: A rows 0-63 x all 64 columns of this wavefront (16 fragment reads)
      G8_STAMP0();
      g8_read_imm<IB + 0 * 2048>(bl[0][0], rB0); g8_read_imm<IB + 0 * 2048>(bl[0][1], rB1);
      g8_read_imm<IB + 1 * 2048>(bl[1][0], rB0); g8_read_imm<IB + 1 * 2048>(bl[1][1], rB1);
      g8_read_imm<IA + 0 * 2048>(af[0][0], rA0); g8_read_imm<IA + 0 * 2048>(af[0][1], rA1);
      g8_read_imm<IA + 1 * 2048>(af[1][0], rA0); g8_read_imm<IA + 1 * 2048>(af[1][1], rA1);
      g8_read_imm<IA + 2 * 2048>(af[2][0], rA0); g8_read_imm<IA + 2 * 2048>(af[2][1], rA1);
      g8_read_imm<IA + 3 * 2048>(af[3][0], rA0); g8_read_imm<IA + 3 * 2048>(af[3][1], rA1);
      g8_read_imm<IB + 2 * 2048>(bh[0][0], rB0); g8_read_imm<IB + 2 * 2048>(bh[0][1], rB1);
      g8_read_imm<IB + 3 * 2048>(bh[1][0], rB0); g8_read_imm<IB + 3 * 2048>(bh[1][1], rB1);
      G8_STAMP(0);
      if (more) {  // rows last read two phases ago (phase A of kt-1)
        dma_b(CUR ^ 1, 0, sb); dma_b(CUR ^ 1, 1, sb); dma_b(CUR ^ 1, 2, sb); dma_b(CUR ^ 1, 3, sb);
        dma_a(CUR ^ 1, 0, sa, second); dma_a(CUR ^ 1, 2, sa, second);
        G8_STAMP(1);
        WS_VMCNT(6);  // younger: these six -> A_hi(kt) has landed (read in phase B)
      } else {
        WS_VMCNT(0);
      }
      G8_STAMP(2);
      __builtin_amdgcn_s_barrier();
      G8_STAMP(3);
      WS_LGKM0_16();
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_setprio(1);
      WS_MFMA_QUAD(0, bl, 0);
      WS_MFMA_QUAD(0, bh, 2);
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
      G8_STAMP(4);
      __builtin_amdgcn_s_barrier();
      G8_STAMP(5);
      // ---- phase B: A rows 64-127 (into the same registers) x all 64 columns
      g8_read_imm<IA + 4 * 2048>(af[0][0], rA0); g8_read_imm<IA + 4 * 2048>(af[0][1], rA1);
      g8_read_imm<IA + 5 * 2048>(af[1][0], rA0); g8_read_imm<IA + 5 * 2048>(af[1][1], rA1);
      g8_read_imm<IA + 6 * 2048>(af[2][0], rA0); g8_read_imm<IA + 6 * 2048>(af[2][1], rA1);
      g8_read_imm<IA + 7 * 2048>(af[3][0], rA0); g8_read_imm<IA + 7 * 2048>(af[3][1], rA1);
      if (more) {  // rows last read two phases ago (phase B of kt-1)
        dma_a(CUR ^ 1, 1, sa, second); dma_a(CUR ^ 1, 3, sa, second);
        WS_VMCNT(2);  // younger: these two -> B(kt+1) and A_lo(kt+1) have landed (read in the next phase A)
      } else {
        WS_VMCNT(0);
      }
      if (CONV) t1 = tap_next(t1);  // the tap after next (scalar state; its per-lane offsets follow under the products)
      G8_STAMP(6);
      __builtin_amdgcn_s_barrier();
      WS_LGKM0_A();
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_setprio(1);
      WS_MFMA_QUAD(4, bh, 2);
      if (CONV) {
        // (the empty asm pins the results HERE: hipcc otherwise sinks the selects to their use, in front of the next
        // K-step's DMA instructions -- the section the other group's products have to cover)
        conv_va(t1);
        asm volatile("" : "+v"(va[0]), "+v"(va[1]), "+v"(va[2]), "+v"(va[3]));
      }
      WS_MFMA_QUAD(4, bl, 0);
      if (CONV) {  // conv_va's instructions one at a time behind the products: they issue in the matrix pipe's shadow
#pragma unroll
        for (int g_ = 0; g_ < 20; ++g_) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);
        }
      }
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      G8_STAMP(7);
    };
    if (CONV) conv_va(t1);  // (the prologue above staged K-step 0 through the generic path; t1 = K-step 1)
    int kt = 0;
    for (; kt + 1 < nk; kt += 2) {
      kstep(std::integral_constant<int, 0>{}, kt);
      kstep(std::integral_constant<int, 1>{}, kt + 1);
    }
    if (kt < nk) kstep(std::integral_constant<int, 0>{}, kt);
#endif
  } else
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    const bool more = kt + 1 < nk, more2 = kt + 2 < nk;
    [[maybe_unused]] const unsigned bA = ldsA + cur * (BM * 128) + offA, bB = ldsB + cur * (BN * 128) + offB;
    if constexpr (PH == 2) {
      // ---- phase A: A rows 0-63 x all 64 columns of this wavefront (16 fragment reads)
      G8_STAMP0();
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        WS_DS_READ(bl[j][0], bB + j * 2048 + c0);
        WS_DS_READ(bl[j][1], bB + j * 2048 + c1);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        WS_DS_READ(af[i][0], bA + i * 2048 + c0);
        WS_DS_READ(af[i][1], bA + i * 2048 + c1);
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        WS_DS_READ(bh[j][0], bB + (2 + j) * 2048 + c0);
        WS_DS_READ(bh[j][1], bB + (2 + j) * 2048 + c1);
      }
      G8_STAMP(0);
      if (more) {  // rows last read two phases ago (phase A of kt-1)
        stage_B(kt + 1, cur ^ 1, 0, t1); stage_B(kt + 1, cur ^ 1, 1, t1);
        stage_B(kt + 1, cur ^ 1, 2, t1); stage_B(kt + 1, cur ^ 1, 3, t1);
        stage_A(kt + 1, cur ^ 1, 0, t1); stage_A(kt + 1, cur ^ 1, 2, t1);
        G8_STAMP(1);
        WS_VMCNT(6);  // younger: these six -> A_hi(kt) has landed (read in phase B)
      } else {
        WS_VMCNT(0);
      }
      G8_STAMP(2);
      __builtin_amdgcn_s_barrier();
      G8_STAMP(3);
      WS_LGKM0_16();
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_setprio(1);
      WS_MFMA_QUAD(0, bl, 0);
      WS_MFMA_QUAD(0, bh, 2);
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
      G8_STAMP(4);
      __builtin_amdgcn_s_barrier();
      G8_STAMP(5);
      // ---- phase B: A rows 64-127 (into the same registers) x all 64 columns
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        WS_DS_READ(af[i][0], bA + (4 + i) * 2048 + c0);
        WS_DS_READ(af[i][1], bA + (4 + i) * 2048 + c1);
      }
      if (more) {  // rows last read two phases ago (phase B of kt-1)
        stage_A(kt + 1, cur ^ 1, 1, t1); stage_A(kt + 1, cur ^ 1, 3, t1);
        WS_VMCNT(2);  // younger: these two -> B(kt+1) and A_lo(kt+1) have landed (read in the next phase A)
      } else {
        WS_VMCNT(0);
      }
      G8_STAMP(6);
      __builtin_amdgcn_s_barrier();
      WS_LGKM0_A();
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_setprio(1);
      WS_MFMA_QUAD(4, bh, 2);
      WS_MFMA_QUAD(4, bl, 0);
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      G8_STAMP(7);
      if (CONV) { t1 = tap_next(t1); }
      continue;
    }
    // ---- phase 1: A rows 0-63 + B cols 0-31 of this wavefront
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      WS_DS_READ(bl[j][0], bB + j * 2048 + c0);
      WS_DS_READ(bl[j][1], bB + j * 2048 + c1);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      WS_DS_READ(af[i][0], bA + i * 2048 + c0);
      WS_DS_READ(af[i][1], bA + i * 2048 + c1);
    }
    if (more) { stage_B(kt + 1, cur ^ 1, 0, t1); stage_B(kt + 1, cur ^ 1, 1, t1); }
    __builtin_amdgcn_s_barrier();
    WS_LGKM0_12();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
    WS_MFMA_QUAD(0, bl, 0);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    // ---- phase 2: B cols 32-63
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      WS_DS_READ(bh[j][0], bB + (2 + j) * 2048 + c0);
      WS_DS_READ(bh[j][1], bB + (2 + j) * 2048 + c1);
    }
    if (more) {
      stage_B(kt + 1, cur ^ 1, 2, t1); stage_B(kt + 1, cur ^ 1, 3, t1);
      WS_VMCNT(6);  // younger: A_lo(kt+1), B(kt+1) -> A_hi(kt) has landed
    } else {
      WS_VMCNT(0);
    }
    __builtin_amdgcn_s_barrier();
    WS_LGKM0_BH();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
    WS_MFMA_QUAD(0, bh, 2);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    // ---- phase 3: A rows 64-127 (into the same registers)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      WS_DS_READ(af[i][0], bA + (4 + i) * 2048 + c0);
      WS_DS_READ(af[i][1], bA + (4 + i) * 2048 + c1);
    }
    if (more) { stage_A(kt + 1, cur ^ 1, 1, t1); stage_A(kt + 1, cur ^ 1, 3, t1); }
    __builtin_amdgcn_s_barrier();
    WS_LGKM0_A();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
    WS_MFMA_QUAD(4, bh, 2);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    // ---- phase 4: no fragment reads
    if (more2) {
      stage_A(kt + 2, cur, 0, t2); stage_A(kt + 2, cur, 2, t2);
      WS_VMCNT(4);  // younger: A_hi(kt+1), A_lo(kt+2) -> A_lo(kt+1) and B(kt+1) have landed
    } else {
      WS_VMCNT(0);
    }
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_s_setprio(1);
    WS_MFMA_QUAD(4, bl, 0);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    if (CONV) { t1 = t2; t2 = tap_next(t2); }
  }
  if (wr == 0) __builtin_amdgcn_s_barrier();  // balance the stagger barrier
#if defined(G8_MXPROBE) && defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] += f32x4{pacc[i][4 * j], pacc[i][4 * j + 1], pacc[i][4 * j + 2], pacc[i][4 * j + 3]};
#endif
#if defined(G8_STAMPS) && defined(__HIP_DEVICE_COMPILE__)
  const unsigned long long st_loop1 = __builtin_amdgcn_s_memtime();
  if (p.partial && lane == 0 && p.ksplit <= 1) {
#if G8_STAMPS == 1
#pragma unroll
    for (int k = 0; k < 8; ++k) atomicAdd(p.partial + wr * 16 + k, (float)st_acc[k]);
#endif
    atomicAdd(p.partial + wr * 16 + 8, (float)nk);
  }
  // tile level, one wavefront per group: [9] entry -> K-step 0 requested and the setup done, [12] -> its data landed +
  // barriers, [10] the K loop, [13] epilogue until the last store is issued, [14] until the stores are acknowledged
  auto g8_tile_end = [&]() {
    const unsigned long long st_e0 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long st_e1 = __builtin_amdgcn_s_memtime();
    if (p.partial && lane == 0 && p.ksplit <= 1 && (wave == 0 || wave == 4)) {
      atomicAdd(p.partial + wr * 16 + 9, (float)(st_setup - st_begin));
      atomicAdd(p.partial + wr * 16 + 12, (float)(st_loop0 - st_setup));
      atomicAdd(p.partial + wr * 16 + 10, (float)(st_loop1 - st_loop0));
      atomicAdd(p.partial + wr * 16 + 13, (float)(st_e0 - st_loop1));
      atomicAdd(p.partial + wr * 16 + 14, (float)(st_e1 - st_e0));
      atomicAdd(p.partial + wr * 16 + 11, 1.0f);
    }
  };
#define G8_TILE_END() g8_tile_end()
#else
#define G8_TILE_END() (void)0
#endif
#undef WS_VMCNT

  // ---- epilogue (fp32).  The MFMAs were issued with the operands swapped (B fragment first) and the B rows permuted
  // at staging, so lane (frow, fq) holds output row m = .. + frow and the 16 consecutive columns ncol + 4*j + r:
  // per output row a wavefront writes 64 contiguous elements as 16-byte stores.
  // Tile indices are compile-time constants (a runtime index into acc would put the accumulators in scratch).
  if (p.ksplit > 1) {  // split-K: raw partial sums of this K slice; the epilogue runs in splitk_finalize_kernel
    float* part = p.partial + ((long long)kslice * (p.M - p.m_base) - p.m_base) * p.partial_ld;  // rows from m_base
    const int ncol0 = n0 + wc * 64 + 16 * fq;
#define WS_PART_ROW(I)                                                                                     \
    {                                                                                                        \
      const int m = m0 + wr * 128 + (I) * 16 + frow;                                                         \
      if (m < p.M) {                                                                                         \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                      \
          const int nb = ncol0 + 4 * j;                                                                      \
          if (nb + 3 < p.N) *(f32x4*)(part + (long long)m * p.partial_ld + nb) = acc[I][j];                  \
          else _Pragma("unroll") for (int r = 0; r < 4; ++r) if (nb + r < p.N)                               \
              part[(long long)m * p.partial_ld + nb + r] = acc[I][j][r];                                     \
        }                                                                                                    \
      }                                                                                                      \
    }
    WS_PART_ROW(0) WS_PART_ROW(1) WS_PART_ROW(2) WS_PART_ROW(3) WS_PART_ROW(4) WS_PART_ROW(5) WS_PART_ROW(6) WS_PART_ROW(7)
#undef WS_PART_ROW
    return;
  }
  const float keep_scale = p.dropout_p > 0.f ? 1.0f / (1.0f - p.dropout_p) : 1.0f;
  [[maybe_unused]] const unsigned long long dseed = p.dropout_p > 0.f ? WS_DROPOUT_SEED(p) : 0ull;
  [[maybe_unused]] const unsigned dthr = dropout_threshold(p.dropout_p);
  const bool vec_c = p.C && (p.dtype_c == WSOVOD_BF16X2 ? vec4_ok(p.C, p.ldc, p.dtype_c)
                                                        : (p.ldc & 7) == 0 && ((uintptr_t)p.C & 15) == 0);
  const int ncol = n0 + wc * 64 + 16 * fq;
  auto emit = [&](const f32x4 a4, const int i, const int j) {
    const int m = m0 + wr * 128 + i * 16 + frow;
    const int nb = ncol + 4 * j;
    if (m >= p.M || nb >= p.N) return;
    const float rs = p.row_scale ? p.row_scale[m] : 1.f;
    const bool full = nb + 3 < p.N;
    const unsigned long long dz = p.dropout_p > 0.f ? dropout_quad(dseed, m, p.N, nb) : 0ull;
    float v[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = nb + r;
      float x = a4[r] * p.alpha;
      if (n < p.N) {
        if (p.row_scale) x *= rs;
        if (p.bias) x += p.bias[n];
        if (p.residual) x += load_as_f32(p.residual, m, p.ldr, n, p.dtype_r);
        if (p.relu) x = fmaxf(x, 0.f);
        if (p.dropout_p > 0.f) x = dropout_keep(dz, r, dthr) ? x * keep_scale : 0.f;
        if (p.group_add) x += p.group_add[(long long)p.row_group[m] * p.ld_ga + n];
        if (p.mask_src)
          x = load_as_f32(p.mask_src, m, p.ldm, n, p.dtype_m) > 0.f ? x * p.mask_scale : 0.f;
        if (p.C && p.accumulate) x += ((float*)p.C)[(long long)m * p.ldc + n];
      }
      v[r] = x;
    }
    if (p.C) {
      if (vec_c && full) {
        store4_from_f32(p.C, m, p.ldc, nb, p.dtype_c, f32x4{v[0], v[1], v[2], v[3]});
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (nb + r < p.N) store_from_f32(p.C, m, p.ldc, nb + r, p.dtype_c, v[r]);
      }
    }
    if (p.Ct) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (nb + r < p.N) store_from_f32(p.Ct, nb + r, p.ldct, m, p.dtype_ct, v[r]);
    }
  };
  // Fast path (bias / residual / ReLU / dropout, row-major output with 16-byte aligned rows, whole tile columns in
  // range): the feature tests are hoisted out of the element loops, bias is fetched once per column tile, residual
  // and output move as 16-byte accesses.
  const bool vec_r = !p.residual || (p.dtype_r == WSOVOD_BF16X2 ? vec4_ok(p.residual, p.ldr, p.dtype_r)
                                                                 : (p.ldr & 7) == 0 && ((uintptr_t)p.residual & 15) == 0);
  const bool plain = vec_c && vec_r && ((uintptr_t)p.bias & 15) == 0 && !p.Ct && !p.row_scale && !p.group_add &&
                     !p.mask_src && !p.accumulate && n0 + BN <= p.N;
  if (plain) {
    f32x4 b4[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) b4[j] = p.bias ? *(const f32x4*)(p.bias + ncol + 4 * j) : f32x4{0.f, 0.f, 0.f, 0.f};
    const float lo = p.relu ? 0.f : -__builtin_inff();
    const int mrow = m0 + wr * 128 + frow;
    const bool drop = p.dropout_p > 0.f;
    const bool has_res = p.residual != nullptr;
#define WS_FAST_ROW(I)                                                                                        \
  if (mrow + (I) * 16 < p.M) {                                                                                \
    const long long mm = mrow + (I) * 16;                                                                     \
    const long long base = mm * p.ldc + ncol;                                                                 \
    f32x4 x[4];                                                                                               \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                           \
      x[j] = acc[I][j] * p.alpha + b4[j];                                                                     \
      if (has_res) x[j] += load4_as_f32(p.residual, mm, p.ldr, ncol + 4 * j, p.dtype_r);                      \
      x[j] = f32x4{fmaxf(x[j][0], lo), fmaxf(x[j][1], lo), fmaxf(x[j][2], lo), fmaxf(x[j][3], lo)};           \
      if (drop) {                                                                                             \
        const unsigned long long dz = dropout_quad(dseed, mm, p.N, ncol + 4 * j);                             \
        _Pragma("unroll") for (int r = 0; r < 4; ++r)                                                         \
            x[j][r] = dropout_keep(dz, r, dthr) ? x[j][r] * keep_scale : 0.f;                                 \
      }                                                                                                       \
    }                                                                                                         \
    if (p.dtype_c == WSOVOD_BF16X2) { /* 16 consecutive values: 32 B of hi, 32 B of lo one half-line further */  \
      bf16_t* q = (bf16_t*)p.C + 2 * mm * p.ldc + x2_pos(ncol);                                               \
      _Pragma("unroll") for (int j = 0; j < 4; j += 2) {                                                      \
        bf16x8 h, l;                                                                                          \
        _Pragma("unroll") for (int r = 0; r < 4; ++r) {                                                       \
          h[r] = (bf16_t)x[j][r];         l[r] = x2_lo(x[j][r], h[r]);                                        \
          h[4 + r] = (bf16_t)x[j + 1][r]; l[4 + r] = x2_lo(x[j + 1][r], h[4 + r]);                            \
        }                                                                                                     \
        *(bf16x8*)(q + 4 * j) = h;                                                                            \
        *(bf16x8*)(q + 32 + 4 * j) = l;                                                                       \
      }                                                                                                       \
    } else if (p.dtype_c == WSOVOD_BF16) {                                                                    \
      _Pragma("unroll") for (int j = 0; j < 4; j += 2)                                                        \
          *(bf16x8*)((bf16_t*)p.C + base + 4 * j) =                                                           \
              bf16x8{(bf16_t)x[j][0],     (bf16_t)x[j][1],     (bf16_t)x[j][2],     (bf16_t)x[j][3],          \
                     (bf16_t)x[j + 1][0], (bf16_t)x[j + 1][1], (bf16_t)x[j + 1][2], (bf16_t)x[j + 1][3]};     \
    } else {                                                                                                  \
      _Pragma("unroll") for (int j = 0; j < 4; ++j) *(f32x4*)((float*)p.C + base + 4 * j) = x[j];             \
    }                                                                                                         \
  }
    WS_FAST_ROW(0) WS_FAST_ROW(1) WS_FAST_ROW(2) WS_FAST_ROW(3) WS_FAST_ROW(4) WS_FAST_ROW(5) WS_FAST_ROW(6) WS_FAST_ROW(7)
#undef WS_FAST_ROW
    G8_TILE_END();
    return;
  }
#define WS_EMIT_ROW(I) emit(acc[I][0], I, 0); emit(acc[I][1], I, 1); emit(acc[I][2], I, 2); emit(acc[I][3], I, 3)
  WS_EMIT_ROW(0); WS_EMIT_ROW(1); WS_EMIT_ROW(2); WS_EMIT_ROW(3);
  WS_EMIT_ROW(4); WS_EMIT_ROW(5); WS_EMIT_ROW(6); WS_EMIT_ROW(7);
  G8_TILE_END();
#undef WS_EMIT_ROW
#undef WS_DS_READ
#undef WS_LGKM0_12
#undef WS_LGKM0_16
#undef WS_LGKM0_A
#undef WS_LGKM0_BH
#undef WS_MFMA_QUAD
}

// split-K finalize: C[m][n] = epilogue(sum over the K slices) -- the epilogue chain of the GEMM kernels, element-wise
__global__ __launch_bounds__(256) void splitk_finalize_kernel(const GemmArgs p) {
  const int n4 = (p.N + 3) >> 2;
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  const int rows = p.M - p.m_base;
  if (idx >= (long long)rows * n4) return;
  const int mr = (int)(idx / n4), nb = (int)(idx - (long long)mr * n4) * 4, m = p.m_base + mr;
  float v[4] = {0.f, 0.f, 0.f, 0.f};
  for (int z = 0; z < p.ksplit; ++z) {
    const float* src = p.partial + ((long long)z * rows + mr) * p.partial_ld + nb;
    if (nb + 3 < p.N) {
      const f32x4 t = *(const f32x4*)src;
      v[0] += t[0]; v[1] += t[1]; v[2] += t[2]; v[3] += t[3];
    } else {
      for (int r = 0; r < 4; ++r)
        if (nb + r < p.N) v[r] += src[r];
    }
  }
  epilogue_store4(p, m, nb, v);
}

}  // namespace

int launch_gemm256_8ph(const GemmArgs& a, bool conv, hipStream_t s, double flops, double bytes, bool allow_split, bool x3,
                       bool merged) {
  allow_split = allow_split || a.ksplit == -1;  // -1: the dispatcher chose this tile itself (no tile_hint)
  static int slot_g = wsovod::prof_slot("gemm_nt_bf16_256x256_8ph");
  static int slot_c = wsovod::prof_slot("conv_igemm_bf16_256x256_8ph");
  static int slot_g3 = wsovod::prof_slot("gemm_nt_bf16x2_256x256_8ph");
  static int slot_c3 = wsovod::prof_slot("conv_igemm_bf16x2_256x256_8ph");
  static bool attr_set = false;
  constexpr int lds_bytes = 2 * (256 + 256) * 128;
  if (!attr_set) {
    WS_CHECK_HIP(hipFuncSetAttribute((const void*)gemm256_8ph_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes), "wsovod_gemm_nt (8-phase tile): LDS opt-in");
    WS_CHECK_HIP(hipFuncSetAttribute((const void*)gemm256_8ph_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes), "wsovod_gemm_nt (8-phase tile): LDS opt-in");
    WS_CHECK_HIP(hipFuncSetAttribute((const void*)gemm256_8ph_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes), "wsovod_gemm_nt (8-phase tile): LDS opt-in");
    WS_CHECK_HIP(hipFuncSetAttribute((const void*)gemm256_8ph_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes), "wsovod_gemm_nt (8-phase tile): LDS opt-in");
    WS_CHECK_HIP(hipFuncSetAttribute((const void*)gemm256_8ph_kernel<false, false, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes), "wsovod_gemm_nt (8-phase tile): LDS opt-in");
    WS_CHECK_HIP(hipFuncSetAttribute((const void*)gemm256_8ph_kernel<true, false, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes), "wsovod_gemm_nt (8-phase tile): LDS opt-in");
    WS_CHECK_HIP(hipFuncSetAttribute((const void*)gemm256_8ph_kernel<false, true, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes), "wsovod_gemm_nt (8-phase tile): LDS opt-in");
    WS_CHECK_HIP(hipFuncSetAttribute((const void*)gemm256_8ph_kernel<true, true, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes), "wsovod_gemm_nt (8-phase tile): LDS opt-in");
    WS_CHECK_HIP(hipFuncSetAttribute((const void*)gemm256_8ph_kernel<false, false, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes), "wsovod_gemm_nt (8-phase tile): LDS opt-in");
    WS_CHECK_HIP(hipFuncSetAttribute((const void*)gemm256_8ph_kernel<true, false, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes), "wsovod_gemm_nt (8-phase tile): LDS opt-in");
    WS_CHECK_HIP(hipFuncSetAttribute((const void*)gemm256_8ph_kernel<false, true, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes), "wsovod_gemm_nt (8-phase tile): LDS opt-in");
    WS_CHECK_HIP(hipFuncSetAttribute((const void*)gemm256_8ph_kernel<true, true, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes), "wsovod_gemm_nt (8-phase tile): LDS opt-in");
    attr_set = true;
  }
  GemmArgs args = a;
#if defined(G8_STAMPS)
  if (const char* dp = getenv("WSOVOD_G8_DEBUG_PTR")) args.partial = (float*)strtoull(dp, nullptr, 16);
#endif
  args.tiles_m = ceil_div(a.M - a.m_base, 256);
  args.tiles_n = ceil_div(a.N, 256);
  {
    // Tile-order group height.  What has to share L2 is the set of workgroups an XCD runs AT THE SAME TIME (32 CUs x 1
    // workgroup), not its whole run of tiles: group_m x (32 / group_m) of them should be near-square.  Measured on the fc1
    // forward shape (16384 x 4096 x 25088): group 2/4/8/11/16/32 -> 1345/1350/1322/1298/1292/1260 TFLOP/s.
    const int run = std::max(1, args.tiles_m * args.tiles_n / 8);
    int g = 1;
    while ((g + 1) * (g + 1) <= run) ++g;
    args.group_m = std::max(1, std::min(std::min(g, 4), args.tiles_m));
  }
  // Split-K for contractions with few output tiles and a long K (the FC layers at 1-4 images per step: fc1 forward is
  // 2 x 16 tiles of K = 25088): `ksplit` copies of the grid fill the chip, each reducing a slice of K into a workspace,
  // and a finalize kernel adds the slices and applies the epilogue.  The 256x256 tile keeps the operand traffic through
  // L2 low (a 64x64 tiling of the same GEMM re-reads the operands 8x / 64x and is L2-bound at ~1 TB/s of HBM).
  args.ksplit = args.slice_steps = 0;
  const int ntiles = args.tiles_m * args.tiles_n, nk = ceil_div(a.K, 64);
  int grid = ntiles;
  // (round 5: also the implicit-GEMM convs at 1 - 4 images per step -- res5 of ONE 800x600 image is 30 x 2 tiles --
  // where the alternative was a 128x128 / 64x64 grid at 0.12 - 0.25 of peak)
  if (allow_split && ntiles <= 128 && nk >= 32) {
    int S = std::min(std::min(8, 256 / ntiles), nk / 16);
    if (const char* fs = getenv("WSOVOD_SPLITK_S")) S = std::max(1, std::min(atoi(fs), nk / 4));  // (experiments)
    if (S >= 2) {
      // Workspace of this process (single-stream use, as the rest of the library).  A captured HIP graph keeps the
      // pointer it was captured with, so a block is NEVER freed once handed out: when a larger one is needed the old
      // block is retired (kept allocated; sizes double, so all retired blocks together are smaller than the live one,
      // and the largest possible request is 8 slices x 128 tiles = 268 MB), and growing under stream capture is refused
      // instead of calling hipMalloc inside the capture (the conv forms make `need` shape-dependent).
      static float* ws = nullptr;
      static size_t ws_bytes = 0;
      static std::vector<float*> retired;
      const long long ldp = ((long long)a.N + 3) / 4 * 4;
      const size_t need = (size_t)S * (a.M - a.m_base) * ldp * sizeof(float);
      if (need > ws_bytes) {
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        if (s && hipStreamIsCapturing(s, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone) {
          wsovod::set_error("wsovod_gemm_nt: the split-K workspace would have to grow under stream capture; run the "
                            "shape once outside the capture first");
          return WSOVOD_ERR_UNSUPPORTED;
        }
        const size_t want = std::max(need, 2 * ws_bytes);
        float* fresh = nullptr;
        if (hipMalloc((void**)&fresh, want) != hipSuccess) {
          wsovod::set_error("wsovod_gemm_nt: cannot allocate the split-K workspace");
          return WSOVOD_ERR_HIP;
        }
        if (ws) retired.push_back(ws);
        ws = fresh;
        ws_bytes = want;
      }
      args.ksplit = S;
      args.slice_steps = ceil_div(nk, S);
      args.partial = ws;
      args.partial_ld = ldp;
      grid = ntiles * S;
    }
  }
  static int slot_fin = wsovod::prof_slot("splitk_finalize");
  {
  wsovod::ProfScope prof(x3 ? (conv ? slot_c3 : slot_g3) : (conv ? slot_c : slot_g), s, flops, bytes);
#define WS_L8(C, X, P) hipLaunchKernelGGL((gemm256_8ph_kernel<C, X, P>), dim3(grid), dim3(512), lds_bytes, s, args)
#define WS_L8L(C, X) hipLaunchKernelGGL((gemm256_8ph_kernel<C, X, 2, true>), dim3(grid), dim3(512), lds_bytes, s, args)
  // the lean two-phase form: whole K-steps only (no K tail select in front of its DMA instructions)
  const char* le = getenv("WSOVOD_G8_LEAN");
  const bool lean = merged && a.K % 64 == 0 && !(le && le[0] == '0') &&
                    (!conv || (a.Cin % 64 == 0 && (!a.A2 || a.Cin2 % 64 == 0) &&
                               a.a_bytes + (long long)(a.pad * a.W + a.pad) * a.Cin * 2 < (1ll << 31)));
  if (a.a_plane && !(lean && x3 && !conv)) {
    wsovod::set_error("wsovod_gemm_nt: a planar bf16x2 A operand is served by the lean two-phase tile only (plain GEMM, "
                      "K a multiple of 32 values)");
    return WSOVOD_ERR_UNSUPPORTED;
  }
  if (lean) {
    if (x3 && conv) WS_L8L(true, true);
    else if (x3) WS_L8L(false, true);
    else if (conv) WS_L8L(true, false);
    else WS_L8L(false, false);
  } else if (merged) {
    if (x3 && conv) WS_L8(true, true, 2);
    else if (x3) WS_L8(false, true, 2);
    else if (conv) WS_L8(true, false, 2);
    else WS_L8(false, false, 2);
  } else {
    if (x3 && conv) WS_L8(true, true, 4);
    else if (x3) WS_L8(false, true, 4);
    else if (conv) WS_L8(true, false, 4);
    else WS_L8(false, false, 4);
  }
#undef WS_L8
#undef WS_L8L
  }
  if (args.ksplit > 1) {
    const long long quads = (long long)(a.M - a.m_base) * ((a.N + 3) / 4);
    wsovod::ProfScope prof(slot_fin, s, 0.0, (double)quads * 16.0 * (args.ksplit + 1));
    hipLaunchKernelGGL(splitk_finalize_kernel, dim3((unsigned)((quads + 255) / 256)), dim3(256), 0, s, args);
  }
  WS_CHECK_LAUNCH("wsovod_gemm_nt(256x256 8-phase)");
  return WSOVOD_OK;
}

}  // namespace wsovod_gemm

// MFMA contraction kernel of the hot path: C[M][N] = epilogue(sum_k A[m][k] * B[n][k]).
//
// One kernel structure serves every dense contraction on the path:
//   * nn.Linear forward  (A = activations, B = weight [out][in])         box_head.py:60-75
//   * nn.Linear dX       (A = dY,          B = weight^T shadow)
//   * nn.Linear dW       (A = dY^T,        B = X^T; reduction over proposals)
//   * conv + folded FrozenBN + ReLU (+ residual) as an implicit GEMM over NHWC input
//     (A rows are gathered per 3x3 tap; never materialised)         resnet_wsl.py:94-110
//   * region x text-embedding cosine-similarity GEMM with the 1/||x|| * T row scale
//     folded into the epilogue                          open_vocabulary_classifier.py:91-102
//
// CDNA4 mapping: 256 threads = 4 wavefronts (2x2), each wavefront owns a
// (BM/2)x(BN/2) block of 16x16 MFMA tiles.  A K-step is 128 BYTES of K per row for both
// dtypes (64 bf16 -> 2x v_mfma_f32_16x16x32_bf16, 32 fp32 -> 8x v_mfma_f32_16x16x4_f32),
// so one LDS image [rows][128 B] and one loader serve bf16 and exact-fp32 alike.
// LDS rows are XOR-swizzled in 16-B chunks (chunk ^= (row>>1)&7): ds_read_b128 fragment
// reads of 16 consecutive rows and the 8-lane ds_write_b128 groups are conflict-free
// (MI355X_MICROARCH.md, LDS table).  Global->LDS is register staged, software-pipelined
// one K-step ahead with LDS double buffering (one barrier per K-step).
// Workgroup ids are remapped so that the 8 XCDs each get a contiguous run of tiles that
// share the B (weight) slab in their private L2.
#include "gemm_common.h"
#include <type_traits>

namespace wsovod_gemm {

// X3 (T = bf16 only): the operands are bf16x2 (include/wsovod_hip.h); the launcher hands them over as bf16 matrices of
// twice the length (K, lda, ldb, Cin doubled), every 128-byte K-step row = [hi of 32 values | lo of the same 32], and a
// K-step computes b_hi*a_hi + b_lo*a_hi + b_hi*a_lo from the two fragment reads the bf16 form makes (see gemm8.hip).
template <typename T, int BM, int BN, bool CONV, int WM = 2, int WN = 2, bool DMA = false, int STAGES = 2, bool X3 = false>
__global__ __launch_bounds__(64 * WM * WN) void gemm_nt_kernel(const GemmArgs p) {
  static_assert(!X3 || sizeof(T) == 2, "X3 is a bf16 variant");
  static_assert(!X3 || STAGES == 2 || DMA, "the deep X3 pipeline stages by LDS-DMA");
  constexpr int EPC = Traits<T>::EPC;
  constexpr int BKE = Traits<T>::BKE;
  constexpr int NT = 64 * WM * WN;       // threads: WM x WN wavefronts
  constexpr int TM = BM / (WM * 16);     // 16x16 tiles per wavefront along M
  constexpr int TN = BN / (WN * 16);
  constexpr int LR = NT / 8;             // rows staged per loader pass (8 x 16-B chunks per row)
  constexpr int RA = BM / LR;            // 16-B chunks each thread stages per K-step (A)
  constexpr int RB = BN / LR;
  static_assert(BM % LR == 0 && BN % LR == 0, "tile rows must be a multiple of the loader pass");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sA = smem;
  char* sB = smem + STAGES * BM * 128;

  // ---- XCD-aware tile id: blocks b, b+8, ... share an XCD; give each XCD a contiguous
  // run of tile ids (M fastest), i.e. tiles that stream the same weight slab. Bijective.
  const int nwg = p.tiles_m * p.tiles_n;
  int wg;
  {
    const int bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  // Grouped order: ids sweep `group_m` M-tiles for every N-tile before moving on, so the contiguous
  // run of ids one XCD executes is a near-square block of the tile grid and re-reads neither operand
  // more than ~sqrt(run) times through its private L2 (measured: FETCH_SIZE, profiles/).
  const int group_size = p.group_m * p.tiles_n;
  const int group_id = wg / group_size;
  const int first_m = group_id * p.group_m;
  const int gm = min(p.tiles_m - first_m, p.group_m);
  const int in_group = wg - group_id * group_size;
  const int tile_m = first_m + in_group % gm;
  const int tile_n = in_group / gm;
  const int m0 = p.m_base + tile_m * BM, n0 = tile_n * BN;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int lrow = tid >> 3;
  // DMA staging writes LDS linearly (wave base + lane*16 B), so the XOR swizzle moves to the SOURCE:
  // the lane that lands in slot (tid&7) of its row fetches logical chunk slot ^ ((row>>1)&7).
  // (LR is a multiple of 16, so the swizzle term is the same for every loader pass.)
  const int lchunk = DMA ? ((tid & 7) ^ ((lrow >> 1) & 7)) : (tid & 7);

  // ---- per-thread loader state.  All global reads are raw buffer loads: a lane whose row
  // or K-chunk is out of range gets voffset = -1, which the hardware range check turns into
  // zeros -- no select on the loaded data, so the loads stay in flight under the MFMAs.
  typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
  const int esz = (int)sizeof(T);
  __amdgpu_buffer_rsrc_t rsrcA, rsrcB;
  int a_off[RA];  // byte offset of this lane's chunk at k = 0 (plain) / of the image (conv); <0 = invalid row
  int hi0[RA], wi0[RA];
  [[maybe_unused]] __amdgpu_buffer_rsrc_t rsrcA2;
  [[maybe_unused]] int pix2_off[RA];  // conv + fused shortcut: this lane's chunk of its output pixel in A2, <0 = row past M
  if (CONV) {
    rsrcA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, (int)p.a_bytes, 0x00020000);
    rsrcA2 = __builtin_amdgcn_make_buffer_rsrc((void*)(p.A2 ? p.A2 : p.A), 0, (int)(p.A2 ? p.a2_bytes : 0), 0x00020000);
  } else {
    const long long rows = min(BM, p.M - m0);
    rsrcA = __builtin_amdgcn_make_buffer_rsrc((void*)(p.A + (long long)m0 * p.lda * esz), 0,
                                              (int)(rows * p.lda * esz), 0x00020000);
  }
  {
    const long long rows = min(BN, p.N - n0);
    rsrcB = __builtin_amdgcn_make_buffer_rsrc((void*)(p.B + (long long)n0 * p.ldb * esz), 0,
                                              (int)(rows * p.ldb * esz), 0x00020000);
  }
#pragma unroll
  for (int i = 0; i < RA; ++i) {
    const int m = m0 + lrow + LR * i;
    const bool ok = m < p.M;
    if (CONV) {
      const int hw = p.Ho * p.Wo;
      const int mm = ok ? m : 0;
      const int img = mm / hw;
      const int rem = mm - img * hw;
      const int ho = rem / p.Wo;
      const int wo = rem - ho * p.Wo;
      hi0[i] = ok ? ho * p.stride - p.pad : -(1 << 28);  // invalid row: every tap falls outside
      wi0[i] = wo * p.stride - p.pad;
      a_off[i] = (img * p.H * p.W * p.Cin + lchunk * EPC) * esz;
      pix2_off[i] = ok ? (((img * p.Ho + ho) * p.Wo + wo) * p.Cin2 + lchunk * EPC) * esz : -1;
    } else {
      hi0[i] = wi0[i] = 0;
      a_off[i] = ok ? (int)(((long long)(lrow + LR * i) * p.lda + lchunk * EPC) * esz) : -1;
    }
  }
  // conv: everything per-lane about a tap is hoisted out of the K loop -- the byte offset of the lane's pixel with the
  // filter at its top-left position, and one validity bit per tap (image bounds / rows past M).  A K-step then costs one
  // add of a wave-uniform tap offset and one bit test per staged chunk instead of two integer multiplies and four compares.
  int pix_off[RA];
  unsigned vmask[RA];
  if (CONV) {
#pragma unroll
    for (int i = 0; i < RA; ++i) {
      pix_off[i] = hi0[i] > -(1 << 27) ? a_off[i] + ((hi0[i] * p.W + wi0[i]) * p.Cin) * esz : 0;  // rows past M: unused
      // branch-free in KH + KW steps (valid filter rows x valid filter columns; a row past M: hi0 = -2^28), as gemm8.hip
      unsigned rowm = 0, colm = 0;
      for (int r = 0; r < p.KH; ++r) rowm |= (unsigned)((unsigned)(hi0[i] + r * p.dil) < (unsigned)p.H) << r;
      for (int q = 0; q < p.KW; ++q) colm |= (unsigned)((unsigned)(wi0[i] + q * p.dil) < (unsigned)p.W) << q;
      unsigned mk = 0;
      for (int r = 0; r < p.KH; ++r) mk |= ((rowm >> r) & 1u) ? (colm << (r * p.KW)) : 0u;
      vmask[i] = mk;
    }
  }
  int b_off[RB];
#pragma unroll
  for (int i = 0; i < RB; ++i) {
    // LDS row (tile j = rho >> 4, tile row f = rho & 15 inside a wavefront's BN/WN columns) is fed from B row
    // 4*TN*(f>>2) + 4*j + (f&3): after the (operand-swapped) MFMAs a lane owns 4*TN CONSECUTIVE output columns, so a
    // wavefront writes 16*TN contiguous elements per output row; the LDS image and its reads are unchanged.
    constexpr int WCOLS = BN / WN;
    const int r_lds = lrow + LR * i;
    const int rho = r_lds % WCOLS;
    const int src = r_lds - rho + 4 * TN * ((rho & 15) >> 2) + 4 * (rho >> 4) + (rho & 3);
    b_off[i] = n0 + src < p.N ? (int)(((long long)src * p.ldb + lchunk * EPC) * esz) : -1;
  }

  const int nk = (p.K + BKE - 1) / BKE;

  auto load_global = [&](int kt, u32x4 (&ra)[RA], u32x4 (&rb)[RB]) {
    int kbase = kt * BKE;
    if (CONV) {
      // K-steps walk (channel chunk, filter tap) with the TAP innermost: the KH*KW taps of one 64-channel chunk re-read
      // the same input pixels shifted by the dilation, so the re-reads come one K-step after each other (the chunk's
      // slice of the tile block, ~1 MB per XCD, stays in L2) instead of one full channel sweep (~4 MB) apart.
      const int taps = p.KH * p.KW;
      const int nk_main = taps * (p.Cin / BKE);
      if (kt >= nk_main) {  // fused 1x1 shortcut: K-steps past the filter read the second input at the output pixel
        const int c2 = (kt - nk_main) * BKE;
        kbase = taps * p.Cin + c2;
#pragma unroll
        for (int i = 0; i < RA; ++i)
          ra[i] = __builtin_amdgcn_raw_buffer_load_b128(rsrcA2, pix2_off[i] >= 0 ? pix2_off[i] + c2 * esz : -1, 0, 0);
      } else {
      const int chunk = kt / taps;
      const int tap = kt - chunk * taps;
      const int c0 = chunk * BKE;
      const int r = tap / p.KW;
      const int q = tap - r * p.KW;
      kbase = tap * p.Cin + c0;  // position of this K-step in the weight rows ([kh][kw][Cin])
      const int delta = (((r * p.W + q) * p.dil) * p.Cin + c0) * esz;  // wave-uniform
#pragma unroll
      for (int i = 0; i < RA; ++i)
        ra[i] = __builtin_amdgcn_raw_buffer_load_b128(rsrcA, ((vmask[i] >> tap) & 1u) ? pix_off[i] + delta : -1, 0, 0);
      }
    }
    const bool k_ok = kbase + lchunk * EPC < p.K;
    if (!CONV) {
#pragma unroll
      for (int i = 0; i < RA; ++i)
        ra[i] = __builtin_amdgcn_raw_buffer_load_b128(rsrcA, (k_ok && a_off[i] >= 0) ? a_off[i] + kbase * esz : -1, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < RB; ++i)
      rb[i] = __builtin_amdgcn_raw_buffer_load_b128(rsrcB, (k_ok && b_off[i] >= 0) ? b_off[i] + kbase * esz : -1, 0, 0);
  };
  auto store_lds = [&](int buf, const u32x4 (&ra)[RA], const u32x4 (&rb)[RB]) {
    char* dA = sA + buf * BM * 128;
    char* dB = sB + buf * BN * 128;
#pragma unroll
    for (int i = 0; i < RA; ++i) *(u32x4*)(dA + lds_off(lrow + LR * i, lchunk)) = ra[i];
#pragma unroll
    for (int i = 0; i < RB; ++i) *(u32x4*)(dB + lds_off(lrow + LR * i, lchunk)) = rb[i];
  };

  // LDS-direct staging (buffer_load ... lds): no staging VGPRs, no ds_write pass.  Each wave
  // instruction lands 8 rows x 128 B contiguously at a wave-uniform LDS base.
  typedef __attribute__((address_space(3))) void lds_void [[maybe_unused]];
  [[maybe_unused]] const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  auto stage_dma = [&](int kt, int buf) {
#if defined(__HIP_DEVICE_COMPILE__)  // LDS address-space pointers / LDS-DMA builtins exist in the device pass only
    int kbase = kt * BKE;
    char* dA = sA + buf * BM * 128 + wave_u * 1024;
    char* dB = sB + buf * BN * 128 + wave_u * 1024;
    int tap = 0, delta = 0;
    bool second = false;  // fused 1x1 shortcut K-steps (wave-uniform)
    if (CONV) {  // (channel chunk, tap) order, tap innermost: see load_global
      const int taps = p.KH * p.KW;
      const int nk_main = taps * (p.Cin / BKE);
      second = kt >= nk_main;
      if (second) {
        delta = (kt - nk_main) * BKE * esz;
        kbase = taps * p.Cin + (kt - nk_main) * BKE;
      } else {
        const int chunk = kt / taps;
        tap = kt - chunk * taps;
        const int c0 = chunk * BKE;
        const int r = tap / p.KW;
        const int q = tap - r * p.KW;
        kbase = tap * p.Cin + c0;
        delta = (((r * p.W + q) * p.dil) * p.Cin + c0) * esz;  // wave-uniform
      }
    }
    const bool k_ok = kbase + lchunk * EPC < p.K;
    if (CONV && second) {
#pragma unroll
      for (int i = 0; i < RA; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcA2, (lds_void*)(dA + LR * i * 128), 16,
                                                 pix2_off[i] >= 0 ? pix2_off[i] + delta : -1, 0, 0, 0);
    } else if (CONV) {
#pragma unroll
      for (int i = 0; i < RA; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcA, (lds_void*)(dA + LR * i * 128), 16,
                                                 ((vmask[i] >> tap) & 1u) ? pix_off[i] + delta : -1, 0, 0, 0);
    } else {
#pragma unroll
      for (int i = 0; i < RA; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcA, (lds_void*)(dA + LR * i * 128), 16,
                                                 (k_ok && a_off[i] >= 0) ? a_off[i] + kbase * esz : -1, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < RB; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcB, (lds_void*)(dB + LR * i * 128), 16,
                                               (k_ok && b_off[i] >= 0) ? b_off[i] + kbase * esz : -1, 0, 0, 0);
#endif
  };

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int frow = lane & 15, fq = lane >> 4;
  auto compute = [&](int cur) {
    const char* cA = sA + cur * BM * 128 + (wm * (BM / WM)) * 128;
    const char* cB = sB + cur * BN * 128 + (wn * (BN / WN)) * 128;
    if constexpr (X3 && DMA && STAGES >= 3) {
      // round 6, small batches: the X3 products behind the DEEP LDS-DMA pipeline (STAGES - 1 K-steps in flight).  At 1 - 2
      // images per step the 128x64 / 64x64 grids are one round of workgroups whose K-steps each wait out a whole DMA
      // round trip (res5 of one image: 144 K-steps x 0.86 us with 0.18 us of MFMAs in each); with two K-steps in flight
      // the trip is shared.  Fragment reads are asm (hipcc must not see them: it would drain the ring with vmcnt(0)),
      // retired by an explicit lgkmcnt(0) that the reads' destinations are tied to; same products in the same order as
      // the two-stage form (bit-identical results).
#if defined(__HIP_DEVICE_COMPILE__)
      typedef __attribute__((address_space(3))) const char lds_cchar;
      const unsigned baseA = (unsigned)(size_t)(lds_cchar*)cA, baseB = (unsigned)(size_t)(lds_cchar*)cB;
      u32x4 af[TM], bfr[TN];
      auto rd = [&](u32x4& dst, unsigned base, int row, int chunk) {
        asm volatile("ds_read_b128 %0, %1" : "=v"(dst) : "v"(base + row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4)));
      };
      // all four fragment sets of the K-step are requested up front (one LDS round trip per K-step instead of three: at two
      // wavefronts per SIMD nothing else hides them); the 16-wavefront tiles cannot afford the registers, these can
      u32x4 al[TM], bh[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) rd(af[i], baseA, i * 16 + frow, fq);        // a_hi
#pragma unroll
      for (int j = 0; j < TN; ++j) rd(bfr[j], baseB, j * 16 + frow, fq + 4);   // b_lo
#pragma unroll
      for (int j = 0; j < TN; ++j) rd(bh[j], baseB, j * 16 + frow, fq);        // b_hi
#pragma unroll
      for (int i = 0; i < TM; ++i) rd(al[i], baseA, i * 16 + frow, fq + 4);    // a_lo
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int i = 0; i < TM; ++i) asm volatile("" : "+v"(af[i]), "+v"(al[i]));
#pragma unroll
      for (int j = 0; j < TN; ++j) asm volatile("" : "+v"(bfr[j]), "+v"(bh[j]));
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)  // b_lo * a_hi
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, bfr[j]),
                                                              __builtin_bit_cast(bf16x8, af[i]), acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)  // b_hi * a_hi
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, bh[j]),
                                                              __builtin_bit_cast(bf16x8, af[i]), acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)  // b_hi * a_lo
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, bh[j]),
                                                              __builtin_bit_cast(bf16x8, al[i]), acc[i][j], 0, 0, 0);
#endif
    }
    else if constexpr (X3) {
      // Two fragment sets live at a time (a third costs the 16-wavefront tile spills at its 128-register budget): b_lo and
      // a_hi first, then b_hi takes b_lo's registers, then a_lo takes a_hi's.
      u32x4 af[TM], bfr[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int row = i * 16 + frow;
        af[i] = *(const u32x4*)(cA + row * 128 + ((fq ^ ((row >> 1) & 7)) << 4));
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int row = j * 16 + frow;
        bfr[j] = *(const u32x4*)(cB + row * 128 + (((fq + 4) ^ ((row >> 1) & 7)) << 4));
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)  // b_lo * a_hi
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, bfr[j]),
                                                              __builtin_bit_cast(bf16x8, af[i]), acc[i][j], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int row = j * 16 + frow;
        bfr[j] = *(const u32x4*)(cB + row * 128 + ((fq ^ ((row >> 1) & 7)) << 4));
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)  // b_hi * a_hi
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, bfr[j]),
                                                              __builtin_bit_cast(bf16x8, af[i]), acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int row = i * 16 + frow;
        af[i] = *(const u32x4*)(cA + row * 128 + (((fq + 4) ^ ((row >> 1) & 7)) << 4));
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)  // b_hi * a_lo
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, bfr[j]),
                                                              __builtin_bit_cast(bf16x8, af[i]), acc[i][j], 0, 0, 0);
    } else {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int chunk = fq + 4 * ks;
      u32x4 af[TM], bfr[TN];
      if constexpr (DMA && STAGES >= 3) {
#if defined(__HIP_DEVICE_COMPILE__)
        // Deep pipeline: hipcc would put `s_waitcnt vmcnt(0)` in front of any ds_read it can see while
        // LDS-DMA is in flight (it cannot tell the ring slots apart) and so drain the ring every K-step.
        // The fragment reads are therefore issued as asm (invisible to that pass) and retired by an
        // explicit lgkmcnt(0) that names every destination (cdna_hip_programming.md 5.7 item 1, form ii).
        typedef __attribute__((address_space(3))) const char lds_cchar;
        const unsigned baseA = (unsigned)(size_t)(lds_cchar*)cA, baseB = (unsigned)(size_t)(lds_cchar*)cB;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const int row = i * 16 + frow;
          asm volatile("ds_read_b128 %0, %1" : "=v"(af[i]) : "v"(baseA + row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4)));
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int row = j * 16 + frow;
          asm volatile("ds_read_b128 %0, %1" : "=v"(bfr[j]) : "v"(baseB + row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4)));
        }
        static_assert(X3 || (TM == 4 && TN == 4), "the lgkmcnt wait statement names 8 fragment registers");
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(af[0]), "+v"(af[1]), "+v"(af[2]), "+v"(af[3]), "+v"(bfr[0]), "+v"(bfr[1]), "+v"(bfr[2]),
                       "+v"(bfr[3]));
        __builtin_amdgcn_sched_barrier(0);
#endif
      } else {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const int row = i * 16 + frow;  // the wavefront's row base is a multiple of 16: swizzle term unchanged
          af[i] = *(const u32x4*)(cA + row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4));
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int row = j * 16 + frow;
          bfr[j] = *(const u32x4*)(cB + row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4));
        }
      }
      if constexpr (sizeof(T) == 2) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(  // operands swapped: see the epilogue
                __builtin_bit_cast(bf16x8, bfr[j]), __builtin_bit_cast(bf16x8, af[i]), acc[i][j], 0, 0, 0);
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(
                  __builtin_bit_cast(f32x4, bfr[j])[e], __builtin_bit_cast(f32x4, af[i])[e], acc[i][j], 0, 0, 0);
      }
    }
    }  // (!X3)
  };

  if constexpr (DMA && STAGES >= 3) {
    // ---- deep LDS-direct pipeline: STAGES-1 K-steps of DMA in flight.  A counted s_waitcnt retires only
    // the oldest stage, a raw s_barrier (no vmcnt(0) drain) publishes it, the freed buffer is re-staged at
    // once, then the MFMAs run -- one barrier per K-step, loads span STAGES-1 compute phases.
    //   RAW: own vmcnt wait, then the barrier every wave passes only after ITS wait => stage kt is complete.
    //   WAR: the buffer re-staged after the barrier of step kt was last read in compute(kt-1), which every
    //        wave finished before reaching that barrier.
    [[maybe_unused]] constexpr int LPS = RA + RB;  // DMA instructions per thread per stage
#pragma unroll
    for (int st = 0; st < STAGES - 1; ++st)
      if (st < nk) stage_dma(st, st);
    for (int kt = 0; kt < nk; ++kt) {
      [[maybe_unused]] const int newer = min(nk - 1 - kt, STAGES - 2);  // stages issued after kt that may still be in flight
#if defined(__HIP_DEVICE_COMPILE__)
      if (newer >= STAGES - 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((STAGES - 2) * LPS) : "memory");
      else if (STAGES > 3 && newer == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPS) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
      __builtin_amdgcn_s_barrier();
      if (kt + STAGES - 1 < nk) stage_dma(kt + STAGES - 1, (kt + STAGES - 1) % STAGES);
      compute(kt % STAGES);
    }
  } else if constexpr (DMA) {
    // ---- LDS-direct pipeline: next K-step's DMA is issued before the MFMAs of the current one;
    // the barrier (with the vmcnt(0) hipcc puts in front of it) retires it.  One barrier per K-step.
    stage_dma(0, 0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
      const int cur = kt & 1;
      if (kt + 1 < nk) stage_dma(kt + 1, cur ^ 1);
      compute(cur);
      __syncthreads();
    }
  } else {
  // ---- software pipeline: two K-steps of global loads in flight (register sets 0/1),
  // LDS double buffered, one barrier per K-step.
  u32x4 ra0[RA], rb0[RB], ra1[RA], rb1[RB];
  load_global(0, ra0, rb0);
  if (nk > 1) load_global(1, ra1, rb1);
  store_lds(0, ra0, rb0);
  __syncthreads();
  for (int kt = 0; kt < nk; kt += 2) {
    // even step: LDS[0] holds kt, set1 holds kt+1 (in flight), set0 is free
    if (kt + 2 < nk) load_global(kt + 2, ra0, rb0);
    compute(0);
    if (kt + 1 < nk) store_lds(1, ra1, rb1);
    __syncthreads();
    if (kt + 1 >= nk) break;
    // odd step: LDS[1] holds kt+1, set0 holds kt+2 (in flight), set1 is free
    if (kt + 3 < nk) load_global(kt + 3, ra1, rb1);
    compute(1);
    if (kt + 2 < nk) store_lds(0, ra0, rb0);
    __syncthreads();
  }
  }

  // ---- epilogue (fp32).  The MFMAs take the B fragment as their first operand and the B rows were permuted at
  // staging, so lane (frow, fq) holds, for row tile i, output row m = .. + frow and the 4*TN consecutive columns
  // n = ncol + 4*j + r: row-major outputs move as 16-byte stores, 16*TN contiguous elements per row and wavefront.
  const float keep_scale = p.dropout_p > 0.f ? 1.0f / (1.0f - p.dropout_p) : 1.0f;
  [[maybe_unused]] const unsigned long long dseed = p.dropout_p > 0.f ? WS_DROPOUT_SEED(p) : 0ull;
  [[maybe_unused]] const unsigned dthr = dropout_threshold(p.dropout_p);
  const bool vec_c = p.C && (p.dtype_c == WSOVOD_BF16X2 ? vec4_ok(p.C, p.ldc, p.dtype_c)
                                                        : (p.ldc & 7) == 0 && ((uintptr_t)p.C & 15) == 0);
  const bool vec_r = !p.residual || (p.dtype_r == WSOVOD_BF16X2 ? vec4_ok(p.residual, p.ldr, p.dtype_r)
                                                                 : (p.ldr & 7) == 0 && ((uintptr_t)p.residual & 15) == 0);
  const int mrow = m0 + wm * (BM / WM) + frow;
  const int ncol = n0 + wn * (BN / WN) + 4 * TN * fq;
  // Fast path (bias / residual / ReLU / dropout, aligned row-major output, all tile columns in range): feature tests
  // hoisted out of the element loops, bias fetched once per column tile, residual and output as vector accesses.
  if (vec_c && vec_r && ((uintptr_t)p.bias & 15) == 0 && !p.Ct && !p.row_scale && !p.group_add && !p.mask_src &&
      !p.accumulate && n0 + BN <= p.N && (TN % 2 == 0 || p.dtype_c != WSOVOD_BF16)) {
    f32x4 b4[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) b4[j] = p.bias ? *(const f32x4*)(p.bias + ncol + 4 * j) : f32x4{0.f, 0.f, 0.f, 0.f};
    const float lo = p.relu ? 0.f : -__builtin_inff();
    const bool drop = p.dropout_p > 0.f, has_res = p.residual != nullptr;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const long long mm = mrow + i * 16;
      if (mm >= p.M) continue;
      const long long base = mm * p.ldc + ncol;
      f32x4 x[TN];
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        x[j] = acc[i][j] * p.alpha + b4[j];
        if (has_res) x[j] += load4_as_f32(p.residual, mm, p.ldr, ncol + 4 * j, p.dtype_r);
        x[j] = f32x4{fmaxf(x[j][0], lo), fmaxf(x[j][1], lo), fmaxf(x[j][2], lo), fmaxf(x[j][3], lo)};
        if (drop) {
          const unsigned long long dz = dropout_quad(dseed, mm, p.N, ncol + 4 * j);
#pragma unroll
          for (int r = 0; r < 4; ++r) x[j][r] = dropout_keep(dz, r, dthr) ? x[j][r] * keep_scale : 0.f;
        }
      }
      if (p.dtype_c == WSOVOD_BF16X2) {
#pragma unroll
        for (int j = 0; j < TN; ++j) store4_from_f32(p.C, mm, p.ldc, ncol + 4 * j, WSOVOD_BF16X2, x[j]);
      } else if (p.dtype_c == WSOVOD_BF16) {
#pragma unroll
        for (int j = 0; j + 1 < TN; j += 2)
          *(bf16x8*)((bf16_t*)p.C + base + 4 * j) =
              bf16x8{(bf16_t)x[j][0],     (bf16_t)x[j][1],     (bf16_t)x[j][2],     (bf16_t)x[j][3],
                     (bf16_t)x[j + 1][0], (bf16_t)x[j + 1][1], (bf16_t)x[j + 1][2], (bf16_t)x[j + 1][3]};
      } else {
#pragma unroll
        for (int j = 0; j < TN; ++j) *(f32x4*)((float*)p.C + base + 4 * j) = x[j];
      }
    }
    return;
  }
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int m = mrow + i * 16;
    if (m >= p.M) continue;
    const float rs = p.row_scale ? p.row_scale[m] : 1.f;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int nb = ncol + 4 * j;
      if (nb >= p.N) continue;
      float v[4];
      const unsigned long long dz = p.dropout_p > 0.f ? dropout_quad(dseed, m, p.N, nb) : 0ull;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = nb + r;
        float x = acc[i][j][r] * p.alpha;
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
        if (vec_c && nb + 3 < p.N) {
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
    }
  }
}

// ---------------------------------------------------------------------------------
// 3x3 / stride 1 / pad 1 / Cin = Cout = 64 convolution (stem conv2, conv3 and the four res2 convs,
// resnet_wsl.py:386-405,59-79) in bf16.  With only 64 output channels the generic implicit GEMM stages
// every input pixel nine times (once per tap) for a 64-wide B tile: 51 FLOP per staged byte, L2-bound.
// Here a workgroup owns an 8x32 block of output pixels of one image: the 10x34 input halo patch is
// DMA-staged into LDS ONCE (zero-filled outside the image by the buffer range check) and all nine taps
// read their A fragments from it at shifted pixel offsets; the 64x64 weight slice of each tap streams
// through a double-buffered 8-KiB LDS tile.  LDS rows are 128 B (64 bf16 channels of one pixel) with the
// same XOR swizzle as the GEMM image, keyed by the patch pixel index, so 16 consecutive pixels read
// conflict-free.  4 wavefronts, each 64 pixels (2 rows x 32) x 64 channels = 4x4 MFMA tiles.
// ---------------------------------------------------------------------------------
constexpr int C64_TH = 8, C64_TW = 32, C64_PW = C64_TW + 2, C64_PH = C64_TH + 2, C64_NPIX = C64_PW * C64_PH;
constexpr int C64_PATCH_BYTES = ((C64_NPIX + 31) / 32 * 32) * 128;  // padded to whole 32-pixel DMA passes

// Epilogue of the 64-channel 3x3 kernels (one-tile and persistent form): bias + residual + ReLU (+ fused 2x2 max pool).
__device__ __forceinline__ void c64_load_bias(const GemmArgs& p, const int fq, f32x4 (&b4)[4]) {
  // this lane's 16 output channels (16*fq ..): loaded once, ahead of the MFMAs, so the epilogue never waits for them
  const bool al = ((uintptr_t)p.bias & 15) == 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float* b = p.bias + 16 * fq + 4 * j;
    b4[j] = !p.bias ? f32x4{0.f, 0.f, 0.f, 0.f} : al ? *(const f32x4*)b : f32x4{b[0], b[1], b[2], b[3]};
  }
}

// Residual rows of this lane's four pixels (clamped into the image: an outside pixel's row is loaded and never used),
// issued ahead of the MFMA loop by the persistent kernel so that the epilogue does not sit out their latency.
__device__ __forceinline__ bool c64_residual_preloadable(const GemmArgs& p) {
  return p.residual && p.dtype_r == WSOVOD_BF16 && (p.ldr & 7) == 0 && ((uintptr_t)p.residual & 15) == 0;
}
__device__ __forceinline__ void c64_load_residual(const GemmArgs& p, const int img, const int y0, const int x0,
                                                  const int wave, const int frow, const int fq, bf16x8 (&rres)[4][2]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int y = min(y0 + wave * 2 + (i >> 1), p.H - 1), x = min(x0 + (i & 1) * 16 + frow, p.W - 1);
    const long long m = ((long long)img * p.H + y) * p.W + x;
    const bf16x8* rp = (const bf16x8*)((const bf16_t*)p.residual + m * p.ldr + 16 * fq);
    rres[i][0] = rp[0];
    rres[i][1] = rp[1];
  }
}

template <bool PRE = false>
__device__ __forceinline__ void c64_epilogue(const GemmArgs& p, const f32x4 (&acc)[4][4], const f32x4 (&bias4)[4],
                                             const int img, const int y0, const int x0, const int wave, const int frow,
                                             const int fq, const bf16x8 (*rres)[2] = nullptr) {
  // epilogue: bias + residual + ReLU.  Operands were swapped in the MFMA and the weight rows permuted at staging, so
  // lane (frow, fq) holds, for pixel group i, the 16 consecutive output channels 16*fq + 4*j + r of pixel frow: its
  // 32 bytes (bf16) go out as two 16-byte stores and the four lanes of a pixel cover its whole 128-byte row.
  const bool vec = (p.ldc & 7) == 0 && ((uintptr_t)p.C & 15) == 0 && ((uintptr_t)p.bias & 15) == 0 &&
                   (!p.residual || ((p.ldr & 7) == 0 && ((uintptr_t)p.residual & 15) == 0));
  const float lo = p.relu ? 0.f : -__builtin_inff();
  const int n0c = 16 * fq;
  if (p.pool) {
    // Fused MaxPool2d(2, 2) (stem tail resnet_wsl.py:418-420, block tail :85-92,107-108): the wavefront's two image rows
    // are the two rows of one pooled row (y0 and wave*2 are even), the horizontal partner is the neighbouring lane
    // (frow ^ 1, same channels).  Values are rounded to bf16 first, as the unfused conv output was, so the pooled map has
    // the same bits as conv -> maxpool2x2_nhwc; the full-resolution map is never written.  (launcher: bf16, aligned)
    const int Hp = p.H >> 1, Wp = p.W >> 1;
#pragma unroll
    for (int ih = 0; ih < 2; ++ih) {
      float best[16];
#pragma unroll
      for (int e = 0; e < 16; ++e) best[e] = -__builtin_inff();
#pragma unroll
      for (int iv = 0; iv < 2; ++iv) {
        const int i = ih + 2 * iv;
        const int y = min(y0 + wave * 2 + iv, p.H - 1), x = min(x0 + ih * 16 + frow, p.W - 1);  // clamped: unused if outside
        const long long m = ((long long)img * p.H + y) * p.W + x;
        float v[16];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) v[4 * j + r] = acc[i][j][r] * p.alpha + bias4[j][r];
        if (p.residual) {
          const bf16x8* rp = (const bf16x8*)((const bf16_t*)p.residual + m * p.ldr + n0c);
          const bf16x8 r0 = PRE ? rres[i][0] : rp[0], r1 = PRE ? rres[i][1] : rp[1];
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            v[e] += (float)r0[e];
            v[8 + e] += (float)r1[e];
          }
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) best[e] = fmaxf(best[e], (float)(bf16_t)fmaxf(v[e], lo));
      }
#pragma unroll
      for (int e = 0; e < 16; ++e) best[e] = fmaxf(best[e], __shfl_xor(best[e], 1));
      const int py = (y0 >> 1) + wave, px = (x0 + ih * 16 + frow) >> 1;
      if ((frow & 1) == 0 && py < Hp && px < Wp) {
        bf16x8 o0, o1;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          o0[e] = (bf16_t)best[e];
          o1[e] = (bf16_t)best[8 + e];
        }
        bf16x8* dst = (bf16x8*)((bf16_t*)p.C + (((long long)img * Hp + py) * Wp + px) * p.ldc + n0c);
        dst[0] = o0;
        dst[1] = o1;
      }
    }
    return;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int y = y0 + wave * 2 + (i >> 1);
    const int x = x0 + (i & 1) * 16 + frow;
    if (y >= p.H || x >= p.W) continue;
    const long long m = ((long long)img * p.H + y) * p.W + x;
    float v[16];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) v[4 * j + r] = acc[i][j][r] * p.alpha;
    if (vec && p.dtype_c == WSOVOD_BF16 && (!p.residual || p.dtype_r == WSOVOD_BF16)) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) v[4 * j + r] += bias4[j][r];
      if (p.residual) {
        const bf16x8* rp = (const bf16x8*)((const bf16_t*)p.residual + m * p.ldr + n0c);
        const bf16x8 r0 = PRE ? rres[i][0] : rp[0], r1 = PRE ? rres[i][1] : rp[1];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          v[e] += (float)r0[e];
          v[8 + e] += (float)r1[e];
        }
      }
      bf16x8 o0, o1;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        o0[e] = (bf16_t)fmaxf(v[e], lo);
        o1[e] = (bf16_t)fmaxf(v[8 + e], lo);
      }
      bf16x8* dst = (bf16x8*)((bf16_t*)p.C + m * p.ldc + n0c);
      dst[0] = o0;
      dst[1] = o1;
    } else {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        float u = v[e] + (p.bias ? p.bias[n0c + e] : 0.f);
        if (p.residual) u += load_as_f32(p.residual, m, p.ldr, n0c + e, p.dtype_r);
        store_from_f32(p.C, m, p.ldc, n0c + e, p.dtype_c, fmaxf(u, lo));
      }
    }
  }
}

__global__ __launch_bounds__(256) void conv3x3_c64_kernel(const GemmArgs p, int tiles_x, int tiles_y) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sP = smem;                    // halo patch [pixel][128 B]
  char* sW = smem + C64_PATCH_BYTES;  // 2 x [64 cout][128 B] weight slices
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tpi = tiles_x * tiles_y;
  const int img = blockIdx.x / tpi;
  const int t = blockIdx.x - img * tpi;
  const int ty = t / tiles_x, tx = t - ty * tiles_x;
  const int y0 = ty * C64_TH, x0 = tx * C64_TW;
  typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
  [[maybe_unused]] const __amdgpu_buffer_rsrc_t rsrcA =
      __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, (int)p.a_bytes, 0x00020000);
  [[maybe_unused]] const __amdgpu_buffer_rsrc_t rsrcB =
      __builtin_amdgcn_make_buffer_rsrc((void*)p.B, 0, (int)(64 * p.ldb * 2), 0x00020000);
  [[maybe_unused]] const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  typedef __attribute__((address_space(3))) void lds_void [[maybe_unused]];

  auto stage_weights = [&](int tap, int buf) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = (tid >> 3) + 32 * i;                 // LDS row = MFMA tile j = row >> 4, tile row f = row & 15
      const int chunk = (tid & 7) ^ ((row >> 1) & 7);      // swizzle on the source
      // LDS row (j, f) holds output channel 16*(f>>2) + 4*j + (f&3): after the MFMA a lane owns 16 CONSECUTIVE
      // channels of its pixel (see the epilogue) while the fragment reads keep their conflict-free row pattern
      const int cout = 16 * ((row & 15) >> 2) + 4 * (row >> 4) + (row & 3);
      const int off = (int)((cout * p.ldb + tap * 64 + chunk * 8) * 2);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcB, (lds_void*)(sW + buf * 8192 + i * 4096 + wave_u * 1024), 16,
                                               off, 0, 0, 0);
    }
#endif
  };
#if defined(__HIP_DEVICE_COMPILE__)
  // halo patch: pixel q of the patch = image pixel (y0-1 + q/PW, x0-1 + q%PW); 8 pixels per wave DMA
  for (int base = 0; base < C64_NPIX; base += 32) {
    const int q = base + (tid >> 3);
    const int py = q / C64_PW, px = q - py * C64_PW;
    const int y = y0 - 1 + py, x = x0 - 1 + px;
    const bool ok = q < C64_NPIX && y >= 0 && y < p.H && x >= 0 && x < p.W;
    const int chunk = (tid & 7) ^ ((q >> 1) & 7);
    const int off = (((img * p.H + y) * p.W + x) * 64 + chunk * 8) * 2;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcA, (lds_void*)(sP + (base + wave_u * 8) * 128), 16, ok ? off : -1, 0,
                                             0, 0);
  }
#endif
  stage_weights(0, 0);
  __syncthreads();

  const int frow = lane & 15, fq = lane >> 4;
  f32x4 bias4[4];
  c64_load_bias(p, fq, bias4);
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
  for (int tap = 0; tap < 9; ++tap) {
    const int cur = tap & 1;
    if (tap + 1 < 9) stage_weights(tap + 1, cur ^ 1);
    const int r = tap / 3, s3 = tap - r * 3;
    const char* cW = sW + cur * 8192;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int chunk = fq + 4 * ks;
      u32x4 af[4], bfr[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int yy = wave * 2 + (i >> 1) + r;          // patch row of this 16-pixel group, shifted by the tap
        const int q = yy * C64_PW + (i & 1) * 16 + frow + s3;
        af[i] = *(const u32x4*)(sP + q * 128 + ((chunk ^ ((q >> 1) & 7)) << 4));
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int row = j * 16 + frow;
        bfr[j] = *(const u32x4*)(cW + row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4));
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, bfr[j]),  // swapped: a lane
                                                              __builtin_bit_cast(bf16x8, af[i]), acc[i][j], 0, 0, 0);  // holds 4 channels
    }
    __syncthreads();
  }
  c64_epilogue(p, acc, bias4, img, y0, x0, wave, frow, fq);
}

// ---------------------------------------------------------------------------------
// Persistent form of the 64-channel 3x3 kernel (round 2).  One workgroup per CU walks its tiles (8 x 32 output pixels
// each); what the one-tile kernel above pays per tile or per tap is paid once or hidden:
//   * all nine 64x64 weight slices are staged ONCE per workgroup and stay resident (72 KiB) -- no per-tap staging, no
//     per-tap barrier;
//   * two halo-patch buffers (2 x 43 KiB): the LDS-DMA of the next tile's patch is issued before the current tile's
//     MFMAs and lands behind them and the epilogue; one barrier per tile;
//   * fragment reads are inline asm (hipcc would otherwise drain the in-flight DMA with s_waitcnt vmcnt(0) in front of
//     every ds_read it can see), software-pipelined one (tap, k-half) step ahead of the MFMAs with counted lgkmcnt
//     waits; every LDS address is loop-invariant (36 + 4 registers), so the tile loop issues no address arithmetic.
// 160 KiB of LDS = 72 KiB weights + 2 x 43 KiB patches: one workgroup (4 wavefronts, one per SIMD) per CU.
// Same products in the same order as the one-tile kernel: bit-identical outputs.
// ---------------------------------------------------------------------------------
// Ablation build switch (tools/c64_ablate.sh; results in profiles/r02_c64_persistent.md, profiles/r03_c64_wreg.md),
// never set in the product build: bit 0 drops the fragment-read + MFMA loop, 1 the epilogue, 2 the patch staging of
// every tile but the first, 3 the fragment reads and their waits (MFMAs on stale registers; LDS-weights form only), 4
// only the waits (the same), 5 the vector-memory wait in front of the epilogue (register-weights form), 6 = per-phase
// s_memtime sums written to GemmArgs.partial (tools/c64_phases.py).  The `alpha` comparisons keep the dropped code
// reachable for the compiler, so the rest of the kernel compiles as in the product build.
#ifndef C64P_ABL
#define C64P_ABL 0
#endif
constexpr int C64P_W_BYTES = 9 * 8192;
constexpr int C64P_LDS_BYTES = C64P_W_BYTES + 2 * C64_PATCH_BYTES;

// WREG (the product form): the 72 weight fragments of a lane (9 taps x 2 k-halves x 4 channel tiles = 288 VGPRs) are read
// from the LDS image ONCE and stay in registers for every tile of the workgroup (one wavefront per SIMD: the 512-entry
// register file is all its own).  A (tap, k-half) step then reads only its 4 pixel fragments for 16 MFMAs -- 4 MFMAs per
// ds_read_b128 instead of 2, i.e. half of the LDS array's bandwidth at the full MFMA rate instead of all of it (the
// LDS-weights form saturates the array: profiles/r02_c64_persistent.md).  Same products in the same order.
template <bool WREG>
__global__ __launch_bounds__(256) void conv3x3_c64p_kernel(const GemmArgs p, int tiles_x, int tiles_y, int n_tiles) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sW = smem;                  // 9 x [64 cout][128 B]
  [[maybe_unused]] char* sP = smem + C64P_W_BYTES;   // 2 x halo patch [pixel][128 B]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int frow = lane & 15, fq = lane >> 4;
  const int tpi = tiles_x * tiles_y;
  typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
  [[maybe_unused]] const __amdgpu_buffer_rsrc_t rsrcA =
      __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, (int)p.a_bytes, 0x00020000);
  [[maybe_unused]] const __amdgpu_buffer_rsrc_t rsrcB =
      __builtin_amdgcn_make_buffer_rsrc((void*)p.B, 0, (int)(64 * p.ldb * 2), 0x00020000);
  [[maybe_unused]] const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  typedef __attribute__((address_space(3))) void lds_void [[maybe_unused]];
#if defined(__HIP_DEVICE_COMPILE__)
  // resident weights: the LDS image of the one-tile kernel's slice, nine times
#pragma unroll
  for (int tap = 0; tap < 9; ++tap)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = (tid >> 3) + 32 * i;
      const int chunk = (tid & 7) ^ ((row >> 1) & 7);
      const int cout = 16 * ((row & 15) >> 2) + 4 * (row >> 4) + (row & 3);
      const int off = (int)((cout * p.ldb + tap * 64 + chunk * 8) * 2);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcB, (lds_void*)(sW + tap * 8192 + i * 4096 + wave_u * 1024), 16, off, 0,
                                               0, 0);
    }
#endif
  // tile-invariant part of the patch staging: this thread's 11 halo pixels (row, column packed) and its 16-byte chunk
  constexpr int C64P_NLOAD = (C64_NPIX + 31) / 32;
  int pyx[C64P_NLOAD], pchunk[C64P_NLOAD];
#pragma unroll
  for (int k = 0; k < C64P_NLOAD; ++k) {
    const int q = k * 32 + (tid >> 3);
    const int py = q / C64_PW, px = q - py * C64_PW;
    pyx[k] = q < C64_NPIX ? (py << 16) | px : (0x4000 << 16);  // past the patch: a row no image has
    pchunk[k] = ((tid & 7) ^ ((q >> 1) & 7)) * 16;
  }
  auto stage_patch = [&](int tile, int buf) {
#if defined(__HIP_DEVICE_COMPILE__)
    const int img = tile / tpi;
    const int t = tile - img * tpi;
    const int ty = t / tiles_x, tx = t - ty * tiles_x;
    const int y0 = ty * C64_TH - 1, x0 = tx * C64_TW - 1;
    char* dst = sP + buf * C64_PATCH_BYTES + wave_u * 1024;
#pragma unroll
    for (int k = 0; k < C64P_NLOAD; ++k) {
      const int y = y0 + (pyx[k] >> 16), x = x0 + (pyx[k] & 0xffff);
      const bool ok = (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
      const int off = ((img * p.H + y) * p.W + x) * 128 + pchunk[k];
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcA, (lds_void*)(dst + k * 4096), 16, ok ? off : -1, 0, 0, 0);
    }
#endif
  };
  int tile = blockIdx.x;
  stage_patch(tile, 0);

  // loop-invariant LDS byte offsets of this lane's fragments (k-half 1 = the same offset ^ 64)
  unsigned aoff[9][4], boff[4];
#pragma unroll
  for (int tap = 0; tap < 9; ++tap) {
    const int r = tap / 3, s3 = tap - r * 3;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int q = (wave * 2 + (i >> 1) + r) * C64_PW + (i & 1) * 16 + frow + s3;
      aoff[tap][i] = (unsigned)(q * 128 + ((fq ^ ((q >> 1) & 7)) << 4));
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int row = j * 16 + frow;
    boff[j] = (unsigned)(row * 128 + ((fq ^ ((row >> 1) & 7)) << 4));
  }
#if defined(__HIP_DEVICE_COMPILE__)
  typedef __attribute__((address_space(3))) const char lds_cchar;
  const unsigned ldsW = (unsigned)(size_t)(lds_cchar*)sW, ldsP = (unsigned)(size_t)(lds_cchar*)sP;
#if (C64P_ABL & 8)
#define C64P_READ(dst, addr) asm volatile("; no read %0 %1" : "=v"(dst) : "v"(addr))
#define C64P_WAIT1(N, R) asm volatile("; no wait" : "+v"(R))
#define C64P_WAIT2(N, R0, R1) asm volatile("; no wait" : "+v"(R0), "+v"(R1))
#elif (C64P_ABL & 16)
#define C64P_READ(dst, addr) asm volatile("ds_read_b128 %0, %1" : "=v"(dst) : "v"(addr))
#define C64P_WAIT1(N, R) asm volatile("; no wait" : "+v"(R))
#define C64P_WAIT2(N, R0, R1) asm volatile("; no wait" : "+v"(R0), "+v"(R1))
#else
#define C64P_READ(dst, addr) asm volatile("ds_read_b128 %0, %1" : "=v"(dst) : "v"(addr))
#define C64P_WAIT1(N, R) asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(R))
#define C64P_WAIT2(N, R0, R1) asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(R0), "+v"(R1))
#endif
#else
  const unsigned ldsW = 0, ldsP = 0;
#define C64P_READ(dst, addr) (void)0
#define C64P_WAIT1(N, R) (void)0
#define C64P_WAIT2(N, R0, R1) (void)0
#endif
#define C64P_RD_B(S, J, TAP, KS) C64P_READ(fb[S][J], (ldsW + (TAP) * 8192 + boff[J]) ^ ((KS) ? 64u : 0u))
#define C64P_RD_A(S, I, TAP, KS) C64P_READ(fa[S][I], (pbase + aoff[TAP][I]) ^ ((KS) ? 64u : 0u))
#define C64P_MF(S, M)                                                                                              \
  acc[(M) >> 2][(M) & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fb[S][(M) & 3]),     \
                                                                   __builtin_bit_cast(bf16x8, fa[S][(M) >> 2]),    \
                                                                   acc[(M) >> 2][(M) & 3], 0, 0, 0)
#define C64P_SB __builtin_amdgcn_sched_barrier(0)
// The 8 fragment reads of a (tap, k-half) step always issue in the order  B0 A0 B1 B2 B3 A1 A2 A3  -- the order its
// MFMAs (m = 4*i + j uses A_i, B_j) first need them.
#define C64P_READ_STEP(S, TAP, KS)                                                                               \
  C64P_RD_B(S, 0, TAP, KS); C64P_RD_A(S, 0, TAP, KS); C64P_RD_B(S, 1, TAP, KS); C64P_RD_B(S, 2, TAP, KS);         \
  C64P_RD_B(S, 3, TAP, KS); C64P_RD_A(S, 1, TAP, KS); C64P_RD_A(S, 2, TAP, KS); C64P_RD_A(S, 3, TAP, KS)
// One step: 16 MFMAs on register set S; the NEXT step's reads (set N = 1 - S) go out one behind every second MFMA, so
// the four wavefronts load the LDS array evenly (one ds_read_b128 per wavefront per 16-cycle MFMA already saturates
// it) and each fragment is 10-14 MFMAs old when it is needed.  Every wait is counted: it leaves in flight exactly the
// reads issued after the fragment it is for (LDS reads return in order).
#define C64P_STEP(S, N, NTAP, NKS)                                                                                \
  C64P_WAIT2(6, fb[S][0], fa[S][0]); C64P_SB;                                                                      \
  C64P_MF(S, 0); C64P_RD_B(N, 0, NTAP, NKS); C64P_SB;                                                              \
  C64P_WAIT1(6, fb[S][1]); C64P_MF(S, 1); C64P_SB;                                                                 \
  C64P_WAIT1(5, fb[S][2]); C64P_MF(S, 2); C64P_RD_A(N, 0, NTAP, NKS); C64P_SB;                                     \
  C64P_WAIT1(5, fb[S][3]); C64P_MF(S, 3); C64P_SB;                                                                 \
  C64P_WAIT1(4, fa[S][1]); C64P_MF(S, 4); C64P_RD_B(N, 1, NTAP, NKS); C64P_SB;                                     \
  C64P_MF(S, 5); C64P_SB;                                                                                          \
  C64P_MF(S, 6); C64P_RD_B(N, 2, NTAP, NKS); C64P_SB;                                                              \
  C64P_MF(S, 7); C64P_SB;                                                                                          \
  C64P_WAIT1(5, fa[S][2]); C64P_MF(S, 8); C64P_RD_B(N, 3, NTAP, NKS); C64P_SB;                                     \
  C64P_MF(S, 9); C64P_SB;                                                                                          \
  C64P_MF(S, 10); C64P_RD_A(N, 1, NTAP, NKS); C64P_SB;                                                             \
  C64P_MF(S, 11); C64P_SB;                                                                                         \
  C64P_WAIT1(6, fa[S][3]); C64P_MF(S, 12); C64P_RD_A(N, 2, NTAP, NKS); C64P_SB;                                    \
  C64P_MF(S, 13); C64P_SB;                                                                                         \
  C64P_MF(S, 14); C64P_RD_A(N, 3, NTAP, NKS); C64P_SB;                                                             \
  C64P_MF(S, 15); C64P_SB
#define C64P_LAST_STEP(S)                                                                                         \
  C64P_WAIT2(6, fb[S][0], fa[S][0]); C64P_SB; C64P_MF(S, 0); C64P_SB;                                              \
  C64P_WAIT1(5, fb[S][1]); C64P_MF(S, 1); C64P_SB;                                                                 \
  C64P_WAIT1(4, fb[S][2]); C64P_MF(S, 2); C64P_SB;                                                                 \
  C64P_WAIT1(3, fb[S][3]); C64P_MF(S, 3); C64P_SB;                                                                 \
  C64P_WAIT1(2, fa[S][1]); C64P_MF(S, 4); C64P_MF(S, 5); C64P_MF(S, 6); C64P_MF(S, 7); C64P_SB;                    \
  C64P_WAIT1(1, fa[S][2]); C64P_MF(S, 8); C64P_MF(S, 9); C64P_MF(S, 10); C64P_MF(S, 11); C64P_SB;                  \
  C64P_WAIT1(0, fa[S][3]); C64P_MF(S, 12); C64P_MF(S, 13); C64P_MF(S, 14); C64P_MF(S, 15); C64P_SB

// WREG form: the weights come from registers -- taps 0-7 from AGPRs (all 256 of them; the MFMA reads its B operand
// there directly), tap 8 from VGPRs -- and a step reads the NEXT step's four A fragments one behind every fourth MFMA; at
// each wait exactly three younger reads are in flight.  The MFMAs are inline asm (the compiler's own selection copies
// AGPR-resident operands to VGPRs first): the accumulators live in VGPRs, the first product of a tile writes them with
// srcC = 0, an accumulator is touched again 16 MFMAs later (no back-to-back dependency), and the epilogue's first VALU
// read of them sits behind an explicit 32-wait-state gap (XDL write -> VALU read needs 11 for an 8-pass MFMA).
#if defined(__HIP_DEVICE_COMPILE__)
#define C64R_MFMA(ACC, W, A, CLS)                                                                  \
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(ACC) : CLS(W), "v"(A))
#define C64R_MFMA0(ACC, W, A, CLS)                                                                 \
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=&v"(ACC) : CLS(W), "v"(A))
#else
#define C64R_MFMA(ACC, W, A, CLS) (void)0
#define C64R_MFMA0(ACC, W, A, CLS) (void)0
#endif
#define C64R_MF(S, M, TAP, KS)                                                                                          \
  do {                                                                                                                  \
    if ((TAP) == 0 && (KS) == 0) C64R_MFMA0(acc[(M) >> 2][(M) & 3], wa[0][0][(M) & 3], fa[S][(M) >> 2], "a");           \
    else if ((TAP) < 8) C64R_MFMA(acc[(M) >> 2][(M) & 3], wa[(TAP) < 8 ? (TAP) : 0][KS][(M) & 3], fa[S][(M) >> 2], "a"); \
    else C64R_MFMA(acc[(M) >> 2][(M) & 3], wv[KS][(M) & 3], fa[S][(M) >> 2], "v");                                      \
  } while (0)
#define C64R_QUAD(S, I, TAP, KS, RD)                                                                                   \
  C64P_WAIT1(3, fa[S][I]); C64R_MF(S, 4 * (I), TAP, KS); RD; C64R_MF(S, 4 * (I) + 1, TAP, KS);                          \
  C64R_MF(S, 4 * (I) + 2, TAP, KS); C64R_MF(S, 4 * (I) + 3, TAP, KS)
#define C64R_STEP(S, N, TAP, KS, NTAP, NKS)                                                                            \
  C64R_QUAD(S, 0, TAP, KS, C64R_RD_A(N, 0, NTAP, NKS)); C64R_QUAD(S, 1, TAP, KS, C64R_RD_A(N, 1, NTAP, NKS));           \
  C64R_QUAD(S, 2, TAP, KS, C64R_RD_A(N, 2, NTAP, NKS)); C64R_QUAD(S, 3, TAP, KS, C64R_RD_A(N, 3, NTAP, NKS))
#define C64R_TAIL(S, I, W, TAP, KS)                                                                                    \
  C64P_WAIT1(W, fa[S][I]); C64R_MF(S, 4 * (I), TAP, KS); C64R_MF(S, 4 * (I) + 1, TAP, KS);                              \
  C64R_MF(S, 4 * (I) + 2, TAP, KS); C64R_MF(S, 4 * (I) + 3, TAP, KS)
#define C64R_LAST_STEP(S, TAP, KS)                                                                                     \
  C64R_TAIL(S, 0, 3, TAP, KS); C64R_TAIL(S, 1, 2, TAP, KS); C64R_TAIL(S, 2, 1, TAP, KS); C64R_TAIL(S, 3, 0, TAP, KS)
// A fragment i of tap (r, s3): patch row wave*2 + (i>>1) + r (4 distinct rows), column (i&1)*16 + frow + s3 -- the second
// 16-pixel half is the first + 2048 bytes exactly (the swizzle key (q>>1)&7 repeats every 16 pixels): 12 address
// registers + an immediate instead of a 36-entry table.
#if defined(__HIP_DEVICE_COMPILE__)
#define C64R_RD_A(S, I, TAP, KS)                                                                                       \
  do {                                                                                                                  \
    const unsigned ad_ = (pbase + aoff12[((I) >> 1) + (TAP) / 3][(TAP) % 3]) ^ ((KS) ? 64u : 0u);                       \
    if ((I) & 1) asm volatile("ds_read_b128 %0, %1 offset:2048" : "=v"(fa[S][I]) : "v"(ad_));                           \
    else asm volatile("ds_read_b128 %0, %1" : "=v"(fa[S][I]) : "v"(ad_));                                               \
  } while (0)
#else
#define C64R_RD_A(S, I, TAP, KS) (void)0
#endif

  f32x4 bias4[4];
  if constexpr (!WREG) c64_load_bias(p, fq, bias4);
  const bool res_pre = c64_residual_preloadable(p);
  __syncthreads();  // weights and the first patch have landed (hipcc drains vmcnt(0) in front of the barrier)
  [[maybe_unused]] u32x4 wa[WREG ? 8 : 1][2][4], wv[2][4];  // WREG: this lane's weight fragments, taps 0-7 (AGPRs) and tap 8 (VGPRs)
  unsigned aoff12[4][3];
  if constexpr (WREG) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const unsigned ad = (ldsW + tap * 8192 + boff[j]) ^ (ks ? 64u : 0u);
          if (tap < 8) asm volatile("ds_read_b128 %0, %1" : "=a"(wa[tap < 8 ? tap : 0][ks][j]) : "v"(ad));
          else asm volatile("ds_read_b128 %0, %1" : "=v"(wv[ks][j]) : "v"(ad));
        }
#pragma unroll
    for (int tap = 0; tap < 8; ++tap)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int j = 0; j < 4; ++j) asm volatile("s_waitcnt lgkmcnt(0)" : "+a"(wa[tap][ks][j]));
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int j = 0; j < 4; ++j) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(wv[ks][j]));
#endif
#pragma unroll
    for (int rr = 0; rr < 4; ++rr)
#pragma unroll
      for (int s3 = 0; s3 < 3; ++s3) {
        const int q = (wave * 2 + rr) * C64_PW + frow + s3;
        aoff12[rr][s3] = (unsigned)(q * 128 + ((fq ^ ((q >> 1) & 7)) << 4));
      }
  }
  if constexpr (WREG) {
    // ---- lean per-tile staging and epilogue of the WREG form (one wavefront per SIMD: every VALU instruction outside the
    // MFMA loop is exposed, so the per-tile address work is hoisted: per-lane offsets relative to the tile origin are
    // computed once, the tile origin is scalar, loads and stores go through buffer resources whose range check drops the
    // lanes outside the image) ----
    constexpr int INVALID = (int)0x80000000;  // buffer offset past any < 2 GiB resource (also after a +16 immediate)
    // tile index -> (image, tile origin) with multiply-high by precomputed reciprocals (exact while n_tiles * tiles per
    // image < 2^32, checked by the launcher): scalar instructions, no division sequence in the tile loop
    const unsigned m_tpi = (unsigned)(((1ull << 32) + (unsigned)tpi - 1) / (unsigned)tpi);
    const unsigned m_tx = (unsigned)(((1ull << 32) + (unsigned)tiles_x - 1) / (unsigned)tiles_x);
    struct TileAt { int img, y0, x0; };
    auto tile_at = [&](int tl) {
      const int img = (int)__umulhi((unsigned)tl, m_tpi);
      const int t = tl - img * tpi;
      const int ty = (int)__umulhi((unsigned)t, m_tx);
      return TileAt{img, ty * C64_TH, (t - ty * tiles_x) * C64_TW};
    };
    // One 16-byte request of a halo patch: request k of this thread is patch pixel q = 32 k + tid/8 (row q / 34 by a
    // multiply-shift, exact below 384), chunk tid%8, swizzled as the fragment reads expect; outside the image (or with
    // `on` false) the offset is out of range and the DMA writes zeros.  About ten VALU instructions: issued one at a time
    // between the MFMA steps they hide in the matrix pipe's shadow, and the request never meets a full address FIFO
    // (44 requests of a workgroup in one burst cost each wavefront ~1.2 k cycles of issue stalls: profiles/r03_c64_wreg.md).
    [[maybe_unused]] const int rq = tid >> 3, rc = tid & 7;
    auto stage_one = [&](const TileAt at, unsigned lds_base, int k, bool on) {
#if defined(__HIP_DEVICE_COMPILE__)
      const int q = k * 32 + rq;
      const int py = (q * 241) >> 13, px = q - py * C64_PW;
      const int y = at.y0 - 1 + py, x = at.x0 - 1 + px;
      const bool ok = on && q < C64_NPIX && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
      const int off = ((at.img * p.H + y) * p.W + x) * 128 + ((rc ^ ((q >> 1) & 7)) << 4);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcA, (lds_void*)(size_t)(lds_base + wave_u * 1024 + k * 4096), 16,
                                               ok ? off : -1, 0, 0, 0);
#endif
    };
    auto stage_patch_r = [&](const TileAt at, unsigned lds_base) {
#pragma unroll
      for (int k = 0; k < C64P_NLOAD; ++k) stage_one(at, lds_base, k, true);
    };
    const int n_img = p.M / (p.H * p.W);
    const int Hp = p.H >> 1, Wp = p.W >> 1;
    const int ldc2 = (int)p.ldc * 2, ldr2 = (int)p.ldr * 2;
    [[maybe_unused]] const __amdgpu_buffer_rsrc_t rsrcC = __builtin_amdgcn_make_buffer_rsrc(
        p.C, 0, (p.pool ? n_img * Hp * Wp : p.M) * ldc2, 0x00020000);
    [[maybe_unused]] const __amdgpu_buffer_rsrc_t rsrcR = __builtin_amdgcn_make_buffer_rsrc(
        (void*)p.residual, 0, p.residual ? p.M * ldr2 : 0, 0x00020000);
    // this lane's pixel group i: tile row wave*2 + (i>>1), tile column (i&1)*16 + frow; its 16 channels 16*fq ..
    int ooff[4], roff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = wave * 2 + (i >> 1), col = (i & 1) * 16 + frow;
      roff[i] = (row * p.W + col) * ldr2 + 32 * fq;
      ooff[i] = p.pool ? (wave * Wp + (((i & 1) * 16 + frow) >> 1)) * ldc2 + 32 * fq : (row * p.W + col) * ldc2 + 32 * fq;
    }
    const float lo = p.relu ? 0.f : -__builtin_inff();
    // Three patch buffers (the third is the weights' LDS image, dead once the fragments sit in registers): the 11 requests
    // of patch t+2 are issued one by one inside tile t's MFMA loop, into the buffer tile t-1 was read from, so a request
    // has more than a whole tile period to land.
    // Vector-memory order per tile:  ... stores(t-1) | residual(t) | patch(t+2) ;  the wait in front of tile t's epilogue
    // leaves exactly the 11 patch(t+2) requests in flight (vmcnt counts in order; past the last tile they are issued
    // out of range): it covers patch(t+1), the stores of tile t-1 and residual(t) without waiting for the youngest.
    __builtin_amdgcn_s_barrier();  // every wavefront has its weight fragments: the weights' LDS image is free
    // the bias moves from 16 registers to the free tail of that image (the epilogue reads its 16 values back per tile)
    float* sBias = (float*)(sW + C64_PATCH_BYTES);
    if (tid < 64) sBias[tid] = p.bias ? p.bias[tid] : 0.f;
    __syncthreads();
    const int stride = (int)gridDim.x;
    unsigned pb0 = ldsP, pb1 = ldsP + C64_PATCH_BYTES, pb2 = ldsW;  // LDS byte addresses of patch t, t+1, t+2
    u32x4 rres[4][2];
    auto load_residual = [&](const TileAt at) {
      const int img = at.img, y0 = at.y0, x0 = at.x0;
      const int rbase = ((img * p.H + y0) * p.W + x0) * ldr2;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const bool in = wave * 2 + (i >> 1) < p.H - y0 && (i & 1) * 16 + frow < p.W - x0;
        const int ro = in ? rbase + roff[i] : INVALID;  // (a lane outside the image reads zeros)
        rres[i][0] = __builtin_amdgcn_raw_buffer_load_b128(rsrcR, ro, 0, 0);
        rres[i][1] = __builtin_amdgcn_raw_buffer_load_b128(rsrcR, ro + 16, 0, 0);
      }
    };
    TileAt at0 = tile_at(tile), at1 = tile_at(tile + stride), at2 = tile_at(tile + 2 * stride);  // tiles t, t+1, t+2
    if (tile + stride < n_tiles) stage_patch_r(at1, pb1);
    if (p.residual) load_residual(at0);
    [[maybe_unused]] u32x4 fa[2][4];
    [[maybe_unused]] unsigned pbase = pb0;
    C64R_RD_A(0, 0, 0, 0); C64R_RD_A(0, 1, 0, 0); C64R_RD_A(0, 2, 0, 0); C64R_RD_A(0, 3, 0, 0);
#if (C64P_ABL & 64)
    long long tph[5] = {0, 0, 0, 0, 0}, tmark = (long long)__builtin_amdgcn_s_memtime();
#define C64R_MARK(K) do { const long long n_ = (long long)__builtin_amdgcn_s_memtime(); tph[K] += n_ - tmark; tmark = n_; } while (0)
#else
#define C64R_MARK(K) (void)0
#endif
    for (;;) {
      C64R_MARK(4);
      const int next = tile + stride;
      const int img = at0.img, y0 = at0.y0, x0 = at0.x0;
      const int rows_left = p.H - y0, cols_left = p.W - x0;
      f32x4 acc[4][4];
      __builtin_amdgcn_sched_barrier(0);
      if (!((C64P_ABL & 1) && p.alpha != 12345.f)) {
      const bool on2 = tile + 2 * stride < n_tiles && !((C64P_ABL & 4) && p.alpha != 12345.f);
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        C64R_STEP(0, 1, tap, 0, tap, 1);
        if (2 * tap < C64P_NLOAD) stage_one(at2, pb2, 2 * tap, on2);
        if (tap < 8) {
          C64R_STEP(1, 0, tap, 1, tap + 1 < 9 ? tap + 1 : 8, 0);
        } else {
          C64R_LAST_STEP(1, 8, 1);
        }
        if (2 * tap + 1 < C64P_NLOAD) stage_one(at2, pb2, 2 * tap + 1, on2);
      }
      }
      C64R_MARK(0);
#if defined(__HIP_DEVICE_COMPILE__)
      // the last MFMAs' results reach the VGPRs before the epilogue reads them (XDL write -> VALU read: 11 wait states for
      // an 8-pass MFMA); the next patch (this wavefront's share) and this tile's residual rows have landed
      if (!((C64P_ABL & 32) && p.alpha != 12345.f)) asm volatile("s_nop 15\n\ts_nop 15\n\ts_waitcnt vmcnt(11)" ::: "memory");
#endif
      C64R_MARK(1);
      if (next < n_tiles) {
        __builtin_amdgcn_s_barrier();  // every wavefront is done reading patch t and has seen its share of patch t+1 land
        C64R_MARK(2);
        pbase = pb1;                   // the first fragments of tile t+1 travel under this tile's epilogue
        C64R_RD_A(0, 0, 0, 0); C64R_RD_A(0, 1, 0, 0); C64R_RD_A(0, 2, 0, 0); C64R_RD_A(0, 3, 0, 0);
      }
      // epilogue: bias + residual + ReLU (+ the 2x2 max pool), the same arithmetic as c64_epilogue's bf16 path
      if (!((C64P_ABL & 2) && p.alpha != 12345.f)) {
      f32x4 bias4[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) bias4[j] = *(const f32x4*)(sBias + 16 * fq + 4 * j);
      if (p.pool) {
        const int obase = ((img * Hp + (y0 >> 1)) * Wp + (x0 >> 1)) * ldc2;
        const int py = (y0 >> 1) + wave;
#pragma unroll
        for (int ih = 0; ih < 2; ++ih) {
          float best[16];
#pragma unroll
          for (int e = 0; e < 16; ++e) best[e] = -__builtin_inff();
#pragma unroll
          for (int iv = 0; iv < 2; ++iv) {
            const int i = ih + 2 * iv;
            float v[16];
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
              for (int r = 0; r < 4; ++r) v[4 * j + r] = acc[i][j][r] * p.alpha + bias4[j][r];
            if (p.residual) {
              const bf16x8 r0 = __builtin_bit_cast(bf16x8, rres[i][0]), r1 = __builtin_bit_cast(bf16x8, rres[i][1]);
#pragma unroll
              for (int e = 0; e < 8; ++e) {
                v[e] += (float)r0[e];
                v[8 + e] += (float)r1[e];
              }
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) best[e] = fmaxf(best[e], (float)(bf16_t)fmaxf(v[e], lo));
          }
#pragma unroll
          for (int e = 0; e < 16; ++e) best[e] = fmaxf(best[e], __shfl_xor(best[e], 1));
          const int px = (x0 + ih * 16 + frow) >> 1;
          const bool st = (frow & 1) == 0 && py < Hp && px < Wp;
          bf16x8 o0, o1;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            o0[e] = (bf16_t)best[e];
            o1[e] = (bf16_t)best[8 + e];
          }
          const int oo = st ? obase + ooff[ih] : INVALID;
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o0), rsrcC, oo, 0, 0);
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o1), rsrcC, oo + 16, 0, 0);
        }
      } else {
        const int obase = ((img * p.H + y0) * p.W + x0) * ldc2;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float v[16];
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) v[4 * j + r] = acc[i][j][r] * p.alpha + bias4[j][r];
          if (p.residual) {
            const bf16x8 r0 = __builtin_bit_cast(bf16x8, rres[i][0]), r1 = __builtin_bit_cast(bf16x8, rres[i][1]);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              v[e] += (float)r0[e];
              v[8 + e] += (float)r1[e];
            }
          }
          bf16x8 o0, o1;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            o0[e] = (bf16_t)fmaxf(v[e], lo);
            o1[e] = (bf16_t)fmaxf(v[8 + e], lo);
          }
          const bool in = wave * 2 + (i >> 1) < rows_left && (i & 1) * 16 + frow < cols_left;
          const int oo = in ? obase + ooff[i] : INVALID;
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o0), rsrcC, oo, 0, 0);
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o1), rsrcC, oo + 16, 0, 0);
        }
      }
      }
      C64R_MARK(3);
      if (next >= n_tiles) break;
      if (p.residual) load_residual(at1);
      at0 = at1; at1 = at2; at2 = tile_at(tile + 3 * stride);
      {
        const unsigned tp = pb0;  // the buffer tile t was read from (free since the barrier) takes patch t+3 next
        pb0 = pb1; pb1 = pb2; pb2 = tp;
      }
      tile = next;
    }
#if (C64P_ABL & 64)
    // per-phase cycle sums of wavefront 0: [MFMA loop, vmcnt wait, barrier, first reads + epilogue, residual + staging]
    if (p.partial && lane == 0 && wave == 0)
      for (int k = 0; k < 5; ++k) p.partial[blockIdx.x * 8 + k] = (float)tph[k];
#endif
#undef C64R_MARK
    return;
  }
  for (int cur = 0;; cur ^= 1) {
    const int next = tile + (int)gridDim.x;
    if (next < n_tiles && !((C64P_ABL & 4) && p.alpha != 12345.f)) stage_patch(next, cur ^ 1);  // lands behind this tile's MFMAs and epilogue
    const int img = tile / tpi;
    const int t = tile - img * tpi;
    const int ty = t / tiles_x, tx = t - ty * tiles_x;
    const int y0 = ty * C64_TH, x0 = tx * C64_TW;
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    [[maybe_unused]] const unsigned pbase = ldsP + (unsigned)(cur * C64_PATCH_BYTES);
    bf16x8 rres[4][2];
    if (res_pre) c64_load_residual(p, img, y0, x0, wave, frow, fq, rres);
    u32x4 fa[2][4], fb[2][4];
    __builtin_amdgcn_sched_barrier(0);
    if (!((C64P_ABL & 1) && p.alpha != 12345.f)) {
    C64P_READ_STEP(0, 0, 0);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      C64P_STEP(0, 1, tap, 1);
      if (tap < 8) {
        C64P_STEP(1, 0, tap + 1 < 9 ? tap + 1 : 8, 0);
      } else {
        C64P_LAST_STEP(1);
      }
    }
    }
#if defined(__HIP_DEVICE_COMPILE__)
    // The next patch has had the whole MFMA loop to land.  Waiting here, BEFORE the epilogue, keeps this tile's stores
    // out of the wait (gfx9 counts stores in vmcnt): they retire behind the next tile's MFMAs.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    if (!((C64P_ABL & 2) && p.alpha != 12345.f)) {
      if (res_pre)
        c64_epilogue<true>(p, acc, bias4, img, y0, x0, wave, frow, fq, rres);
      else
        c64_epilogue(p, acc, bias4, img, y0, x0, wave, frow, fq);
    }
    if (next >= n_tiles) break;
    __builtin_amdgcn_s_barrier();  // every wavefront is done reading patch `cur` and has seen its share of the next land
    tile = next;
  }
#undef C64P_READ
#undef C64P_WAIT1
#undef C64P_WAIT2
#undef C64P_READ_STEP
#undef C64P_RD_A
#undef C64P_RD_B
#undef C64P_MF
#undef C64P_SB
#undef C64P_STEP
#undef C64P_LAST_STEP
#undef C64R_MF
#undef C64R_MFMA
#undef C64R_MFMA0
#undef C64R_RD_A
#undef C64R_QUAD
#undef C64R_STEP
#undef C64R_TAIL
#undef C64R_LAST_STEP
}

// Epilogue of the bf16x2 64-channel kernels: bias + (bf16x2) residual + ReLU (+ 2x2 / stride-2 max pool), bf16x2 stores.
template <int XG, int NI>
__device__ __forceinline__ void c64x_epilogue(const GemmArgs& p, const f32x4 (&acc)[NI][4], const f32x4 (&bias4)[4], const int img,
                                              const int y0, const int x0, const int wave, const int frow, const int fq,
                                              char* sE = nullptr) {
  // ---- epilogue: lane (frow, fq) holds, for pixel group i, channels 16 fq + 4 j + r of pixel frow: slots
  // 64 (fq >> 1) + 16 (fq & 1) (hi) and 32 further (lo) of the pixel's 128
  const float lo_clip = p.relu ? 0.f : -__builtin_inff();
  const int slot = 64 * (fq >> 1) + 16 * (fq & 1);
  auto load16 = [&](const bf16_t* q, float (&v)[16]) {  // bf16x2 residual values of this lane's 16 channels
    const bf16x8 h0 = *(const bf16x8*)q, h1 = *(const bf16x8*)(q + 8), l0 = *(const bf16x8*)(q + 32), l1 = *(const bf16x8*)(q + 40);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      v[e] = (float)h0[e] + (float)l0[e];
      v[8 + e] = (float)h1[e] + (float)l1[e];
    }
  };
  auto store16 = [&](bf16_t* q, const float (&v)[16]) {
    bf16x8 h0, h1, l0, l1;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      h0[e] = (bf16_t)v[e];
      h1[e] = (bf16_t)v[8 + e];
      l0[e] = x2_lo(v[e], h0[e]);
      l1[e] = x2_lo(v[8 + e], h1[e]);
    }
    *(bf16x8*)q = h0;
    *(bf16x8*)(q + 8) = h1;
    *(bf16x8*)(q + 32) = l0;
    *(bf16x8*)(q + 40) = l1;
  };
  // Through LDS (sE = 4 KiB private to the wavefront; round 5): a lane's four 16-byte pieces of a pixel lie 256 B from its
  // neighbour lane's, so a direct store instruction is 64 separate 16-byte write requests.  The pieces are parked as
  // [pixel][16 chunks] (chunk ^= pixel: conflict-free both ways) and leave as whole pixels, 1 KiB contiguous per instruction.
  typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
  auto park16 = [&](const int prow, const float (&v)[16]) {
    bf16x8 h0, h1, l0, l1;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      h0[e] = (bf16_t)v[e];
      h1[e] = (bf16_t)v[8 + e];
      l0[e] = x2_lo(v[e], h0[e]);
      l1[e] = x2_lo(v[8 + e], h1[e]);
    }
    const int c0 = 8 * (fq >> 1) + 2 * (fq & 1);
    char* row = sE + prow * 256;
    *(bf16x8*)(row + ((c0 ^ prow) << 4)) = h0;
    *(bf16x8*)(row + (((c0 + 1) ^ prow) << 4)) = h1;
    *(bf16x8*)(row + (((c0 + 4) ^ prow) << 4)) = l0;
    *(bf16x8*)(row + (((c0 + 5) ^ prow) << 4)) = l1;
  };
  auto flush = [&](const int npix, char* gbase, const long long pix_bytes, const int nvalid) {  // gbase: pixel 0 of the group
    const int lp = (fq * 16 + frow) >> 4, c = frow;  // lane = 16 fq + frow -> (pixel within a pass of 4, chunk)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (4 * k >= npix) break;
      const int P = 4 * k + lp;
      const u32x4 d = *(const u32x4*)(sE + P * 256 + ((c ^ P) << 4));
      if (P < nvalid) *(u32x4*)(gbase + P * pix_bytes + c * 16) = d;
    }
  };
  if (p.pool) {  // MaxPool2d(2, 2): the wavefront's two image rows are one pooled row, the horizontal partner is lane frow ^ 1
    const int Hp = p.H >> 1, Wp = p.W >> 1;
#pragma unroll
    for (int ih = 0; ih < XG; ++ih) {
      float best[16];
#pragma unroll
      for (int e = 0; e < 16; ++e) best[e] = -__builtin_inff();
#pragma unroll
      for (int iv = 0; iv < 2; ++iv) {
        const int i = ih + XG * iv;
        const int y = min(y0 + wave * 2 + iv, p.H - 1), x = min(x0 + ih * 16 + frow, p.W - 1);  // clamped: unused if outside
        const long long m = ((long long)img * p.H + y) * p.W + x;
        float v[16];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) v[4 * j + r] = acc[i][j][r] * p.alpha + bias4[j][r];
        if (p.residual) {
          float rv[16];
          load16((const bf16_t*)p.residual + 2 * m * p.ldr + slot, rv);
#pragma unroll
          for (int e = 0; e < 16; ++e) v[e] += rv[e];
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) best[e] = fmaxf(best[e], fmaxf(v[e], lo_clip));
      }
#pragma unroll
      for (int e = 0; e < 16; ++e) best[e] = fmaxf(best[e], __shfl_xor(best[e], 1));
      const int py = (y0 >> 1) + wave, px = (x0 + ih * 16 + frow) >> 1;
      if (sE) {
        if ((frow & 1) == 0) park16(frow >> 1, best);
        const int px0 = (x0 + ih * 16) >> 1;
        if (py < Hp && px0 < Wp)
          flush(8, (char*)((bf16_t*)p.C + 2 * (((long long)img * Hp + py) * Wp + px0) * p.ldc), 4ll * p.ldc, Wp - px0);
        continue;
      }
      if ((frow & 1) == 0 && py < Hp && px < Wp)
        store16((bf16_t*)p.C + 2 * (((long long)img * Hp + py) * Wp + px) * p.ldc + slot, best);
    }
    return;
  }
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int y = y0 + wave * 2 + (i / XG);
    const int x = x0 + (i % XG) * 16 + frow;
    if (sE) {
      const int xg = x0 + (i % XG) * 16;
      if (y >= p.H || xg >= p.W) continue;  // (wave-uniform)
      const long long mc = ((long long)img * p.H + y) * p.W + min(x, p.W - 1);  // clamped: parked, never flushed
      float v[16];
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) v[4 * j + r] = acc[i][j][r] * p.alpha + bias4[j][r];
      if (p.residual) {
        float rv[16];
        load16((const bf16_t*)p.residual + 2 * mc * p.ldr + slot, rv);
#pragma unroll
        for (int e = 0; e < 16; ++e) v[e] += rv[e];
      }
#pragma unroll
      for (int e = 0; e < 16; ++e) v[e] = fmaxf(v[e], lo_clip);
      park16(frow, v);
      flush(16, (char*)((bf16_t*)p.C + 2 * (((long long)img * p.H + y) * p.W + xg) * p.ldc), 4ll * p.ldc, p.W - xg);
      continue;
    }
    if (y >= p.H || x >= p.W) continue;
    const long long m = ((long long)img * p.H + y) * p.W + x;
    float v[16];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) v[4 * j + r] = acc[i][j][r] * p.alpha + bias4[j][r];
    if (p.residual) {
      float rv[16];
      load16((const bf16_t*)p.residual + 2 * m * p.ldr + slot, rv);
#pragma unroll
      for (int e = 0; e < 16; ++e) v[e] += rv[e];
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) v[e] = fmaxf(v[e], lo_clip);
    store16((bf16_t*)p.C + 2 * m * p.ldc + slot, v);
  }
}

// ---------------------------------------------------------------------------------
// The 64 -> 64 channel 3x3 convolution on bf16x2 maps (MODEL.HIP.PRECISION = "parity"; include/wsovod_hip.h).  Same
// decomposition as conv3x3_c64_kernel -- a workgroup owns 8 x 32 output pixels, its 10 x 34 halo patch is DMA-staged once
// and all nine taps read their A fragments from it -- with 256-byte pixel rows (64 values = [hi 0-31 | lo 0-31 | hi 32-63 |
// lo 32-63]) and a 16-KiB weight slice per tap (double buffered).  A tap is two K-steps of 32 values; a K-step reads
// (a_hi, a_lo) for its 4 pixel groups and (b_hi, b_lo) for the 4 channel tiles (16 ds_read_b128) and issues 48 MFMAs
// b_hi*a_hi + b_lo*a_hi + b_hi*a_lo: 3 MFMAs per fragment read against 2 in the bf16 kernel.
// LDS rows of 256 B: the 16-byte chunk index is XORed with the low 4 bits of the row (pixel / weight row) index, on the
// DMA source address and on the read address, so the 16 consecutive rows of a fragment read fall into 16 different slots.
// The epilogue adds bias and the (bf16x2) residual, applies ReLU, optionally the 2x2 / stride-2 max pool (pool = 2),
// and stores bf16x2.
// ---------------------------------------------------------------------------------
// Tile width TW = 16 (default): 8 x 16 output pixels per workgroup, 80 KiB of LDS -> TWO workgroups per CU: one's patch
// staging and per-tap barriers hide behind the other's MFMAs (two wavefronts per SIMD).  TW = 32: the bf16 kernel's
// 8 x 32 tile, 120 KiB, one workgroup per CU (3 MFMAs per fragment read instead of 2, but nothing to overlap the
// prologue with): measured slower (WSOVOD_C64X_TW=32 selects it for A/B runs).
template <int TW>
struct C64X {
  static constexpr int XG = TW / 16, NI = 2 * XG;       // 16-pixel groups per image row / per wavefront (2 rows)
  static constexpr int PW = TW + 2, NPIX = PW * C64_PH;
  static constexpr int NPIX_PAD = (NPIX + 15) / 16 * 16;  // whole 16-pixel DMA passes (4 wavefronts x 4 pixels)
  static constexpr int PATCH_BYTES = NPIX_PAD * 256;
  static constexpr int W_BYTES = 64 * 256;
  static constexpr int LDS_BYTES = PATCH_BYTES + 2 * W_BYTES;
};

template <int TW>
__global__ __launch_bounds__(256) void conv3x3_c64_x3_kernel(const GemmArgs p, int tiles_x, int tiles_y) {
  using G = C64X<TW>;
  constexpr int XG = G::XG, NI = G::NI, PW = G::PW;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sP = smem;                    // halo patch [pixel][256 B]
  char* sW = smem + G::PATCH_BYTES;   // 2 x [64 cout][256 B] weight slices
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tpi = tiles_x * tiles_y;
  const int img = blockIdx.x / tpi;
  const int t = blockIdx.x - img * tpi;
  const int ty = t / tiles_x, tx = t - ty * tiles_x;
  const int y0 = ty * C64_TH, x0 = tx * TW;
  typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
  [[maybe_unused]] const __amdgpu_buffer_rsrc_t rsrcA =
      __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, (int)p.a_bytes, 0x00020000);
  [[maybe_unused]] const __amdgpu_buffer_rsrc_t rsrcB =
      __builtin_amdgcn_make_buffer_rsrc((void*)p.B, 0, (int)(64 * p.ldb * 2), 0x00020000);
  [[maybe_unused]] const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  typedef __attribute__((address_space(3))) void lds_void [[maybe_unused]];

  // a wavefront DMA instruction lands 4 rows x 256 B: lane -> (row lane >> 4, slot lane & 15)
  auto stage_weights = [&](int tap, int buf) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = i * 16 + wave_u * 4 + (lane >> 4);   // LDS row = MFMA tile j = row >> 4, tile row f = row & 15
      const int chunk = (lane & 15) ^ (row & 15);          // swizzle on the source
      const int cout = 16 * ((row & 15) >> 2) + 4 * (row >> 4) + (row & 3);  // a lane ends up with 16 consecutive channels
      const int off = (int)((cout * p.ldb + tap * 128 + chunk * 8) * 2);    // (p.ldb, in bf16 slots = 2 x 9 x 64)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcB, (lds_void*)(sW + buf * G::W_BYTES + (i * 16 + wave_u * 4) * 256), 16,
                                               off, 0, 0, 0);
    }
#endif
  };
#if defined(__HIP_DEVICE_COMPILE__)
  for (int base = 0; base < G::NPIX_PAD; base += 16) {
    const int q = base + wave_u * 4 + (lane >> 4);
    const int py = q / PW, px = q - py * PW;
    const int y = y0 - 1 + py, x = x0 - 1 + px;
    const bool ok = q < G::NPIX && y >= 0 && y < p.H && x >= 0 && x < p.W;
    const int chunk = (lane & 15) ^ (q & 15);
    const int off = (((img * p.H + y) * p.W + x) * 128 + chunk * 8) * 2;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcA, (lds_void*)(sP + (base + wave_u * 4) * 256), 16, ok ? off : -1, 0, 0, 0);
  }
#endif
  stage_weights(0, 0);
  __syncthreads();

  const int frow = lane & 15, fq = lane >> 4;
  f32x4 bias4[4];
  c64_load_bias(p, fq, bias4);
  f32x4 acc[NI][4];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
  for (int tap = 0; tap < 9; ++tap) {
    const int cur = tap & 1;
    if (tap + 1 < 9) stage_weights(tap + 1, cur ^ 1);
    const int r = tap / 3, s3 = tap - r * 3;
    const char* cW = sW + cur * G::W_BYTES;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {  // 32-value group of the 64 input channels: chunks 8 kk + (0-3 hi | 4-7 lo)
      u32x4 ah[NI], al[NI], bh[4], bl[4];
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const int q = (wave * 2 + (i / XG) + r) * PW + (i % XG) * 16 + frow + s3;
        const char* row = sP + q * 256;
        ah[i] = *(const u32x4*)(row + (((8 * kk + fq) ^ (q & 15)) << 4));
        al[i] = *(const u32x4*)(row + (((8 * kk + 4 + fq) ^ (q & 15)) << 4));
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int row = j * 16 + frow;
        bh[j] = *(const u32x4*)(cW + row * 256 + (((8 * kk + fq) ^ (row & 15)) << 4));
        bl[j] = *(const u32x4*)(cW + row * 256 + (((8 * kk + 4 + fq) ^ (row & 15)) << 4));
      }
#pragma unroll
      for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, bh[j]),
                                                              __builtin_bit_cast(bf16x8, ah[i]), acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, bl[j]),
                                                              __builtin_bit_cast(bf16x8, ah[i]), acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, bh[j]),
                                                              __builtin_bit_cast(bf16x8, al[i]), acc[i][j], 0, 0, 0);
    }
    __syncthreads();
  }
  c64x_epilogue<XG, NI>(p, acc, bias4, img, y0, x0, wave, frow, fq);
}

// The same convolution with the 64 input channels taken in two HALVES of 32 (the form the dispatcher uses): an 8 x 32-pixel
// tile whose 10 x 34 halo patch is staged as 128-byte rows ([hi | lo] of 32 channels: 44 KiB) with an 8-KiB weight slice
// per (half, tap), double buffered -- 60 KiB per workgroup, so TWO workgroups fit a CU although a wavefront now owns 4
// pixel groups x 4 channel tiles: a K-step is 16 fragment reads for 48 MFMAs (the 8 x 16 tile above: 12 for 24, which
// keeps the LDS pipe as busy as the matrix pipe), and the weight slices travel from L2 once per 256 pixels instead of
// once per 128.  The patch of the second half is staged after the ninth tap of the first (the other workgroup's MFMAs
// cover it).  Rows are swizzled like the GEMM image (chunk ^= (row >> 1) & 7, on the DMA source and on the read).
struct C64XH {
  static constexpr int TW = 32, XG = 2, NI = 4;
  static constexpr int PW = TW + 2, NPIX = PW * C64_PH;
  static constexpr int NPIX_PAD = (NPIX + 31) / 32 * 32;  // whole 32-pixel DMA passes (4 wavefronts x 8 rows of 128 B)
  static constexpr int PATCH_BYTES = NPIX_PAD * 128;
  static constexpr int W_BYTES = 64 * 128;
  static constexpr int LDS_BYTES = PATCH_BYTES + 2 * W_BYTES;
};

template <bool LEPI>
__global__ __launch_bounds__(256, 2) void conv3x3_c64_x3h_kernel(const GemmArgs p, int tiles_x, int tiles_y) {
  using G = C64XH;
  constexpr int XG = G::XG, NI = G::NI, PW = G::PW;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sP = smem;                    // halo patch of one channel half [pixel][128 B]
  char* sW = smem + G::PATCH_BYTES;   // 2 x [64 cout][128 B] weight slices
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tpi = tiles_x * tiles_y;
  const int img = blockIdx.x / tpi;
  const int t = blockIdx.x - img * tpi;
  const int ty = t / tiles_x, tx = t - ty * tiles_x;
  const int y0 = ty * C64_TH, x0 = tx * G::TW;
  typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
  [[maybe_unused]] const __amdgpu_buffer_rsrc_t rsrcA =
      __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, (int)p.a_bytes, 0x00020000);
  [[maybe_unused]] const __amdgpu_buffer_rsrc_t rsrcB =
      __builtin_amdgcn_make_buffer_rsrc((void*)p.B, 0, (int)(64 * p.ldb * 2), 0x00020000);
  [[maybe_unused]] const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  typedef __attribute__((address_space(3))) void lds_void [[maybe_unused]];

  // a wavefront DMA instruction lands 8 rows x 128 B: lane -> (row lane >> 3, slot lane & 7)
  auto stage_weights = [&](int step, int buf) {  // step = 9 * half + tap
#if defined(__HIP_DEVICE_COMPILE__)
    const int half = step >= 9 ? 1 : 0, tap = step - 9 * half;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = i * 32 + wave_u * 8 + (lane >> 3);   // LDS row = MFMA tile j = row >> 4, tile row f = row & 15
      const int chunk = (lane & 7) ^ ((row >> 1) & 7);     // swizzle on the source
      const int cout = 16 * ((row & 15) >> 2) + 4 * (row >> 4) + (row & 3);  // a lane ends up with 16 consecutive channels
      const int off = (int)((cout * p.ldb + tap * 128 + half * 64 + chunk * 8) * 2);  // (p.ldb, in bf16 slots = 2 x 9 x 64)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcB, (lds_void*)(sW + buf * G::W_BYTES + (i * 32 + wave_u * 8) * 128), 16,
                                               off, 0, 0, 0);
    }
#endif
  };
  auto stage_patch = [&](int half) {
#if defined(__HIP_DEVICE_COMPILE__)
    for (int base = 0; base < G::NPIX_PAD; base += 32) {
      const int q = base + wave_u * 8 + (lane >> 3);
      const int py = q / PW, px = q - py * PW;
      const int y = y0 - 1 + py, x = x0 - 1 + px;
      const bool ok = q < G::NPIX && y >= 0 && y < p.H && x >= 0 && x < p.W;
      const int chunk = (lane & 7) ^ ((q >> 1) & 7);
      const int off = (((img * p.H + y) * p.W + x) * 128 + half * 64 + chunk * 8) * 2;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcA, (lds_void*)(sP + (base + wave_u * 8) * 128), 16, ok ? off : -1, 0, 0, 0);
    }
#endif
  };
#if C64XH_PHASES  // instrumented build (tools/c64x_phases.py): s_memtime ticks per phase, wavefront 0 of every workgroup
  long long tph[6] = {0, 0, 0, 0, 0, 0}, tmark = (long long)__builtin_amdgcn_s_memtime();
#define C64XH_MARK(K) do { const long long n_ = (long long)__builtin_amdgcn_s_memtime(); tph[K] += n_ - tmark; tmark = n_; } while (0)
#else
#define C64XH_MARK(K) (void)0
#endif
  stage_patch(0);
  stage_weights(0, 0);
  stage_weights(1, 1);

  const int frow = lane & 15, fq = lane >> 4;
  f32x4 acc[NI][4];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  int qb[NI], boff[4];  // patch pixel of this lane's fragment row with the filter at its top-left tap; weight-slice offsets
#pragma unroll
  for (int i = 0; i < NI; ++i) qb[i] = (wave * 2 + (i / XG)) * PW + (i % XG) * 16 + frow;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int row = j * 16 + frow;
    boff[j] = row * 128 + ((fq ^ ((row >> 1) & 7)) << 4);
  }
  // fragments of K-step `step` (= 9 * half + tap): (a_hi, a_lo) from the patch, (b_hi, b_lo) from weight buffer step & 1;
  // the lo chunk of a row is the hi chunk's slot ^ 4
  auto read_a = [&](int step, u32x4 (&ah)[NI], u32x4 (&al)[NI]) {
    const int tap = step >= 9 ? step - 9 : step;
    const int r = tap / 3;
    const int d = r * PW + (tap - r * 3);
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int q = qb[i] + d;
      const int off = q * 128 + ((fq ^ ((q >> 1) & 7)) << 4);
      ah[i] = *(const u32x4*)(sP + off);
      al[i] = *(const u32x4*)(sP + (off ^ 64));
    }
  };
  auto read_b = [&](int step, u32x4 (&bh)[4], u32x4 (&bl)[4]) {
    const char* cW = sW + (step & 1) * G::W_BYTES;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      bh[j] = *(const u32x4*)(cW + boff[j]);
      bl[j] = *(const u32x4*)(cW + (boff[j] ^ 64));
    }
  };
  auto mfma = [&](const u32x4 (&ah)[NI], const u32x4 (&al)[NI], const u32x4 (&bh)[4], const u32x4 (&bl)[4]) {
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, bh[j]),
                                                            __builtin_bit_cast(bf16x8, ah[i]), acc[i][j], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, bl[j]),
                                                            __builtin_bit_cast(bf16x8, ah[i]), acc[i][j], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, bh[j]),
                                                            __builtin_bit_cast(bf16x8, al[i]), acc[i][j], 0, 0, 0);
  };
  // Software pipeline over the 18 K-steps, two per trip (fragment sets 0 / 1): a step first requests what comes NEXT --
  // the weight slice two steps ahead by DMA into the buffer whose fragments are already in registers, the next step's
  // fragments from LDS into the other register set -- then issues its own 48 MFMAs, then meets the barrier (which retires
  // both).  hipcc moves a __syncthreads() up across MFMAs (they touch registers only), which would put the DMA wait in
  // FRONT of the products it is meant to hide behind: the scheduling barriers pin the order.
  // Inside a step the 16 fragment reads go out one per three MFMAs: a wavefront issues in order, and sixteen reads in a
  // row (16 KiB per wavefront, eight wavefronts on the CU's one LDS pipe) hold its MFMAs back for the whole burst.
#define C64XH_INTERLEAVE(N)                                   \
  do {                                                        \
    _Pragma("unroll") for (int g_ = 0; g_ < (N); ++g_) {      \
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);      \
      __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);      \
    }                                                         \
  } while (0)
  u32x4 ah0[NI], al0[NI], bh0[4], bl0[4], ah1[NI], al1[NI], bh1[4], bl1[4];
  __syncthreads();
  read_a(0, ah0, al0);
  read_b(0, bh0, bl0);
  __syncthreads();  // every wavefront holds the fragments of step 0: weight buffer 0 is free for step 2
  C64XH_MARK(0);
  // one trip = two K-steps; MODE 0: a regular trip, 1: steps 8 / 9 (the patch of the second half is staged under step 8's
  // MFMAs and step 9's patch fragments are read behind the barrier), 2: steps 16 / 17 (nothing left to request).  The trips
  // are branch-free inside (hipcc's group scheduling works per basic block).
  auto trip = [&](const int step, auto mode) {
    constexpr int MODE = decltype(mode)::value;
    if constexpr (MODE != 2) stage_weights(step + 2, 0);
    if constexpr (MODE == 1) stage_patch(1);  // the first half's patch is dead: step 8's fragments are in registers
    read_b(step + 1, bh1, bl1);
    if constexpr (MODE != 1) read_a(step + 1, ah1, al1);
    mfma(ah0, al0, bh0, bl0);
    if constexpr (MODE != 1) C64XH_INTERLEAVE(16);
    else C64XH_INTERLEAVE(8);
    __builtin_amdgcn_sched_barrier(0);
    C64XH_MARK(2);
    __syncthreads();
    __builtin_amdgcn_sched_barrier(0);
    C64XH_MARK(3);
    if constexpr (MODE == 1) {  // the second half's patch landed at the barrier
      read_a(9, ah1, al1);
      __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (MODE != 2) {
      stage_weights(step + 3, 1);
      read_b(step + 2, bh0, bl0);
      read_a(step + 2, ah0, al0);
    }
    mfma(ah1, al1, bh1, bl1);
    if constexpr (MODE != 2) C64XH_INTERLEAVE(16);
    __builtin_amdgcn_sched_barrier(0);
    C64XH_MARK(2);
    __syncthreads();
    __builtin_amdgcn_sched_barrier(0);
    C64XH_MARK(3);
  };
#pragma unroll 1
  for (int step = 0; step < 8; step += 2) trip(step, std::integral_constant<int, 0>{});
  trip(8, std::integral_constant<int, 1>{});
#pragma unroll 1
  for (int step = 10; step < 16; step += 2) trip(step, std::integral_constant<int, 0>{});
  trip(16, std::integral_constant<int, 2>{});
  f32x4 bias4[4];  // (loaded here: 16 registers the loop's two fragment sets leave no room for)
  c64_load_bias(p, fq, bias4);
  // (the patch is dead: every fragment of step 17 was read in front of trip 16's first barrier)
  c64x_epilogue<XG, NI>(p, acc, bias4, img, y0, x0, wave, frow, fq, LEPI ? sP + wave_u * 4096 : nullptr);
#if C64XH_PHASES
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  C64XH_MARK(5);
  if (p.partial && tid == 0) {
    float* dst = p.partial + (blockIdx.x & 511) * 8;
    for (int k = 0; k < 6; ++k) atomicAdd(dst + k, (float)tph[k]);
    atomicAdd(dst + 7, 1.f);
  }
#endif
#undef C64XH_MARK
#undef C64XH_INTERLEAVE
}

template <typename T, int BM, int BN, bool CONV, int WM = 2, int WN = 2, bool DMA = false, int STAGES = 2, bool X3 = false>
int launch(const GemmArgs& a, hipStream_t s, const char* slot_name, double flops, double bytes) {
  static int slot = wsovod::prof_slot(slot_name);
  static bool attr_set = false;
  constexpr int lds_bytes = STAGES * (BM + BN) * 128;
  auto kfn = gemm_nt_kernel<T, BM, BN, CONV, WM, WN, DMA, STAGES, X3>;
  if (!attr_set) {
    WS_CHECK_HIP(hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes), "wsovod_gemm_nt: LDS opt-in");
    attr_set = true;
  }
  GemmArgs args = a;
  args.tiles_m = ceil_div(a.M - a.m_base, BM);
  args.tiles_n = ceil_div(a.N, BN);
  {
    const int run = std::max(1, args.tiles_m * args.tiles_n / 8);  // tiles per XCD
    int g = 1;
    while ((g + 1) * (g + 1) <= run) ++g;
    args.group_m = std::max(1, std::min(g, args.tiles_m));
  }
  wsovod::ProfScope prof(slot, s, flops, bytes);
  hipLaunchKernelGGL(kfn, dim3(args.tiles_m * args.tiles_n), dim3(64 * WM * WN), lds_bytes, s, args);
  WS_CHECK_LAUNCH(slot_name);
  return WSOVOD_OK;
}

template <typename T, bool CONV>
int dispatch_tile(const GemmArgs& a, int tile, hipStream_t s, double flops, double bytes) {
  constexpr bool bf = sizeof(T) == 2;
  switch (tile) {
    case 8256256:  // 8 wavefronts in two staggered groups, four phases per K-step (gemm8.hip); bf16 only
      if constexpr (bf) return launch_gemm256_8ph(a, CONV, s, flops, bytes);
      wsovod::set_error("wsovod_gemm_nt: tile 8256256 is bf16 only");
      return WSOVOD_ERR_UNSUPPORTED;
    case 2256256:  // the same tile with two phases per K-step
      if constexpr (bf) return launch_gemm256_8ph(a, CONV, s, flops, bytes, false, false, true);
      wsovod::set_error("wsovod_gemm_nt: tile 2256256 is bf16 only");
      return WSOVOD_ERR_UNSUPPORTED;
    case 256256:  // 16 wavefronts (4x4), 128 KiB LDS, one workgroup per CU: 128 FLOP per staged byte
      return launch<T, 256, 256, CONV, 4, 4, true>(a, s, CONV ? (bf ? "conv_igemm_bf16_256x256" : "conv_igemm_f32_256x256")
                                                       : (bf ? "gemm_nt_bf16_256x256" : "gemm_nt_f32_256x256"),
                                             flops, bytes);
    case 3256128:  // 256x128, 3-stage deep pipeline (144 KiB LDS)
      return launch<T, 256, 128, CONV, 4, 2, true, 3>(a, s, CONV ? (bf ? "conv_igemm_bf16_256x128_s3" : "conv_igemm_f32_256x128_s3")
                                                                : (bf ? "gemm_nt_bf16_256x128_s3" : "gemm_nt_f32_256x128_s3"),
                                                      flops, bytes);
    case 4128128:  // 128x128, 4-stage deep pipeline (128 KiB LDS)
      return launch<T, 128, 128, CONV, 2, 2, true, 4>(a, s, CONV ? (bf ? "conv_igemm_bf16_128x128_s4" : "conv_igemm_f32_128x128_s4")
                                                                : (bf ? "gemm_nt_bf16_128x128_s4" : "gemm_nt_f32_128x128_s4"),
                                                      flops, bytes);
    case 3128128:  // 128x128, 3-stage (96 KiB LDS)
      return launch<T, 128, 128, CONV, 2, 2, true, 3>(a, s, CONV ? (bf ? "conv_igemm_bf16_128x128_s3" : "conv_igemm_f32_128x128_s3")
                                                                : (bf ? "gemm_nt_bf16_128x128_s3" : "gemm_nt_f32_128x128_s3"),
                                                      flops, bytes);
    case 256128:
      return launch<T, 256, 128, CONV, 4, 2, true>(a, s, CONV ? (bf ? "conv_igemm_bf16_256x128" : "conv_igemm_f32_256x128")
                                                       : (bf ? "gemm_nt_bf16_256x128" : "gemm_nt_f32_256x128"),
                                             flops, bytes);
    case 1256064:  // few output columns (stem / res2 convs, Cout = 64): tall LDS-direct tile
      return launch<T, 256, 64, CONV, 4, 1, true>(a, s, CONV ? (bf ? "conv_igemm_bf16_256x64_dma" : "conv_igemm_f32_256x64_dma")
                                                            : (bf ? "gemm_nt_bf16_256x64_dma" : "gemm_nt_f32_256x64_dma"),
                                                  flops, bytes);
    case 1128064:
      return launch<T, 128, 64, CONV, 2, 2, true>(a, s, CONV ? (bf ? "conv_igemm_bf16_128x64_dma" : "conv_igemm_f32_128x64_dma")
                                                            : (bf ? "gemm_nt_bf16_128x64_dma" : "gemm_nt_f32_128x64_dma"),
                                                  flops, bytes);
    case 1128128:  // 128x128 with LDS-direct staging (A/B comparison against the register-staged form)
      return launch<T, 128, 128, CONV, 2, 2, true>(a, s, CONV ? (bf ? "conv_igemm_bf16_128x128_dma" : "conv_igemm_f32_128x128_dma")
                                                             : (bf ? "gemm_nt_bf16_128x128_dma" : "gemm_nt_f32_128x128_dma"),
                                                   flops, bytes);
    case 128128:
      return launch<T, 128, 128, CONV>(a, s, CONV ? (bf ? "conv_igemm_bf16_128x128" : "conv_igemm_f32_128x128")
                                                 : (bf ? "gemm_nt_bf16_128x128" : "gemm_nt_f32_128x128"),
                                       flops, bytes);
    case 128064:
      return launch<T, 128, 64, CONV>(a, s, CONV ? (bf ? "conv_igemm_bf16_128x64" : "conv_igemm_f32_128x64")
                                                : (bf ? "gemm_nt_bf16_128x64" : "gemm_nt_f32_128x64"),
                                      flops, bytes);
    case 64128:
      return launch<T, 64, 128, CONV>(a, s, CONV ? (bf ? "conv_igemm_bf16_64x128" : "conv_igemm_f32_64x128")
                                                : (bf ? "gemm_nt_bf16_64x128" : "gemm_nt_f32_64x128"),
                                      flops, bytes);
    case 64064:
      return launch<T, 64, 64, CONV>(a, s, CONV ? (bf ? "conv_igemm_bf16_64x64" : "conv_igemm_f32_64x64")
                                               : (bf ? "gemm_nt_bf16_64x64" : "gemm_nt_f32_64x64"),
                                     flops, bytes);
    default:
      wsovod::set_error("wsovod_gemm_nt: unknown tile_hint %d", tile);
      return WSOVOD_ERR_INVALID_ARGUMENT;
  }
}

// bf16x2 operands (X3 kernels): the tiles the "parity" precision uses.  Slot names carry "bf16x2"; their FLOP figures
// are the EXECUTED bf16 MFMA FLOPs (3 x 2 M N K of the layer).
template <bool CONV>
int dispatch_tile_x3(const GemmArgs& a, int tile, hipStream_t s, double flops, double bytes) {
  switch (tile) {
    case 8256256:
      return launch_gemm256_8ph(a, CONV, s, flops, bytes, false, true);
    case 2256256:  // the two-phase ("merged") form of that tile
      return launch_gemm256_8ph(a, CONV, s, flops, bytes, false, true, true);
    case 256256:
      return launch<bf16_t, 256, 256, CONV, 4, 4, true, 2, true>(a, s, CONV ? "conv_igemm_bf16x2_256x256" : "gemm_nt_bf16x2_256x256",
                                                               flops, bytes);
    case 256128:
      return launch<bf16_t, 256, 128, CONV, 4, 2, true, 2, true>(a, s, CONV ? "conv_igemm_bf16x2_256x128" : "gemm_nt_bf16x2_256x128",
                                                               flops, bytes);
    case 512128:  // round 5, the 128-channel convs of res3: 512 pixels x 128 channels, 16 wavefronts as 8 x 2 (64 x 64 each,
                  // the register profile of the 256x256 tile), 160 KiB of LDS -- 48 MFMAs per wavefront and K-step where the
                  // 256x128 tile has 24 for the same barrier, DMA issue and fragment-read latency
      if constexpr (CONV)
        return launch<bf16_t, 512, 128, true, 8, 2, true, 2, true>(a, s, "conv_igemm_bf16x2_512x128", flops, bytes);
      wsovod::set_error("wsovod_gemm_nt: tile 512128 is an implicit-GEMM conv tile");
      return WSOVOD_ERR_UNSUPPORTED;
    case 1256064:
      return launch<bf16_t, 256, 64, CONV, 4, 1, true, 2, true>(a, s, CONV ? "conv_igemm_bf16x2_256x64" : "gemm_nt_bf16x2_256x64",
                                                              flops, bytes);
    case 1128064:
      return launch<bf16_t, 128, 64, CONV, 2, 2, true, 2, true>(a, s, CONV ? "conv_igemm_bf16x2_128x64" : "gemm_nt_bf16x2_128x64",
                                                              flops, bytes);
    case 3128064:  // round 6: three DMA stages (72 KiB: still two workgroups per CU)
      return launch<bf16_t, 128, 64, CONV, 2, 2, true, 3, true>(a, s, CONV ? "conv_igemm_bf16x2_128x64_s3" : "gemm_nt_bf16x2_128x64_s3",
                                                              flops, bytes);
    case 3064064:  // round 6: the 64x64 tile by LDS-DMA, four stages (64 KiB: two workgroups per CU)
      return launch<bf16_t, 64, 64, CONV, 2, 2, true, 4, true>(a, s, CONV ? "conv_igemm_bf16x2_64x64_s4" : "gemm_nt_bf16x2_64x64_s4",
                                                             flops, bytes);
    case 1128128:
    case 128128:
      return launch<bf16_t, 128, 128, CONV, 2, 2, true, 2, true>(a, s, CONV ? "conv_igemm_bf16x2_128x128" : "gemm_nt_bf16x2_128x128",
                                                               flops, bytes);
    case 64064:
    default:
      return launch<bf16_t, 64, 64, CONV, 2, 2, false, 2, true>(a, s, CONV ? "conv_igemm_bf16x2_64x64" : "gemm_nt_bf16x2_64x64",
                                                              flops, bytes);
  }
}

// Tile choice (measured on MI355X, tools/probe_kernels.py): the 256x256 LDS-direct tile moves 128 FLOP
// per staged byte and wins whenever it still yields about one workgroup per CU; smaller problems
// step down to tiles that keep the 256 CUs busy.
int auto_tile(int M, int N) {
  auto tiles = [&](int bm, int bn) { return (long long)ceil_div(M, bm) * ceil_div(N, bn); };
  if (N <= 64) {
    if (tiles(256, 64) >= 230) return 1256064;
    if (tiles(128, 64) >= 230) return 1128064;
    return 64064;
  }
  if (M > 128 && N > 128 && tiles(256, 256) >= 230) return 256256;
  if (M > 128 && tiles(256, 128) >= 230) return 256128;
  if (tiles(128, 128) >= 230) return 1128128;
  if (N <= 128 && tiles(128, 64) >= 230) return 1128064;
  return 64064;
}

}  // namespace wsovod_gemm
using namespace wsovod_gemm;

extern "C" int wsovod_gemm_nt(const wsovod_gemm_desc* d, wsovod_stream_t stream) {
  WS_CHECK_ARG(d != nullptr, "wsovod_gemm_nt: null descriptor");
  WS_CHECK_ARG(d->dtype_in == WSOVOD_F32 || d->dtype_in == WSOVOD_BF16 || d->dtype_in == WSOVOD_BF16X2,
               "wsovod_gemm_nt: bad dtype_in %d", d->dtype_in);
  const bool x2 = d->dtype_in == WSOVOD_BF16X2;  // bf16x2 operands = bf16 matrices of twice the length for the kernels
  WS_CHECK_ARG(d->M >= 0 && d->N >= 0 && d->K >= 0, "wsovod_gemm_nt: negative dimension");
  if (d->M == 0 || d->N == 0) return WSOVOD_OK;
  WS_CHECK_ARG(d->A && d->B, "wsovod_gemm_nt: null operand");
  WS_CHECK_ARG(d->C || d->Ct, "wsovod_gemm_nt: no output");
  const int esz = d->dtype_in == WSOVOD_F32 ? 4 : 2;
  const int epc = x2 ? 32 : 16 / esz;  // bf16x2: whole 32-value groups
  const int xs = x2 ? 2 : 1;           // bf16 slots per value
  WS_CHECK_ARG(((uintptr_t)d->A & 15) == 0 && ((uintptr_t)d->B & 15) == 0, "wsovod_gemm_nt: A/B must be 16-byte aligned");
  WS_CHECK_ARG(d->ldb % (x2 ? 4 : epc) == 0, "wsovod_gemm_nt: ldb=%lld must be a multiple of %d elements", d->ldb, x2 ? 4 : epc);
  WS_CHECK_ARG(128ll * d->ldb * esz * xs < (1ll << 31), "wsovod_gemm_nt: ldb too large for buffer addressing");
  WS_CHECK_ARG(d->K % epc == 0, "wsovod_gemm_nt: K=%d must be a multiple of %d elements", d->K, epc);
  WS_CHECK_ARG(!x2 || (d->dropout_p == 0.f || d->C), "wsovod_gemm_nt: bad bf16x2 call");
  WS_CHECK_ARG(!d->accumulate || (d->C && d->dtype_c == WSOVOD_F32), "wsovod_gemm_nt: accumulate needs an fp32 C");
  WS_CHECK_ARG(d->dropout_p >= 0.f && d->dropout_p < 1.f, "wsovod_gemm_nt: dropout_p must be in [0,1)");
  WS_CHECK_ARG(!d->group_add || d->row_group, "wsovod_gemm_nt: group_add needs row_group");

  GemmArgs a;
  memset(&a, 0, sizeof(a));
  a.A = (const char*)d->A;
  a.B = (const char*)d->B;
  const bool planar = d->a_plane_bytes != 0;
  WS_CHECK_ARG(!planar || (x2 && !d->conv && d->a_plane_bytes > 0 && d->a_plane_bytes % 16 == 0 &&
                           d->a_plane_bytes + 256ll * d->lda * 2 < (1ll << 31) && d->lda % 8 == 0),
               "wsovod_gemm_nt: a_plane_bytes needs a bf16x2 plain GEMM, 16-byte aligned planes and planes + 256 rows < 2 GiB");
  a.a_plane = d->a_plane_bytes;
  a.lda = planar ? d->lda : d->lda * xs;  // (a plane row holds lda bf16 values; the interleaved row 2 * lda slots)
  a.ldb = d->ldb * xs;
  a.M = d->M;
  a.N = d->N;
  a.K = d->K * xs;
  a.C = d->C;
  a.ldc = d->ldc;
  a.dtype_c = d->dtype_c;
  a.Ct = d->Ct;
  a.ldct = d->ldct;
  a.dtype_ct = d->dtype_ct;
  a.alpha = d->alpha;
  a.row_scale = d->row_scale;
  a.bias = d->bias;
  a.residual = d->residual;
  a.ldr = d->ldr;
  a.dtype_r = d->dtype_r;
  a.relu = d->relu;
  a.dropout_p = d->dropout_p;
  a.seed = d->dropout_seed;
  a.seed_add = d->dropout_seed_add;
  a.row_group = d->row_group;
  a.group_add = d->group_add;
  a.ld_ga = d->ld_ga;
  a.mask_src = d->mask_src;
  a.ldm = d->ldm;
  a.dtype_m = d->dtype_m;
  a.mask_scale = d->mask_scale;
  a.accumulate = d->accumulate;

  double bytes;
  if (d->conv) {
    const wsovod_conv_geom& g = d->geom;
    const int bke = d->dtype_in == WSOVOD_BF16 ? 64 : 32;  // values per K-step (bf16x2: 32 values = 64 bf16 slots)
    WS_CHECK_ARG(g.Cin > 0 && g.Cin % bke == 0, "wsovod_gemm_nt(conv): Cin=%d must be a multiple of %d", g.Cin, bke);
    WS_CHECK_ARG(!d->A2 || (d->Cin2 > 0 && d->Cin2 % bke == 0 && ((uintptr_t)d->A2 & 15) == 0),
                 "wsovod_gemm_nt(conv): the fused shortcut input needs Cin2 (%d) a multiple of %d and 16-byte alignment", d->Cin2, bke);
    WS_CHECK_ARG(d->K == g.KH * g.KW * g.Cin + (d->A2 ? d->Cin2 : 0), "wsovod_gemm_nt(conv): K=%d != KH*KW*Cin (+ Cin2)", d->K);
    WS_CHECK_ARG(g.KH * g.KW <= 32, "wsovod_gemm_nt(conv): filters of more than 32 taps are not supported (per-tap validity mask)");
    WS_CHECK_ARG((long long)d->M == (long long)g.n_img * g.Ho * g.Wo, "wsovod_gemm_nt(conv): M=%d != n_img*Ho*Wo", d->M);
    WS_CHECK_ARG(g.stride >= 1 && g.dil >= 1 && g.pad >= 0, "wsovod_gemm_nt(conv): bad stride/dil/pad");
    a.H = g.H;
    a.W = g.W;
    a.Cin = g.Cin * xs;
    a.Ho = g.Ho;
    a.Wo = g.Wo;
    a.KH = g.KH;
    a.KW = g.KW;
    a.stride = g.stride;
    a.pad = g.pad;
    a.dil = g.dil;
    a.pool = g.pool;
    a.a_bytes = (long long)g.n_img * g.H * g.W * g.Cin * esz * xs;
    if (d->A2) {
      a.A2 = (const char*)d->A2;
      a.Cin2 = d->Cin2 * xs;
      a.a2_bytes = (long long)g.n_img * g.Ho * g.Wo * d->Cin2 * esz * xs;
      WS_CHECK_ARG(a.a2_bytes < (1ll << 31), "wsovod_gemm_nt(conv): fused shortcut input exceeds the 2 GiB buffer-addressing limit");
    }
    WS_CHECK_ARG(a.a_bytes < (1ll << 31), "wsovod_gemm_nt(conv): input of %lld bytes exceeds the 2 GiB buffer-addressing limit", a.a_bytes);
    bytes = ((double)g.n_img * g.H * g.W * g.Cin + (double)d->N * d->K + (d->A2 ? (double)g.n_img * g.Ho * g.Wo * d->Cin2 : 0.0)) * esz * xs;
  } else {
    WS_CHECK_ARG(d->lda % (x2 ? 4 : epc) == 0, "wsovod_gemm_nt: lda=%lld must be a multiple of %d elements", d->lda, x2 ? 4 : epc);
    WS_CHECK_ARG(128ll * d->lda * esz * xs < (1ll << 31), "wsovod_gemm_nt: lda too large for buffer addressing");
    bytes = ((double)d->M * d->K + (double)d->N * d->K) * esz * xs;
  }
  bytes += (double)d->M * d->N * ((d->C ? (d->dtype_c == WSOVOD_BF16 ? 2 : 4) : 0) + (d->Ct ? (d->dtype_ct == WSOVOD_BF16 ? 2 : 4) : 0));
  WS_CHECK_ARG((d->dtype_c != WSOVOD_BF16X2 || !d->C || (d->N % 32 == 0 && d->ldc % 32 == 0 && ((uintptr_t)d->C & 15) == 0)) &&
                   d->dtype_ct != WSOVOD_BF16X2 && d->dtype_m != WSOVOD_BF16X2,
               "wsovod_gemm_nt: a bf16x2 output needs N and ldc multiples of 32 (Ct / mask_src cannot be bf16x2)");
  WS_CHECK_ARG(d->dtype_r != WSOVOD_BF16X2 || !d->residual || (d->N % 32 == 0 && d->ldr % 32 == 0 && ((uintptr_t)d->residual & 15) == 0),
               "wsovod_gemm_nt: a bf16x2 residual needs N and ldr multiples of 32");
  const double flops = (x2 ? 6.0 : 2.0) * d->M * d->N * d->K;  // bf16x2: three bf16 MFMA products per value pair
  hipStream_t s = (hipStream_t)stream;
  if (d->conv && d->tile_hint == 0 && d->dtype_in == WSOVOD_BF16 && a.Cin == 64 && d->N == 64 && a.KH == 3 &&
      a.KW == 3 && a.stride == 1 && a.pad == 1 && a.dil == 1 && a.Ho == a.H && a.Wo == a.W && d->C && !d->Ct && !d->A2 &&
      !d->row_scale && !d->group_add && !d->mask_src && !d->accumulate && d->dropout_p == 0.f) {
    static int slot = wsovod::prof_slot("conv3x3_c64_halo_bf16");
    if (a.pool) {
      WS_CHECK_ARG(a.pool == 2 && d->dtype_c == WSOVOD_BF16 && (d->ldc & 7) == 0 && ((uintptr_t)d->C & 15) == 0 &&
                       (!d->bias || ((uintptr_t)d->bias & 15) == 0) &&
                       (!d->residual || (d->dtype_r == WSOVOD_BF16 && (d->ldr & 7) == 0 && ((uintptr_t)d->residual & 15) == 0)),
                   "wsovod_gemm_nt(conv, pool): the fused 2x2 max pool needs bf16, 16-byte aligned output / residual");
    }
    const int lds_bytes = C64_PATCH_BYTES + 2 * 8192;
    const int tiles_x = ceil_div(a.W, C64_TW), tiles_y = ceil_div(a.H, C64_TH);
    const int n_tiles = d->geom.n_img * tiles_x * tiles_y;
    const char* pe = getenv("WSOVOD_C64_PERSIST");  // "0": always the one-tile kernel (A/B runs, tests)
    const bool persist = !(pe && pe[0] == '0');
    wsovod::ProfScope prof(slot, s, flops, bytes);
    if (persist && n_tiles >= 512) {  // enough tiles for two per CU: resident weights + double-buffered patches
      static bool attr = false;
      if (!attr) {
        WS_CHECK_HIP(hipFuncSetAttribute((const void*)conv3x3_c64p_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, C64P_LDS_BYTES), "wsovod_gemm_nt: LDS opt-in");
        WS_CHECK_HIP(hipFuncSetAttribute((const void*)conv3x3_c64p_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, C64P_LDS_BYTES), "wsovod_gemm_nt: LDS opt-in");
        attr = true;
      }
      const char* wr = getenv("WSOVOD_C64_WREG");  // "0": weights read from LDS at every step (the round-2 form; A/B runs)
      // the register-weights form: bf16 output, 16-byte aligned rows, buffer-addressable output / residual
      const bool lean = d->dtype_c == WSOVOD_BF16 && (d->ldc & 7) == 0 && ((uintptr_t)d->C & 15) == 0 &&
                        (!d->bias || ((uintptr_t)d->bias & 3) == 0) && (double)d->M * d->ldc * 2 < 2147483648.0 &&
                        (!d->residual || (d->dtype_r == WSOVOD_BF16 && (d->ldr & 7) == 0 && ((uintptr_t)d->residual & 15) == 0 &&
                                          (double)d->M * d->ldr * 2 < 2147483648.0)) &&
                        (double)(n_tiles + 3 * 256) * (tiles_x * tiles_y) < 4294967296.0;
#if (C64P_ABL & 64)
      if (const char* dp = getenv("WSOVOD_C64_DEBUG_PTR")) a.partial = (float*)strtoull(dp, nullptr, 16);
#endif
      if ((wr && wr[0] == '0') || !lean)
        hipLaunchKernelGGL(conv3x3_c64p_kernel<false>, dim3(256), dim3(256), C64P_LDS_BYTES, s, a, tiles_x, tiles_y, n_tiles);
      else
        hipLaunchKernelGGL(conv3x3_c64p_kernel<true>, dim3(256), dim3(256), C64P_LDS_BYTES, s, a, tiles_x, tiles_y, n_tiles);
      WS_CHECK_LAUNCH("wsovod_gemm_nt(conv3x3_c64 persistent)");
      return WSOVOD_OK;
    }
    hipLaunchKernelGGL(conv3x3_c64_kernel, dim3(n_tiles), dim3(256), lds_bytes, s, a, tiles_x,
                       tiles_y);
    WS_CHECK_LAUNCH("wsovod_gemm_nt(conv3x3_c64)");
    return WSOVOD_OK;
  }
  if (d->conv && d->tile_hint == 0 && x2 && d->geom.Cin == 64 && d->N == 64 && a.KH == 3 && a.KW == 3 && a.stride == 1 &&
      a.pad == 1 && a.dil == 1 && a.Ho == a.H && a.Wo == a.W && d->C && d->dtype_c == WSOVOD_BF16X2 && !d->Ct && !d->A2 &&
      !d->row_scale && !d->group_add && !d->mask_src && !d->accumulate && d->dropout_p == 0.f && d->ldc % 32 == 0 &&
      (!d->residual || (d->dtype_r == WSOVOD_BF16X2 && d->ldr % 32 == 0)) && (a.pool == 0 || a.pool == 2) &&
      !(getenv("WSOVOD_C64_X3") && getenv("WSOVOD_C64_X3")[0] == '0')) {
    static int slot = wsovod::prof_slot("conv3x3_c64_halo_bf16x2");
    static bool attr = false;
    if (!attr) {
      WS_CHECK_HIP(hipFuncSetAttribute((const void*)conv3x3_c64_x3_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, C64X<16>::LDS_BYTES), "wsovod_gemm_nt: LDS opt-in");
      WS_CHECK_HIP(hipFuncSetAttribute((const void*)conv3x3_c64_x3_kernel<32>, hipFuncAttributeMaxDynamicSharedMemorySize, C64X<32>::LDS_BYTES), "wsovod_gemm_nt: LDS opt-in");
      WS_CHECK_HIP(hipFuncSetAttribute((const void*)conv3x3_c64_x3h_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, C64XH::LDS_BYTES), "wsovod_gemm_nt: LDS opt-in");
      WS_CHECK_HIP(hipFuncSetAttribute((const void*)conv3x3_c64_x3h_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, C64XH::LDS_BYTES), "wsovod_gemm_nt: LDS opt-in");
      attr = true;
    }
    const char* tw = getenv("WSOVOD_C64X_TW");  // "16" / "32": the whole-K forms (A/B runs); default: the half-K 8 x 32 tile
    const bool wide = tw && tw[0] == '3', halfk = !tw || (tw[0] != '3' && tw[0] != '1');
    const int tiles_x = ceil_div(a.W, wide || halfk ? 32 : 16), tiles_y = ceil_div(a.H, C64_TH);
    const int n_tiles = d->geom.n_img * tiles_x * tiles_y;
#if C64XH_PHASES
    if (const char* dp = getenv("WSOVOD_C64_DEBUG_PTR")) a.partial = (float*)strtoull(dp, nullptr, 16);
#endif
    wsovod::ProfScope prof(slot, s, flops, bytes);
    // epilogue through LDS (whole-pixel stores): output rows of exactly one pixel's 256 B ("0": direct stores; A/B runs)
    static const bool lepi_on = !(getenv("WSOVOD_C64X_LEPI") && getenv("WSOVOD_C64X_LEPI")[0] == '0');
    if (halfk && lepi_on && ((uintptr_t)d->C & 15) == 0 && d->ldc % 4 == 0)
      hipLaunchKernelGGL(conv3x3_c64_x3h_kernel<true>, dim3(n_tiles), dim3(256), C64XH::LDS_BYTES, s, a, tiles_x, tiles_y);
    else if (halfk)
      hipLaunchKernelGGL(conv3x3_c64_x3h_kernel<false>, dim3(n_tiles), dim3(256), C64XH::LDS_BYTES, s, a, tiles_x, tiles_y);
    else if (wide)
      hipLaunchKernelGGL(conv3x3_c64_x3_kernel<32>, dim3(n_tiles), dim3(256), C64X<32>::LDS_BYTES, s, a, tiles_x, tiles_y);
    else
      hipLaunchKernelGGL(conv3x3_c64_x3_kernel<16>, dim3(n_tiles), dim3(256), C64X<16>::LDS_BYTES, s, a, tiles_x, tiles_y);
    WS_CHECK_LAUNCH("wsovod_gemm_nt(conv3x3_c64 bf16x2)");
    return WSOVOD_OK;
  }
  WS_CHECK_ARG(!(d->conv && d->geom.pool), "wsovod_gemm_nt(conv): geom.pool is an epilogue of the 64-channel 3x3 "
               "kernels only (stride 1, pad 1, dilation 1, no tile_hint)");
  int tile = d->tile_hint ? d->tile_hint : auto_tile(d->M, d->N);
  // plain bf16 contractions take the staggered 8-wavefront form of the 256x256 tile (measured +11-13 % on the FC
  // shapes, tools/gemm_ab.py); the implicit-GEMM conv stays on the 16-wavefront form (tools/conv_ab.py)
  // (round 5: in its lean two-phase form -- 1080 -> 1230, 705 -> 770, 1196 -> 1229 TFLOP/s against the four-phase form on
  // the dX shapes 16384 x 4096 x 4096 / x 1088 and 16384 x 25088 x 4096, bit-identical; tools/gemm_ab.py)
  static const int bf16_tile = getenv("WSOVOD_BF16_TILE") ? atoi(getenv("WSOVOD_BF16_TILE")) : 2256256;
  if (!d->tile_hint && tile == 256256 && d->dtype_in == WSOVOD_BF16 && !d->conv) tile = bf16_tile;
  // few rows, long K (the FC layers at 1-4 images per step): the same tile with split-K instead of a small-tile grid
  // (measured, M = 512: K = 25088 230 -> 143 us; at K = 4096 the workspace round trip costs more than it saves: 40 -> 53 us)
  if (!d->tile_hint && d->dtype_in == WSOVOD_BF16 && !d->conv && d->M >= 256 && d->N >= 256 && d->K >= 8192 &&
      (long long)ceil_div(d->M, 256) * ceil_div(d->N, 256) <= 128)
    tile = bf16_tile;
  a.ksplit = d->tile_hint == 0 ? -1 : 0;  // split-K may only change the summation order when the caller named no tile
  // Tile-round tail of the one-workgroup-per-CU 256x256 conv tile: the last, partly filled round of 256 tiles costs a whole
  // tile time (32 images of 75x100, 512 channels: 1876 tiles = 7.33 rounds -> 8).  When the leftover tiles, cut in halves
  // along N (256x128, still one workgroup per CU), fit ONE round, the rows behind the last full round go through a second
  // launch with that tile: ~0.6 of a tile time instead of 1.  Same products in the same order per output element.
  if (d->conv && !d->tile_hint && tile == 256256 && d->dtype_in != WSOVOD_F32 &&
      !(getenv("WSOVOD_CONV_TAIL") && getenv("WSOVOD_CONV_TAIL")[0] == '0')) {
    const int tm = ceil_div(d->M, 256), tn = ceil_div(d->N, 256);
    const long long tiles = (long long)tm * tn;
    const int rounds = (int)(tiles / 256);
    const int main_tm = (int)((long long)rounds * 256 / tn);
    const long long tail_rows = (long long)d->M - (long long)main_tm * 256;
    const long long tail_tiles = ceil_div((int)tail_rows, 256) * (long long)ceil_div(d->N, 128);
    if (rounds >= 1 && tail_rows > 0 && tail_tiles <= 256 && tiles - (long long)main_tm * tn > 0) {
      const double fmain = (double)main_tm * 256 / d->M;
      GemmArgs am = a, at = a;
      am.M = main_tm * 256;
      at.m_base = main_tm * 256;
      // bf16x2: the main rounds on the lean two-phase 8-wavefront tile (round 5: 1331 / 1452 / 1384 against 1270 / 1366 /
      // 1315 TFLOP/s executed for the 16-wavefront tile on res4 / res5 / res5a, tools/conv_x2_ab.py; WSOVOD_CONV_8PH=0: A/B)
      static const bool conv8 = !(getenv("WSOVOD_CONV_8PH") && getenv("WSOVOD_CONV_8PH")[0] == '0');
      int rc = x2 ? dispatch_tile_x3<true>(am, conv8 ? 2256256 : 256256, s, flops * fmain, bytes * fmain)
                  : dispatch_tile<bf16_t, true>(am, 256256, s, flops * fmain, bytes * fmain);
      if (rc != WSOVOD_OK) return rc;
      // round 5: a tail of few tiles and a long K (res5: 84 tiles, 72 K-steps) as split-K slices of the lean 8-wavefront
      // tile (3 x 84 workgroups of 24 K-steps + the finalize pass) instead of 168 half-width tiles of 72
      // (WSOVOD_CONV_TAIL_SPLITK=0: A/B runs; changes the summation order of the tail rows only)
      const bool tail_split = !(getenv("WSOVOD_CONV_TAIL_SPLITK") && getenv("WSOVOD_CONV_TAIL_SPLITK")[0] == '0');
      const int tail_t = ceil_div((int)tail_rows, 256) * tn, nk64 = ceil_div(d->K, 64);
      if (x2 && conv8 && tail_split && tail_t <= 128 && nk64 >= 32 && std::min(std::min(8, 256 / tail_t), nk64 / 16) >= 2)
        return dispatch_tile_x3<true>(at, 2256256, s, flops * (1.0 - fmain), bytes * (1.0 - fmain));
      return x2 ? dispatch_tile_x3<true>(at, 256128, s, flops * (1.0 - fmain), bytes * (1.0 - fmain))
                : dispatch_tile<bf16_t, true>(at, 256128, s, flops * (1.0 - fmain), bytes * (1.0 - fmain));
    }
  }
  if (x2) {
    // plain contractions: the 8-wavefront tile in its two-phase form (48 MFMAs per phase; measured on the fc1 / fc2 /
    // projection shapes: 6.63 -> 6.14, 1.12 -> 1.03, 0.274 -> 0.263 ms against the four-phase form, tools/x2_probe.py);
    // the implicit-GEMM convs stay on the 16-wavefront tile (res5: 2.32 vs 2.40 ms)
    if (!d->tile_hint && tile == 256256 &&
        (!d->conv || !(getenv("WSOVOD_CONV_8PH") && getenv("WSOVOD_CONV_8PH")[0] == '0')))
      tile = 2256256;  // (round 5: also the implicit-GEMM convs, in the lean form of that tile)
    // few rows, long K (fc1 at 1-4 images per step): the same tile with split-K, as for plain bf16 above -- a 64x64 grid
    // re-reads the 411-MB bf16x2 weight through L2 eight times over (M = 512: 0.80 ms at 130 TFLOP/s algorithmic)
    if (!d->tile_hint && !d->conv && d->M >= 256 && d->N >= 256 && d->K >= 8192 &&
        (long long)ceil_div(d->M, 256) * ceil_div(d->N, 256) <= 128)
      tile = 2256256;
    // skinny heads (the MIL [cls | det] rows, N = 2K <= 64, on the 4096 box features): the 64x64 grid is M / 64 workgroups
    // streaming 1 MB of A each behind a two-stage register pipeline (125 us at 32 images = 2.1 TB/s); split-K slices of the
    // 256-wide tile stream the same rows by DMA (the B rows past N are range-checked away: their MFMAs run on zeros)
    if (!d->tile_hint && !d->conv && d->N <= 64 && d->M >= 256 && d->K >= 2048 && ceil_div(d->M, 256) <= 128 &&
        !(getenv("WSOVOD_X2_SKINNY") && getenv("WSOVOD_X2_SKINNY")[0] == '0'))
      tile = 2256256;
    // few rows, shorter K (fc2, the stacked heads, the convs of res4 / res5 at 1 - 4 images): re-measured in round 5 on bf16x2
    // operands (tools/small_batch_tiles.py, profiles/r05_small_batch_tiles.md).  T = number of 256x256 tiles:
    //   T >= 150           one partly filled round of the 256x256 tile beats two of 256x128 (3 images, res5: 250 vs 337 us)
    //   T >= 100           256x128 (one round)
    //   128x64 grid >= 200 the four-wavefront 128x64 tile, two workgroups per CU (1 - 2 images: fc2 74 vs 78 us on the
    //                      64x64 grid, res4 57 vs 60, res5 124 vs 128 on 128x128)
    //   else, K >= 2048    split-K of the 256x256 tile (the stacked heads, N = 1088: 48 vs 70 us)
    if (!d->tile_hint && d->M >= 256 && d->N >= 256 && tile != 2256256 && tile != 256256) {
      auto tiles = [&](int bm, int bn) { return (long long)ceil_div(d->M, bm) * ceil_div(d->N, bn); };
      const char* cs = getenv("WSOVOD_CONV_SPLITK");  // "1": also convs may take the split form (experiments, tests)
      if (tiles(256, 256) >= 150) tile = 2256256;
      else if (tiles(256, 128) >= 200) tile = 256128;
      else if (tiles(128, 64) >= 200) tile = 1128064;
      else if (d->K >= 2048 && (!d->conv || (cs && cs[0] == '1'))) tile = 2256256;
    }
    // the 128-channel convs of res3 at >= 14 images: 512x128 tiles (tools/res3_ab.py, 32 images: 1029 -> 1220, 779 -> 908,
    // 666 -> 828 TFLOP/s executed against 256x128 on the plain / residual / stride-2 layer; bit-identical results)
    if (!d->tile_hint && d->conv && d->N > 64 && d->N <= 128 && (long long)ceil_div(d->M, 512) >= 200 &&
        !(getenv("WSOVOD_CONV_512") && getenv("WSOVOD_CONV_512")[0] == '0'))
      tile = 512128;
    if (!d->tile_hint && d->conv && getenv("WSOVOD_CONV_SPLITK") && getenv("WSOVOD_CONV_SPLITK")[0] == '1' && d->M >= 256 &&
        d->N >= 256 && d->K >= 2048 && (long long)ceil_div(d->M, 256) * ceil_div(d->N, 256) <= 128)
      tile = 2256256;
    // round 6: the small grids of 1 - 2 images per step behind a deep LDS-DMA pipeline (their K-steps were one DMA round
    // trip each); WSOVOD_X2_DEEP=0: the two-stage forms (A/B runs, bit-identical results)
    static const bool deep = !(getenv("WSOVOD_X2_DEEP") && getenv("WSOVOD_X2_DEEP")[0] == '0');
    if (!d->tile_hint && deep) {
      if (tile == 1128064) tile = 3128064;
      else if (tile == 64064 && d->conv && d->K >= 1024 && d->M >= 1024) tile = 3064064;
    }
    if (planar) {  // the planar A operand exists in the lean two-phase tile (split-K included)
      WS_CHECK_ARG(!d->tile_hint || d->tile_hint == 2256256, "wsovod_gemm_nt: a planar A operand takes tile 2256256 only");
      tile = 2256256;
    }
    return d->conv ? dispatch_tile_x3<true>(a, tile, s, flops, bytes) : dispatch_tile_x3<false>(a, tile, s, flops, bytes);
  }
  if (d->dtype_in == WSOVOD_BF16)
    return d->conv ? dispatch_tile<bf16_t, true>(a, tile, s, flops, bytes) : dispatch_tile<bf16_t, false>(a, tile, s, flops, bytes);
  return d->conv ? dispatch_tile<float, true>(a, tile, s, flops, bytes) : dispatch_tile<float, false>(a, tile, s, flops, bytes);
}

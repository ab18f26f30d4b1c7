// Weight-gradient contraction without operand transposes:  C[i][j] = alpha * sum_m P[m][i] * Q[m][j]   (bf16 in, fp32 out)
//
//   nn.Linear dW = dY^T X   (P = dY (rows = proposals, cols = out features), Q = X (rows = proposals, cols = in features);
//   both ROW-MAJOR as the forward pass left them -- the reduction runs over the SLOW index of both operands).
//
// An MFMA fragment wants 8 consecutive reduction elements per lane, which here are 8 different ROWS of the operand.
// gfx950's ds_read_b64_tr_b16 does that transpose in the LDS read path (cdna_hip_programming.md T10): per 16-lane
// group it reads a 4-row x 16-column block and hands lane i column i of the 4 rows.  So the operand tiles are staged
// row-major exactly as they lie in HBM (LDS-direct DMA, 256-byte sub-rows, the guide's conflict-free image (b):
// off = 256*row + 16*(chunk ^ (((row&3)<<2) | ((row>>2)&3))), applied to the DMA SOURCE address and to the read
// address) and every fragment is two transposed 8-byte reads.  Replaces the x^T / dY^T copies (transpose_cast,
// mask_transpose's second output) that the NT kernel needed: 0.5 ms of an 9.6 ms step.
//
// Shape of the kernel = gemm8.hip: 256x256 output tile, 8 wavefronts as 2 (i) x 4 (j), 128x64 per wavefront, two groups
// staggered by one barrier, MFMAs at raised priority, counted vmcnt.  A K-step is 64 reduction rows = two PHASES of
// 32 rows; a phase reads its 8 + 4 fragments (24 transposed reads), issues the DMA of the same 32-row half of the NEXT
// K-step (4 instructions: P/Q x two 128-column sub-images) and waits vmcnt(4), i.e. for the half issued one phase
// earlier, which is read one phase later.  A row half is overwritten two phases after its last read.
#include "gemm_common.h"
#include "f16mx.h"
#include <vector>

namespace wsovod_gemm {

namespace {

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

struct TnArgs {
  const char* P;
  const char* Q;
  long long ldp, ldq;  // elements
  int Mred, NI, NJ;
  float* C;
  long long ldc;
  float alpha;
  int accumulate;
  int tiles_i, tiles_j, group_m;
  // split-K of the LAST, partial round of tiles: blocks [0, full_tiles) own a whole tile; after them every remaining tile
  // is cut into `ksplit` slices of `slice_steps` K-steps whose partial sums meet by fp32 atomic adds (0 = no split)
  int full_tiles, ksplit, slice_steps;
#if defined(TN_STAMPS)
  float* dbg;  // instrumented builds only (tools/tn_phases.py): per-group cycle sums of the phase sections
#endif
  int q_x2;  // Q is a bf16x2 matrix (include/wsovod_hip.h): only the hi halves of its values are read, column k at bf16
             // slot 64 (k / 32) + k % 32 of the row; ldq is then counted in bf16 slots (2 per value)
  // SGD form (wsovod_gemm_tn_sgd): C is the PARAMETER itself; the tile's gradient never goes to memory
  float* mom;            // momentum buffer, the parameter's shape
  bf16_t* shadow;        // optional bf16 / bf16x2 operand copy of the parameter, refreshed in the same pass
  int shadow_x2;         // 1: bf16x2; 2: f16mx with the per-tensor E8M0 byte *mx_scale (round 6)
  const unsigned char* mx_scale;
  float lr, wd, mu, gscale;
  const float* lr_dev;   // optional device scalar read instead of lr
  // SGD form with a split tile-round tail: the K slices of the tail tiles meet by atomics in a COMPACT scratch
  // [tail tile][256][256] fp32 (TAILBUF instantiation, launched as its own grid with bid_base = full_tiles), and
  // tn_sgd_tail_kernel applies the update of those tiles from it
  float* tail_buf;
  int bid_base;
};

// LEAN = 1 (round 5): the same schedule with the per-phase address arithmetic removed from the half of a phase that the
// other group's MFMAs have to cover (s_memtime stamps, tools/tn_phases.py: 24 reads + 4 DMA pieces took ~700 ticks
// against the 512 of 32 MFMAs, and every v_add of it competes with the other wave's MFMAs for the SIMD's issue port):
//  * LDS as [P: K-step 0 | K-step 1][Q: K-step 0 | K-step 1] (an operand's two buffers inside 64 KiB): the buffer and
//    the 32-row half of a read are the instruction's IMMEDIATE offset (K loop unrolled by two), the 24 per-lane address
//    registers are loop constants -- no v_add per read;
//  * DMA source offsets advance by a running 32-bit add per phase (rows beyond the reduction and columns beyond the
//    matrix fall to the buffer resource's range check: such a lane starts at 2^31) -- no 64-bit multiply, no selects.
template <int OFF>
__device__ __forceinline__ void tr_read_imm(__attribute__((ext_vector_type(2))) unsigned int& dst, unsigned addr) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF));
#endif
}

template <int LEAN, bool SGD = false, bool TAILBUF = false>
__global__ __launch_bounds__(512) void gemm_tn8_kernel(const TnArgs p) {
  constexpr int BI = 256, BJ = 256, BK = 64;
  constexpr int OP_BYTES = 2 * BK * 256;        // one operand of one K-step: 2 sub-images x 64 rows x 256 B
  constexpr int STEP_BYTES = 2 * OP_BYTES;      // P + Q
  // byte strides of the staging image: buffer (K-step parity) and operand
  constexpr int BUF_STRIDE = LEAN ? OP_BYTES : STEP_BYTES, OPQ_BASE = LEAN == 2 ? 2 * 5 * 8192 : LEAN ? 2 * OP_BYTES : OP_BYTES;
  // LEAN = 2: a ring of FIVE 32-row slots (160 KiB) laid out [operand][sub-image][slot][32 rows][256 B] -- a (operand,
  // sub-image) region spans 40 KiB, so the slot is still an immediate of the reads -- and the DMA runs THREE phases ahead
  // of its readers instead of one: with one phase (~700 cycles) of lead the counted wait in front of the barrier still
  // stalled 125 - 230 cycles per phase on the pieces' L2 / HBM latency (tools/tn_phases.py).
  constexpr int REG = 5 * 32 * 256;  // bytes of one (operand, sub-image) region
  constexpr int SUB_STRIDE = LEAN == 2 ? REG : BK * 256;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  int wg, slice = 0;
  const int bid = TAILBUF ? (int)blockIdx.x + p.bid_base : (int)blockIdx.x;
  {
    const int nwg = p.ksplit ? p.full_tiles : p.tiles_i * p.tiles_j;  // tiles that are remapped per XCD
    if (bid < nwg) {
      const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
      wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    } else {  // a slice of a tail tile
      const int t = bid - nwg;
      wg = nwg + t / p.ksplit;
      slice = t - (t / p.ksplit) * p.ksplit;
    }
  }
  const bool sliced = p.ksplit && bid >= p.full_tiles;
  const int group_size = p.group_m * p.tiles_j;
  const int group_id = wg / group_size;
  const int first_i = group_id * p.group_m;
  const int gm = min(p.tiles_i - first_i, p.group_m);
  const int in_group = wg - group_id * group_size;
  const int tile_i = first_i + in_group % gm;
  const int tile_j = in_group / gm;
  const int i0 = tile_i * BI, j0 = tile_j * BJ;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 2, wc = wave & 3;

  // ---- DMA lane geometry: a pass = 32 rows x 256 B of one sub-image; lane -> (row tid>>4, 16-byte slot tid&15)
  const int lrow = tid >> 4, lslot = tid & 15;
  const int lchunk = lslot ^ (((lrow & 3) << 2) | ((lrow >> 2) & 3));  // inverse swizzle on the source (rows r, r+32 agree)
  // column offsets (bytes) of this lane's chunk in the two sub-images; <0 = beyond the matrix (zero fill)
  int pcol[2], qcol[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int ci = i0 + s * 128 + lchunk * 8, cj = j0 + s * 128 + lchunk * 8;
    pcol[s] = ci < p.NI ? ci * 2 : -1;
    qcol[s] = cj < p.NJ ? (p.q_x2 ? (((cj >> 5) << 6) | (cj & 31)) : cj) * 2 : -1;
  }
  typedef __attribute__((address_space(3))) void lds_void [[maybe_unused]];
  [[maybe_unused]] const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const int nk_all = (p.Mred + BK - 1) / BK;
  const int kt0 = sliced ? slice * p.slice_steps : 0;
  const int nk = sliced ? min(nk_all, kt0 + p.slice_steps) : nk_all;  // this block reduces K-steps [kt0, nk)
  if (kt0 >= nk) return;  // (an empty slice: the even slice length of the host may leave the last one without K-steps)
  // LEAN = 2 stages whole phases past the end of its K range (ring bookkeeping without branches): rows behind the range
  // must read as zeros, so the resources end at the block's last row
  const long long row_end = LEAN == 2 ? min((long long)p.Mred, (long long)nk * BK) : (long long)p.Mred;
  [[maybe_unused]] const __amdgpu_buffer_rsrc_t rsrcP = __builtin_amdgcn_make_buffer_rsrc(
      (void*)p.P, 0, (int)min(row_end * p.ldp * 2, (long long)0x7fffffff), 0x00020000);
  [[maybe_unused]] const __amdgpu_buffer_rsrc_t rsrcQ = __builtin_amdgcn_make_buffer_rsrc(
      (void*)p.Q, 0, (int)min(row_end * p.ldq * 2, (long long)0x7fffffff), 0x00020000);
  // stage rows [32*half, 32*half+32) of K-step kt into buffer buf: 4 DMA instructions
  // LEAN: running byte offsets of this lane's pieces in the NEXT 32-row block to stage (blocks are staged in order)
  [[maybe_unused]] unsigned voP[2], voQ[2];
  [[maybe_unused]] const unsigned stepP = (unsigned)(32ll * p.ldp * 2), stepQ = (unsigned)(32ll * p.ldq * 2);
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const long long r0 = (long long)kt0 * BK + lrow;
    voP[s] = pcol[s] >= 0 ? (unsigned)(r0 * p.ldp * 2 + pcol[s]) : 0x80000000u;
    voQ[s] = qcol[s] >= 0 ? (unsigned)(r0 * p.ldq * 2 + qcol[s]) : 0x80000000u;
  }
  // LEAN = 2: the next 32-row block into ring slot `slot`
  auto stage_slot = [&](int slot) {
#if defined(__HIP_DEVICE_COMPILE__)
    char* d = smem + slot * 8192 + wave_u * 1024;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcP, (lds_void*)(d + s * REG), 16, (int)voP[s], 0, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcQ, (lds_void*)(d + OPQ_BASE + s * REG), 16, (int)voQ[s], 0, 0, 0);
      voP[s] += stepP;
      voQ[s] += stepQ;
    }
#endif
  };
  auto stage_half = [&](int kt, int buf, int half) {
#if defined(__HIP_DEVICE_COMPILE__)
    char* d = smem + buf * BUF_STRIDE + half * 32 * 256 + wave_u * 1024;
    if (LEAN) {
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcP, (lds_void*)(d + s * (BK * 256)), 16, (int)voP[s], 0, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcQ, (lds_void*)(d + OPQ_BASE + s * (BK * 256)), 16, (int)voQ[s], 0, 0, 0);
        voP[s] += stepP;
        voQ[s] += stepQ;
      }
      return;
    }
    const int m = kt * BK + half * 32 + lrow;
    const bool ok = m < p.Mred;
    const long long rp = (long long)m * p.ldp * 2, rq = (long long)m * p.ldq * 2;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcP, (lds_void*)(d + s * (BK * 256)), 16,
                                               (ok && pcol[s] >= 0) ? (int)(rp + pcol[s]) : -1, 0, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcQ, (lds_void*)(d + OPQ_BASE + s * (BK * 256)), 16,
                                               (ok && qcol[s] >= 0) ? (int)(rq + qcol[s]) : -1, 0, 0, 0);
    }
#endif
  };

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- transposed-read lane geometry (T10): within its 16-lane group, lane 4q+pp supplies the address of block row q,
  // columns 4pp..4pp+3; the group g = lane>>4 covers reduction rows 8g..8g+7 of a 32-row phase as two 4-row blocks.
  const int g = lane >> 4, q4 = (lane & 15) >> 2, pp = lane & 3;
  [[maybe_unused]] unsigned offA[2][8], offB[2][4];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int row = 8 * g + 4 * h + q4;  // row inside the 32-row half
    const int swz = ((row & 3) << 2) | ((row >> 2) & 3);
#pragma unroll
    for (int t = 0; t < 8; ++t)
      offA[h][t] = (unsigned)(wr * SUB_STRIDE + row * 256 + (((2 * t + (pp >> 1)) ^ swz) << 4) + 8 * (pp & 1));
#pragma unroll
    for (int t = 0; t < 4; ++t)
      offB[h][t] = (unsigned)(OPQ_BASE + (wc >> 1) * SUB_STRIDE + row * 256 +
                              (((2 * ((wc & 1) * 4 + t) + (pp >> 1)) ^ swz) << 4) + 8 * (pp & 1));
  }

  u32x2 al[8], ah[8], bl[4], bh[4];  // low / high 4 reduction elements of every fragment
#if defined(__HIP_DEVICE_COMPILE__)
  typedef __attribute__((address_space(3))) const char lds_cchar;
  const unsigned lds0 = (unsigned)(size_t)(lds_cchar*)smem;
#define WS_TR_READ(dst, addr) asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(dst) : "v"(addr))
#define WS_LGKM0_ALL()                                                                                               \
  asm volatile("s_waitcnt lgkmcnt(0)"                                                                                \
               : "+v"(al[0]), "+v"(al[1]), "+v"(al[2]), "+v"(al[3]), "+v"(al[4]), "+v"(al[5]), "+v"(al[6]), "+v"(al[7]), \
                 "+v"(ah[0]), "+v"(ah[1]), "+v"(ah[2]), "+v"(ah[3]), "+v"(ah[4]), "+v"(ah[5]), "+v"(ah[6]), "+v"(ah[7]), \
                 "+v"(bl[0]), "+v"(bl[1]), "+v"(bl[2]), "+v"(bl[3]), "+v"(bh[0]), "+v"(bh[1]), "+v"(bh[2]), "+v"(bh[3]))
#define WS_VMCNT(N) asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory")
#else
  const unsigned lds0 = 0;
#define WS_TR_READ(dst, addr) (void)0
#define WS_LGKM0_ALL() (void)0
#define WS_VMCNT(N) (void)0
#endif

#if defined(TN_STAMPS) && defined(__HIP_DEVICE_COMPILE__)
  unsigned long long st_t = 0, st_acc[6] = {0, 0, 0, 0, 0, 0};
#define TN_STAMP0() st_t = __builtin_amdgcn_s_memtime()
#define TN_STAMP(k)                                              \
  {                                                              \
    const unsigned long long now = __builtin_amdgcn_s_memtime(); \
    st_acc[k] += now - st_t;                                     \
    st_t = now;                                                  \
  }
#else
#define TN_STAMP0() (void)0
#define TN_STAMP(k) (void)0
#endif
  // ---- prologue: K-step 0 completely, then the stagger barrier
  if (LEAN == 2) {
    stage_slot(0);
    stage_slot(1);
    stage_slot(2);
    WS_VMCNT(8);
  } else {
    stage_half(kt0, 0, 0);
    stage_half(kt0, 0, 1);
    WS_VMCNT(0);
  }
  __builtin_amdgcn_s_barrier();
  if (wr == 1) __builtin_amdgcn_s_barrier();  // stagger: the second i-half runs one barrier behind

  // A phase stages the SAME 32-row half of the next K-step (that half of the other buffer was last read two phases
  // ago) and waits vmcnt(4): the half issued one phase earlier has landed before the barrier its readers pass first;
  // it is read one phase later.  (A three-phase-deep ring with the reads retired before the barrier was measured
  // 18 % slower: the loop no longer unrolls over the two halves and the LDS latency moves in front of the barrier.)
  auto mfma_block = [&]() {
    TN_STAMP(0);
    __builtin_amdgcn_s_barrier();
    TN_STAMP(1);
    WS_LGKM0_ALL();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(  // Q fragment first: a lane holds 4 consecutive j
            __builtin_bit_cast(bf16x8, __builtin_shufflevector(bl[j], bh[j], 0, 1, 2, 3)),
            __builtin_bit_cast(bf16x8, __builtin_shufflevector(al[i], ah[i], 0, 1, 2, 3)), acc[i][j], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    TN_STAMP(2);
    __builtin_amdgcn_s_barrier();
    TN_STAMP(3);
  };
  if (LEAN == 2) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
      for (int t = 0; t < 8; ++t) offA[h][t] += lds0;
#pragma unroll
      for (int t = 0; t < 4; ++t) offB[h][t] += lds0;
    }
#endif
    // phase q reads slot q % 5 and refills slot (q + 3) % 5 (last read two phases ago) with the block of phase q + 3; the
    // wait leaves the two youngest blocks in flight.  The phase count is padded to a multiple of five: blocks past the
    // range are zero-filled by the resources' range check, their products add nothing
    const int nph = 2 * (nk - kt0), nph5 = (nph + 4) / 5 * 5;
    auto phase = [&](auto slot_c) {
      constexpr int SLOT = decltype(slot_c)::value;
      TN_STAMP0();
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        tr_read_imm<SLOT * 8192>(bl[t], offB[0][t]);
        tr_read_imm<SLOT * 8192>(bh[t], offB[1][t]);
      }
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        tr_read_imm<SLOT * 8192>(al[t], offA[0][t]);
        tr_read_imm<SLOT * 8192>(ah[t], offA[1][t]);
      }
      TN_STAMP(4);
      stage_slot((SLOT + 3) % 5);
      TN_STAMP(5);
      WS_VMCNT(8);
      mfma_block();
    };
    for (int q = 0; q < nph5; q += 5) {
      phase(std::integral_constant<int, 0>{});
      phase(std::integral_constant<int, 1>{});
      phase(std::integral_constant<int, 2>{});
      phase(std::integral_constant<int, 3>{});
      phase(std::integral_constant<int, 4>{});
    }
    WS_VMCNT(0);  // the refills issued by the last phases (zero blocks) must not land in the epilogue's staging area
  } else if (LEAN) {
#if defined(__HIP_DEVICE_COMPILE__)
    // the per-lane read addresses include the LDS base; buffer and half are immediates of the unrolled phases
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
      for (int t = 0; t < 8; ++t) offA[h][t] += lds0;
#pragma unroll
      for (int t = 0; t < 4; ++t) offB[h][t] += lds0;
    }
#endif
    // K-steps in pairs (static buffer index): an odd count is padded with one K-step beyond the reduction, which the DMA
    // zero-fills (rows >= Mred are out of the resource's range; a slice that is not the last has an even count: host)
    const int nkp = kt0 + ((nk - kt0 + 1) & ~1);
    auto phase = [&](auto imm, bool more, int buf, int half) {
      constexpr int IMM = decltype(imm)::value;  // buf * BUF_STRIDE + half * 8192
      TN_STAMP0();
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        tr_read_imm<IMM>(bl[t], offB[0][t]);
        tr_read_imm<IMM>(bh[t], offB[1][t]);
      }
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        tr_read_imm<IMM>(al[t], offA[0][t]);
        tr_read_imm<IMM>(ah[t], offA[1][t]);
      }
      TN_STAMP(4);
      if (more) {
        stage_half(0, buf ^ 1, half);
        TN_STAMP(5);
        WS_VMCNT(4);
      } else {
        WS_VMCNT(0);
      }
      mfma_block();
    };
    for (int kt = kt0; kt < nkp; kt += 2) {
      const bool more = kt + 2 < nkp;
      phase(std::integral_constant<int, 0>{}, true, 0, 0);
      phase(std::integral_constant<int, 32 * 256>{}, true, 0, 1);
      phase(std::integral_constant<int, BUF_STRIDE>{}, more, 1, 0);
      phase(std::integral_constant<int, BUF_STRIDE + 32 * 256>{}, more, 1, 1);
    }
  } else {
    for (int kt = kt0; kt < nk; ++kt) {
      const int cur = (kt - kt0) & 1;
      const bool more = kt + 1 < nk;
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        [[maybe_unused]] const unsigned base = lds0 + cur * STEP_BYTES + half * (32 * 256);
        TN_STAMP0();
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          WS_TR_READ(bl[t], base + offB[0][t]);
          WS_TR_READ(bh[t], base + offB[1][t]);
        }
#pragma unroll
        for (int t = 0; t < 8; ++t) {
          WS_TR_READ(al[t], base + offA[0][t]);
          WS_TR_READ(ah[t], base + offA[1][t]);
        }
        if (more) {
          stage_half(kt + 1, cur ^ 1, half);
          WS_VMCNT(4);
        } else {
          WS_VMCNT(0);
        }
        mfma_block();
      }
    }
  }
  if (wr == 0) __builtin_amdgcn_s_barrier();
  if (LEAN == 2) __builtin_amdgcn_s_barrier();  // (every wave is past its vmcnt(0))
#if defined(TN_STAMPS) && defined(__HIP_DEVICE_COMPILE__)
  if (p.dbg && lane == 0 && !sliced) {
#pragma unroll
    for (int k = 0; k < 6; ++k) atomicAdd(p.dbg + wr * 8 + k + (k >= 4 ? 1 : 0), (float)st_acc[k]);
    atomicAdd(p.dbg + wr * 8 + 4, (float)(LEAN == 2 ? (2 * (nk - kt0) + 4) / 5 * 5 : 2 * (nk - kt0)));
  }
#endif

  // ---- epilogue: acc[i][j][r] = C[i0 + wr*128 + i*16 + (lane&15)][j0 + wc*64 + j*16 + (lane>>4)*4 + r]
  const int frow = lane & 15, fq = lane >> 4;
  if (sliced) {
    // Partial sum of a K slice.  Float atomics run at full rate only on contiguous 256-byte wave accesses (one lane per
    // (row, 4-column piece), as the accumulators lie, is ~6x slower): the wavefront's 128 x 64 block goes through its own
    // 16-KB piece of the (now idle) staging buffers in two halves and is added row by row, 64 consecutive floats each.
    // 8 wavefronts x 64 rows x 64 floats = exactly the 128 KiB of staging buffers; the 16-byte pieces of a row are
    // XOR-swizzled with the row so that the 16 rows a store instruction touches fall into different banks
    constexpr int SROW = 64;
    float* stg = (float*)smem + wave * (64 * SROW);
    const int cbase = j0 + wc * 64;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          *(f32x4*)(stg + (i * 16 + frow) * SROW + (((j * 4 + fq) ^ frow) << 2)) = acc[half * 4 + i][j] * p.alpha;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      const int rbase = i0 + wr * 128 + half * 64;
      // (TAILBUF: the tile's own 256 x 256 block of the compact scratch instead of its place in C)
      float* dst = TAILBUF ? p.tail_buf + (long long)(wg - p.full_tiles) * 65536 + (long long)(wr * 128 + half * 64) * 256 + wc * 64
                           : p.C + (long long)rbase * p.ldc + cbase;
      const long long ldd = TAILBUF ? 256 : p.ldc;
      for (int row = 0; row < 64; ++row) {
        if (rbase + row >= p.NI) break;
        const float v = stg[row * SROW + ((((lane >> 2) ^ (row & 15)) << 2) | (lane & 3))];
        if (cbase + lane < p.NJ) unsafeAtomicAdd(dst + row * ldd + lane, v);
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
    return;
  }
  if constexpr (SGD) {
    // ---- the momentum-SGD update of this tile applied from the accumulators (torch.optim.SGD, dampening 0, as
    // sgd_momentum_multi_kernel spells it): g = alpha * acc * gscale; buf = mu * buf + (g + wd * p); p -= lr * buf; the
    // bf16 / bf16x2 operand copy refreshed.  Per element 8 B read + 12 / 16 B written instead of 4 B (dW) + 12 B read +
    // 12 / 16 B written by the two-kernel form.  The launcher guarantees whole quads (NJ % 8 == 0), ldc == NJ rows, no
    // K slices.  The wavefront's 128 x 64 block goes through the (idle) staging buffers so that one instruction touches
    // whole 256-byte row pieces (4 rows x 64 floats), as the sliced path's atomics do.
    constexpr int SROW = 64;
    float* stg = (float*)smem + wave * (64 * SROW);
    const float lr = p.lr_dev ? *p.lr_dev : p.lr;
    const float ag = p.alpha, gs = p.gscale, mu = p.mu, wd = p.wd;
    const int mx_exp = p.shadow_x2 == 2 ? (int)*p.mx_scale - 127 : 0;
    const float mx_iq = __builtin_ldexpf(1.0f, -mx_exp), mx_il = __builtin_ldexpf(1.0f, -(mx_exp - 11));
    const int cbase = j0 + wc * 64;
    const int rsub = lane >> 4, c4 = (lane & 15) << 2;  // lane -> (row it * 4 + rsub, 4 consecutive columns c4 ..)
#pragma unroll
    for (int half = 0; half < 2; ++half) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          *(f32x4*)(stg + (i * 16 + frow) * SROW + (((j * 4 + fq) ^ frow) << 2)) = acc[half * 4 + i][j];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      const int rbase = i0 + wr * 128 + half * 64;
      if (cbase + c4 < p.NJ) {
#pragma unroll 4
        for (int it = 0; it < 16; ++it) {
          const int row = it * 4 + rsub;
          if (rbase + row >= p.NI) break;
          f32x4 gv = *(const f32x4*)(stg + row * SROW + ((((c4 >> 2)) ^ (row & 15)) << 2));
          const long long e = (long long)(rbase + row) * p.ldc + cbase + c4;
          f32x4 pv = __builtin_nontemporal_load((const f32x4*)(p.C + e));
          f32x4 bv = __builtin_nontemporal_load((const f32x4*)(p.mom + e));
          gv = gv * ag;
          gv = gv * gs;
          bv = mu * bv + (gv + wd * pv);
          pv -= lr * bv;
          __builtin_nontemporal_store(bv, (f32x4*)(p.mom + e));
          __builtin_nontemporal_store(pv, (f32x4*)(p.C + e));
          if (p.shadow && p.shadow_x2 == 2) {
            wsovod_mx::f16x4 h4;
            int q4, l4;
            wsovod_mx::mx_enc4(pv, mx_iq, mx_il, h4, q4, l4);
            char* grp = (char*)p.shadow + ((e >> 5) << 7);
            const int w = (int)(e & 31);
            *(wsovod_mx::f16x4*)(grp + 2 * w) = h4;
            *(int*)(grp + 64 + w) = q4;
            *(int*)(grp + 96 + w) = l4;
          } else if (p.shadow) {
            const bf16x4 hi = bf16x4{(bf16_t)pv[0], (bf16_t)pv[1], (bf16_t)pv[2], (bf16_t)pv[3]};
            if (p.shadow_x2) {
              bf16_t* q = p.shadow + ((e >> 5) << 6) + (e & 31);
              *(bf16x4*)q = hi;
              *(bf16x4*)(q + 32) = bf16x4{(bf16_t)(pv[0] - (float)hi[0]), (bf16_t)(pv[1] - (float)hi[1]),
                                          (bf16_t)(pv[2] - (float)hi[2]), (bf16_t)(pv[3] - (float)hi[3])};
            } else {
              *(bf16x4*)(p.shadow + e) = hi;
            }
          }
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
    return;
  }
  const bool vec = (p.ldc & 3) == 0 && ((uintptr_t)p.C & 15) == 0 && j0 + BJ <= p.NJ;
#define WS_TN_ROW(I)                                                                                      \
  {                                                                                                       \
    const int ii = i0 + wr * 128 + (I) * 16 + frow;                                                       \
    if (ii < p.NI) {                                                                                      \
      float* crow = p.C + (long long)ii * p.ldc + j0 + wc * 64 + fq * 4;                                  \
      _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                     \
        f32x4 x = acc[I][j] * p.alpha;                                                                    \
        if (vec) {                                                                                 \
          if (p.accumulate) x += *(const f32x4*)(crow + j * 16);                                          \
          *(f32x4*)(crow + j * 16) = x;                                                                   \
        } else {                                                                                          \
          _Pragma("unroll") for (int r = 0; r < 4; ++r) if (j0 + wc * 64 + fq * 4 + j * 16 + r < p.NJ)    \
              crow[j * 16 + r] = p.accumulate ? crow[j * 16 + r] + x[r] : x[r];                           \
        }                                                                                                 \
      }                                                                                                   \
    }                                                                                                     \
  }
  WS_TN_ROW(0) WS_TN_ROW(1) WS_TN_ROW(2) WS_TN_ROW(3) WS_TN_ROW(4) WS_TN_ROW(5) WS_TN_ROW(6) WS_TN_ROW(7)
#undef WS_TN_ROW
#undef WS_TR_READ
#undef WS_LGKM0_ALL
#undef WS_VMCNT
}

// zero the output tiles [first_tile, tiles_i*tiles_j) (same tile id -> (i, j) map as the main kernel) before their
// K slices are added into them
__global__ __launch_bounds__(256) void tn_zero_tail_kernel(const TnArgs p) {
  // 16 workgroups per tile (16 rows each): one workgroup per 256-KB tile took 17 us for the 32 tail tiles of fc1's dW
  const int wg = p.full_tiles + (blockIdx.x >> 4), part = blockIdx.x & 15;
  const int group_size = p.group_m * p.tiles_j;
  const int group_id = wg / group_size;
  const int first_i = group_id * p.group_m;
  const int gm = min(p.tiles_i - first_i, p.group_m);
  const int in_group = wg - group_id * group_size;
  const int i0 = (first_i + in_group % gm) * 256 + part * 16, j0 = (in_group / gm) * 256;
  const bool vec = (p.ldc & 3) == 0 && ((uintptr_t)p.C & 15) == 0;
  for (int e = threadIdx.x; e < 16 * 64; e += 256) {
    const int i = i0 + (e >> 6), j = j0 + (e & 63) * 4;
    if (i >= p.NI) break;
    float* c = p.C + (long long)i * p.ldc + j;
    if (vec && j + 3 < p.NJ) {
      *(f32x4*)c = f32x4{0.f, 0.f, 0.f, 0.f};
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (j + r < p.NJ) c[r] = 0.f;
    }
  }
}

// SGD form: the update of the tail tiles [full_tiles, tiles) from the compact scratch their K slices were added into
// (same tile id -> (i, j) map as the main kernel; the arithmetic of the fused epilogue)
__global__ __launch_bounds__(256) void tn_sgd_tail_kernel(const TnArgs p) {
  const int wg = p.full_tiles + (blockIdx.x >> 4), part = blockIdx.x & 15;
  const int group_size = p.group_m * p.tiles_j;
  const int group_id = wg / group_size;
  const int first_i = group_id * p.group_m;
  const int gm = min(p.tiles_i - first_i, p.group_m);
  const int in_group = wg - group_id * group_size;
  const int i0 = (first_i + in_group % gm) * 256 + part * 16, j0 = (in_group / gm) * 256;
  const float* src = p.tail_buf + (long long)(wg - p.full_tiles) * 65536 + (long long)part * 16 * 256;
  const float lr = p.lr_dev ? *p.lr_dev : p.lr;
  const float gs = p.gscale, mu = p.mu, wd = p.wd;
  for (int e = threadIdx.x; e < 16 * 64; e += 256) {
    const int r = e >> 6, c4 = (e & 63) * 4;
    const int i = i0 + r, j = j0 + c4;
    if (i >= p.NI) break;
    if (j >= p.NJ) continue;  // (NJ % 8 == 0: whole quads)
    f32x4 gv = *(const f32x4*)(src + r * 256 + c4);
    const long long q = (long long)i * p.ldc + j;
    f32x4 pv = *(const f32x4*)(p.C + q);
    f32x4 bv = *(const f32x4*)(p.mom + q);
    gv = gv * gs;
    bv = mu * bv + (gv + wd * pv);
    pv -= lr * bv;
    *(f32x4*)(p.mom + q) = bv;
    *(f32x4*)(p.C + q) = pv;
    if (p.shadow && p.shadow_x2 == 2) {
      const int mx_exp = (int)*p.mx_scale - 127;
      wsovod_mx::f16x4 h4;
      int q4, l4;
      wsovod_mx::mx_enc4(pv, __builtin_ldexpf(1.0f, -mx_exp), __builtin_ldexpf(1.0f, -(mx_exp - 11)), h4, q4, l4);
      char* grp = (char*)p.shadow + ((q >> 5) << 7);
      const int w = (int)(q & 31);
      *(wsovod_mx::f16x4*)(grp + 2 * w) = h4;
      *(int*)(grp + 64 + w) = q4;
      *(int*)(grp + 96 + w) = l4;
    } else if (p.shadow) {
      const bf16x4 hi = bf16x4{(bf16_t)pv[0], (bf16_t)pv[1], (bf16_t)pv[2], (bf16_t)pv[3]};
      if (p.shadow_x2) {
        bf16_t* d = p.shadow + ((q >> 5) << 6) + (q & 31);
        *(bf16x4*)d = hi;
        *(bf16x4*)(d + 32) = bf16x4{(bf16_t)(pv[0] - (float)hi[0]), (bf16_t)(pv[1] - (float)hi[1]),
                                    (bf16_t)(pv[2] - (float)hi[2]), (bf16_t)(pv[3] - (float)hi[3])};
      } else {
        *(bf16x4*)(p.shadow + q) = hi;
      }
    }
  }
}

}  // namespace
}  // namespace wsovod_gemm

static int tn_launch(const void* P, long long ldp, const void* Q, long long ldq, int q_dtype, int Mred, int NI, int NJ,
                     float* C, long long ldc, float alpha, int accumulate, const wsovod_tn_sgd* upd, wsovod_stream_t stream);

extern "C" int wsovod_gemm_tn_ex(const void* P, long long ldp, const void* Q, long long ldq, int q_dtype, int Mred, int NI,
                                 int NJ, float* C, long long ldc, float alpha, int accumulate, wsovod_stream_t stream) {
  return tn_launch(P, ldp, Q, ldq, q_dtype, Mred, NI, NJ, C, ldc, alpha, accumulate, nullptr, stream);
}

extern "C" int wsovod_gemm_tn_sgd(const void* P, long long ldp, const void* Q, long long ldq, int q_dtype, int Mred, int NI,
                                  int NJ, float alpha, const wsovod_tn_sgd* upd, wsovod_stream_t stream) {
  WS_CHECK_ARG(upd && upd->param && upd->momentum_buf, "wsovod_gemm_tn_sgd: null parameter / momentum buffer");
  WS_CHECK_ARG((((uintptr_t)upd->param | (uintptr_t)upd->momentum_buf) & 15) == 0 && ((uintptr_t)upd->shadow & 7) == 0,
               "wsovod_gemm_tn_sgd: parameter / momentum buffer must be 16-byte aligned (shadow: 8)");
  WS_CHECK_ARG(!upd->shadow || !upd->shadow_is_bf16x2 || NJ % 32 == 0,
               "wsovod_gemm_tn_sgd: a bf16x2 / f16mx shadow needs rows of whole 32-value groups");
  WS_CHECK_ARG(!upd->shadow || upd->shadow_is_bf16x2 != 2 || (upd->mx_scale && ((uintptr_t)upd->shadow & 15) == 0),
               "wsovod_gemm_tn_sgd: an f16mx shadow needs its per-tensor scale byte and 16-byte alignment");
  return tn_launch(P, ldp, Q, ldq, q_dtype, Mred, NI, NJ, upd->param, NJ, alpha, 2, upd, stream);
}

extern "C" int wsovod_gemm_tn(const void* P, long long ldp, const void* Q, long long ldq, int Mred, int NI, int NJ,
                              float* C, long long ldc, float alpha, int accumulate, wsovod_stream_t stream) {
  return wsovod_gemm_tn_ex(P, ldp, Q, ldq, WSOVOD_BF16, Mred, NI, NJ, C, ldc, alpha, accumulate, stream);
}

static int tn_launch(const void* P, long long ldp, const void* Q, long long ldq, int q_dtype, int Mred, int NI, int NJ,
                     float* C, long long ldc, float alpha, int accumulate, const wsovod_tn_sgd* upd, wsovod_stream_t stream) {
  using namespace wsovod_gemm;
  WS_CHECK_ARG(q_dtype == WSOVOD_BF16 || q_dtype == WSOVOD_BF16X2, "wsovod_gemm_tn: Q must be bf16 or bf16x2");
  const bool q_x2 = q_dtype == WSOVOD_BF16X2;
  WS_CHECK_ARG(!q_x2 || (NJ % 32 == 0 && ldq % 4 == 0), "wsovod_gemm_tn: a bf16x2 Q needs NJ a multiple of 32");
  if (q_x2) ldq *= 2;  // bf16 slots per row
  WS_CHECK_ARG(Mred >= 0 && NI >= 0 && NJ >= 0, "wsovod_gemm_tn: negative dimension");
  if (NI == 0 || NJ == 0) return WSOVOD_OK;
  WS_CHECK_ARG(P && Q && C, "wsovod_gemm_tn: null pointer");
  WS_CHECK_ARG((((uintptr_t)P | (uintptr_t)Q) & 15) == 0, "wsovod_gemm_tn: P/Q must be 16-byte aligned");
  WS_CHECK_ARG(ldp % 8 == 0 && ldq % 8 == 0 && NI % 8 == 0 && NJ % 8 == 0,
               "wsovod_gemm_tn: row strides and column counts must be multiples of 8 bf16 elements");
  WS_CHECK_ARG((long long)Mred * ldp * 2 < (1ll << 31) && (long long)Mred * ldq * 2 < (1ll << 31),
               "wsovod_gemm_tn: operand exceeds the 2 GiB buffer-addressing limit");
  TnArgs a;
  a.P = (const char*)P;
  a.Q = (const char*)Q;
  a.ldp = ldp;
  a.ldq = ldq;
  a.Mred = Mred;
  a.NI = NI;
  a.NJ = NJ;
  a.C = C;
  a.ldc = ldc;
  a.alpha = alpha;
  a.accumulate = accumulate & 1;
  a.q_x2 = q_x2 ? 1 : 0;
  a.mom = nullptr;
  a.shadow = nullptr;
  a.shadow_x2 = 0;
  a.mx_scale = nullptr;
  a.lr = a.wd = a.mu = 0.f;
  a.gscale = 1.f;
  a.lr_dev = nullptr;
  a.tail_buf = nullptr;
  a.bid_base = 0;
  if (upd) {
    a.mom = upd->momentum_buf;
    a.shadow = (bf16_t*)upd->shadow;
    a.shadow_x2 = upd->shadow_is_bf16x2;
    a.mx_scale = upd->mx_scale;
    a.lr = upd->lr;
    a.wd = upd->weight_decay;
    a.mu = upd->momentum;
    a.gscale = upd->grad_scale;
    a.lr_dev = upd->lr_dev;
  }
#if defined(TN_STAMPS)
  a.dbg = getenv("WSOVOD_TN_DEBUG_PTR") ? (float*)strtoull(getenv("WSOVOD_TN_DEBUG_PTR"), nullptr, 16) : nullptr;
#endif
  a.tiles_i = ceil_div(NI, 256);
  a.tiles_j = ceil_div(NJ, 256);
  {
    const int run = std::max(1, a.tiles_i * a.tiles_j / 8);
    int g = 1;
    while ((g + 1) * (g + 1) <= run) ++g;
    // near-square CONCURRENT set per XCD (32 workgroups), see gemm8.hip; fc1 dW shape: group 4/8/14/16 -> 1197/1207/1154/1172
    a.group_m = std::max(1, std::min(std::min(g, 8), a.tiles_i));
  }
  static int slot_plain = wsovod::prof_slot("gemm_tn_bf16_256x256_tr");
  static int slot_sgd = wsovod::prof_slot("gemm_tn_bf16_256x256_tr_sgd");
  const int slot = upd ? slot_sgd : slot_plain;
  static bool attr_set = false;
  constexpr int lds_bytes = 2 * 2 * 2 * 64 * 256;  // 2 K-steps x (P, Q) x 2 sub-images x 64 rows x 256 B = 128 KiB
  if (!attr_set) {
    WS_CHECK_HIP(hipFuncSetAttribute((const void*)gemm_tn8_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes), "wsovod_gemm_tn: LDS opt-in");
    WS_CHECK_HIP(hipFuncSetAttribute((const void*)gemm_tn8_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes), "wsovod_gemm_tn: LDS opt-in");
    WS_CHECK_HIP(hipFuncSetAttribute((const void*)gemm_tn8_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024), "wsovod_gemm_tn: LDS opt-in (160 KiB)");
    WS_CHECK_HIP(hipFuncSetAttribute((const void*)gemm_tn8_kernel<2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024), "wsovod_gemm_tn_sgd: LDS opt-in (160 KiB)");
    WS_CHECK_HIP(hipFuncSetAttribute((const void*)gemm_tn8_kernel<2, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024), "wsovod_gemm_tn_sgd (tail): LDS opt-in (160 KiB)");
    attr_set = true;
  }
  hipStream_t s = (hipStream_t)stream;
  const double flops = 2.0 * Mred * NI * NJ;
  const double bytes = 2.0 * Mred * ((double)NI + NJ) + (upd ? (upd->shadow ? (upd->shadow_is_bf16x2 ? 20.0 : 18.0) : 16.0) : 4.0) * NI * NJ;
  wsovod::ProfScope prof(slot, s, flops, bytes);
  // Tail split: with more than one round of tiles on the 256 CUs, a last round that fills less than half of them is cut
  // along K so that it fills the chip (fc1 dW: 1568 tiles = 6 rounds + 32 tiles -> 32 x 8 slices; one round of a 7-round
  // launch becomes an eighth of a round plus the atomics).  The summation order of those tiles' slices is not fixed;
  // `accumulate` bit 1 switches the split off for a bit-reproducible result.
  const int ntiles = a.tiles_i * a.tiles_j, cus = 256;
  const int tail = ntiles % cus, nk = ceil_div(Mred, 64);
  a.full_tiles = ntiles;
  a.ksplit = a.slice_steps = 0;
  a.accumulate = accumulate & 1;
  int grid = ntiles;
  if (!(accumulate & 2) && ntiles <= cus / 2 && nk >= 32) {
    // Few output tiles (the 1024 -> 512 projection: 8; the stacked heads: 80): EVERY tile is cut along the reduction so
    // that the grid fills the chip; the slices meet by the same fp32 atomic adds as a split tail round
    const int S = std::min(std::min(8, cus / ntiles), nk / 16);
    if (S >= 2) {
      a.full_tiles = 0;
      a.ksplit = S;
      a.slice_steps = (ceil_div(nk, S) + 1) & ~1;  // even: the kernel walks K-steps in pairs
      grid = ntiles * S;
      if (!a.accumulate) hipLaunchKernelGGL(tn_zero_tail_kernel, dim3(ntiles * 16), dim3(256), 0, s, a);
    }
  } else if (!(accumulate & 2) && ntiles > cus && tail > 0 && tail <= cus / 2 &&
             !(getenv("WSOVOD_TN_TAIL") && getenv("WSOVOD_TN_TAIL")[0] == '0')) {
    const int S = std::min(8, cus / tail);
    if (nk >= 8 * S) {
      a.full_tiles = ntiles - tail;
      a.ksplit = S;
      a.slice_steps = (ceil_div(nk, S) + 1) & ~1;
      grid = a.full_tiles + tail * S;
      if (!a.accumulate) hipLaunchKernelGGL(tn_zero_tail_kernel, dim3(tail * 16), dim3(256), 0, s, a);
    }
  }
  static const int lean = getenv("WSOVOD_TN_LEAN") ? atoi(getenv("WSOVOD_TN_LEAN")) : 2;
  if (upd) {
    // The update needs the finished sum of a tile in one place: whole tiles, the fused epilogue.  WSOVOD_TN_SGD_TAIL=1
    // (measured and NOT the default): a last round that fills less than half the chip cut along K as the plain form does,
    // its slices meeting by atomics in a compact scratch and a small pass applying the update of those tiles from it --
    // fc1 (1568 tiles = 6 rounds + 32) at 1 image 0.526 against 0.496 ms for fc1 + fc2, at 8 images 0.969 against 0.944: the
    // memset + atomics + extra pass cost more than the partly filled round at these reduction lengths.
    const bool want_tail = ntiles > cus && tail > 0 && tail <= cus / 2 &&
                           getenv("WSOVOD_TN_SGD_TAIL") && getenv("WSOVOD_TN_SGD_TAIL")[0] == '1';
    const int S = want_tail ? std::min(8, cus / tail) : 0;
    if (want_tail && S >= 2 && nk >= 8 * S) {
      static float* tb = nullptr;
      static size_t tb_bytes = 0;
      static std::vector<float*> retired;  // (never freed: a captured step graph keeps the pointer it was captured with)
      const size_t need = (size_t)tail * 65536 * sizeof(float);
      if (need > tb_bytes) {
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        if (s && hipStreamIsCapturing(s, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone) {
          wsovod::set_error("wsovod_gemm_tn_sgd: the tail scratch would have to grow under stream capture; run the shape once "
                            "outside the capture first");
          return WSOVOD_ERR_UNSUPPORTED;
        }
        float* fresh = nullptr;
        const size_t want = std::max(need, std::min<size_t>(2 * tb_bytes, (size_t)128 * 65536 * sizeof(float)));
        if (hipMalloc((void**)&fresh, want) != hipSuccess) {
          wsovod::set_error("wsovod_gemm_tn_sgd: cannot allocate the tail scratch");
          return WSOVOD_ERR_HIP;
        }
        if (tb) retired.push_back(tb);
        tb = fresh;
        tb_bytes = want;
      }
      a.tail_buf = tb;
      a.full_tiles = ntiles - tail;
      a.ksplit = S;
      a.slice_steps = (ceil_div(nk, S) + 1) & ~1;
      a.bid_base = a.full_tiles;
      WS_CHECK_HIP(hipMemsetAsync(tb, 0, need, s), "wsovod_gemm_tn_sgd: tail scratch");
      hipLaunchKernelGGL((gemm_tn8_kernel<2, true>), dim3(a.full_tiles), dim3(512), 160 * 1024, s, a);
      hipLaunchKernelGGL((gemm_tn8_kernel<2, false, true>), dim3(tail * S), dim3(512), 160 * 1024, s, a);
      hipLaunchKernelGGL(tn_sgd_tail_kernel, dim3(tail * 16), dim3(256), 0, s, a);
    } else {
      hipLaunchKernelGGL((gemm_tn8_kernel<2, true>), dim3(grid), dim3(512), 160 * 1024, s, a);
    }
  } else if (lean == 2)
    hipLaunchKernelGGL(gemm_tn8_kernel<2>, dim3(grid), dim3(512), 160 * 1024, s, a);
  else if (lean == 1)
    hipLaunchKernelGGL(gemm_tn8_kernel<1>, dim3(grid), dim3(512), lds_bytes, s, a);
  else
    hipLaunchKernelGGL(gemm_tn8_kernel<0>, dim3(grid), dim3(512), lds_bytes, s, a);
  WS_CHECK_LAUNCH("wsovod_gemm_tn");
  return WSOVOD_OK;
}

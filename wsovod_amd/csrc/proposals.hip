// Proposal-side index work of the detection path: greedy non-maximum suppression over score-sorted boxes
// (the RPN's per-image NMS, proposal_utils.py:123, and the per-class NMS of the eval tail,
// fast_rcnn_open_vocabulary.py:176) and the anchor -> proposal decode of the RPN (rpn.py:495-515).
//
// Everything here is HBM / latency bound integer and compare work -- no MFMA.  Results are index sets and
// must be bit-identical to the reference's, so every float expression is written with explicit
// round-to-nearest intrinsics (no FMA contraction) in the reference's operation order.
//
// NMS is two launches over SEGMENTS (one image of the RPN batch, or one (image, class) pair of the eval tail):
//   1. nms_mask_kernel   -- thread (i, c): 64-bit word c of box i's suppression row = which of the 64 boxes
//                           [64c, 64c+64) of i's segment, ranked after i, overlap it by more than the
//                           threshold.  All lanes of a wavefront walk the same 64 candidate boxes, so the
//                           candidate loads are wave-uniform (one request, broadcast).
//   2. nms_scan_kernel   -- one wavefront per segment walks the rows in score order.  The 'removed' bitmap lives
//                           in registers (lane l owns words l, l+64, ...); rows are staged 64 at a time through
//                           LDS by a second wavefront while the current chunk is scanned; the
//                           word that decides the current chunk is mirrored in a wave-uniform scalar, so the
//                           sequential dependency per row is one scalar bit test.  Kept positions are written in
//                           order (they stay sorted by score) and the scan stops at max_keep.
#include "common.h"

// Index results (bins, keep sets, labels) must match the reference bit for bit: no mul+add fusion anywhere in this
// file (HIP's __fmul_rn & co. are plain operators and would still be contracted under the default fp-contract=fast).
#pragma clang fp contract(off)

namespace {

__device__ __forceinline__ int find_segment(const int* __restrict__ seg, int G, int i) {
  int lo = 0, hi = G;  // seg[lo] <= i < seg[hi]
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (seg[mid] <= i) lo = mid; else hi = mid;
  }
  return lo;
}

__global__ __launch_bounds__(256) void nms_mask_kernel(const float4* __restrict__ boxes, const int* __restrict__ seg,
                                                       int G, int N, int W, float thr,
                                                       unsigned long long* __restrict__ mask) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int c = blockIdx.y;
  if (i >= N) return;
  const int s = find_segment(seg, G, i);
  const int s0 = seg[s], s1 = seg[s + 1];
  const int il = i - s0;
  const int j0 = c * 64;
  unsigned long long bits = 0ull;
  if (j0 + 63 > il && s0 + j0 < s1) {
    const float4 a = boxes[i];
    const float area_a = __fmul_rn(__fsub_rn(a.z, a.x), __fsub_rn(a.w, a.y));
    const int jend = min(64, s1 - s0 - j0);
    for (int b = max(0, il + 1 - j0); b < jend; ++b) {
      const float4 q = boxes[s0 + j0 + b];
      const float area_b = __fmul_rn(__fsub_rn(q.z, q.x), __fsub_rn(q.w, q.y));
      const float w = fmaxf(0.f, __fsub_rn(fminf(a.z, q.z), fmaxf(a.x, q.x)));
      const float h = fmaxf(0.f, __fsub_rn(fminf(a.w, q.w), fmaxf(a.y, q.y)));
      const float inter = __fmul_rn(w, h);
      const float ovr = __fdiv_rn(inter, __fsub_rn(__fadd_rn(area_a, area_b), inter));
      if (ovr > thr) bits |= 1ull << b;
    }
  }
  mask[(long long)i * W + c] = bits;
}

constexpr int kScanRegs = 4;  // 'removed' words per lane -> segments of up to 64 * 64 * 4 = 16384 boxes

__global__ __launch_bounds__(128) void nms_scan_kernel(const unsigned long long* __restrict__ mask,
                                                       const int* __restrict__ seg,
                                                       const unsigned char* __restrict__ valid, int W, int max_keep,
                                                       int* __restrict__ keep_idx, int* __restrict__ keep_count) {
  // wavefront 0 scans, wavefront 1 stages the next 64-row chunk meanwhile; one barrier per chunk
  extern __shared__ unsigned long long rows[];  // 2 x [64 rows][W words]
  __shared__ int done;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int s0 = seg[blockIdx.x], n = seg[blockIdx.x + 1] - s0;
  const int Wn = (n + 63) >> 6;  // words (= chunks) this segment really uses
  auto load_chunk = [&](int t) {
    unsigned long long* dst = rows + (size_t)(t & 1) * 64 * W;
    const long long base = (long long)(s0 + t * 64) * W;
    const int live = min(64, n - t * 64) * W;
    for (int e = lane; e < live; e += 64) dst[e] = mask[base + e];
  };
  if (threadIdx.x == 0) done = 0;
  if (wave == 1 && Wn > 0) load_chunk(0);
  __syncthreads();
  unsigned long long remv[kScanRegs] = {0ull, 0ull, 0ull, 0ull};
  int count = 0;
  for (int t = 0; t < Wn; ++t) {
    if (wave == 1) {
      if (t + 1 < Wn) load_chunk(t + 1);
    } else {
      const unsigned long long* cur_rows = rows + (size_t)(t & 1) * 64 * W;
      const int rows_here = min(64, n - t * 64);
      // the word of the removed bitmap that covers this chunk, as a wave-uniform value
      unsigned long long mine = remv[0];
#pragma unroll
      for (int r = 1; r < kScanRegs; ++r)
        if ((t >> 6) == r) mine = remv[r];
      unsigned long long cur = __shfl(mine, t & 63, 64);
      if (valid) cur |= ~__ballot(lane < rows_here && valid[s0 + t * 64 + lane] != 0);
      for (int b = 0; b < rows_here; ++b) {
        if ((cur >> b) & 1ull) continue;
        if (lane == 0) keep_idx[s0 + count] = t * 64 + b;
        if (++count >= max_keep) break;
        const unsigned long long* row = cur_rows + (size_t)b * W;
        cur |= row[t];
#pragma unroll
        for (int r = 0; r < kScanRegs; ++r) {
          const int w = lane + 64 * r;
          if (w < Wn) remv[r] |= row[w];
        }
      }
      if (count >= max_keep && lane == 0) done = 1;
    }
    __syncthreads();
    if (done) break;
  }
  if (threadIdx.x == 0) keep_count[blockIdx.x] = count;
}

// Anchor -> proposal decode (Box2BoxTransform.apply_deltas, detectron2 box_regression.py; called
// from rpn.py:495-515) + clip to the image + the min-size test of find_top_rpn_proposals
// (proposal_utils.py:112-121).  One thread per selected anchor.  valid = finite && both sides > min_size.
__global__ __launch_bounds__(256) void rpn_decode_kernel(const float4* __restrict__ anchors,
                                                         const float4* __restrict__ deltas,
                                                         const long long* __restrict__ index, int per_image,
                                                         long long anchors_per_image, const float* __restrict__ sizes,
                                                         float wx, float wy, float ww, float wh, float scale_clamp,
                                                         float min_size, int total, float4* __restrict__ out,
                                                         unsigned char* __restrict__ valid) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= total) return;
  const int img = t / per_image;
  const long long a = index ? index[t] : (long long)(t - img * per_image);
  const float4 an = anchors[a];
  const float4 d = deltas[(long long)img * anchors_per_image + a];
  const float widths = __fsub_rn(an.z, an.x), heights = __fsub_rn(an.w, an.y);
  const float ctr_x = __fadd_rn(an.x, __fmul_rn(0.5f, widths)), ctr_y = __fadd_rn(an.y, __fmul_rn(0.5f, heights));
  const float dx = __fdiv_rn(d.x, wx), dy = __fdiv_rn(d.y, wy);
  const float dw = fminf(__fdiv_rn(d.z, ww), scale_clamp), dh = fminf(__fdiv_rn(d.w, wh), scale_clamp);
  const float pcx = __fadd_rn(__fmul_rn(dx, widths), ctr_x), pcy = __fadd_rn(__fmul_rn(dy, heights), ctr_y);
  const float pw = __fmul_rn(expf(dw), widths), ph = __fmul_rn(expf(dh), heights);
  float x1 = __fsub_rn(pcx, __fmul_rn(0.5f, pw)), y1 = __fsub_rn(pcy, __fmul_rn(0.5f, ph));
  float x2 = __fadd_rn(pcx, __fmul_rn(0.5f, pw)), y2 = __fadd_rn(pcy, __fmul_rn(0.5f, ph));
  const bool finite = isfinite(x1) && isfinite(y1) && isfinite(x2) && isfinite(y2);
  const float H = sizes[2 * img], Wd = sizes[2 * img + 1];
  x1 = fminf(fmaxf(x1, 0.f), Wd); x2 = fminf(fmaxf(x2, 0.f), Wd);
  y1 = fminf(fmaxf(y1, 0.f), H);  y2 = fminf(fmaxf(y2, 0.f), H);
  out[t] = make_float4(x1, y1, x2, y2);
  valid[t] = finite && __fsub_rn(x2, x1) > min_size && __fsub_rn(y2, y1) > min_size;
}

// Patch rows of the RPN's 3x3 conv for the weight gradient: only the <= BATCH_SIZE_PER_IMAGE sampled anchors of
// an image carry a loss, so dW = dH_act^T . im2col(x)[act] needs the im2col rows of those pixels only
// (rpn.py:296-375 -> label == -1 anchors contribute nothing).  One workgroup per selected output pixel, 16-byte
// chunks; taps outside the map and rows with index < 0 are zero.
template <typename T>
__global__ __launch_bounds__(256) void im2col_rows_kernel(const T* __restrict__ x, const long long* __restrict__ rows,
                                                          int H, int W, int Cin, int Ho, int Wo, int KH, int KW,
                                                          int stride, int pad, int dil, T* __restrict__ out) {
  constexpr int EPC = 16 / (int)sizeof(T);
  const long long r = rows[blockIdx.x];
  const int cpc = Cin / EPC;               // chunks per tap
  const int chunks = KH * KW * cpc;
  uint4* dst = (uint4*)(out + (long long)blockIdx.x * KH * KW * Cin);
  int img = 0, ho = 0, wo = 0;
  if (r >= 0) {
    img = (int)(r / ((long long)Ho * Wo));
    const int rem = (int)(r - (long long)img * Ho * Wo);
    ho = rem / Wo;
    wo = rem - ho * Wo;
  }
  for (int c = threadIdx.x; c < chunks; c += 256) {
    const int tap = c / cpc, cc = c - tap * cpc;
    const int kh = tap / KW, kw = tap - kh * KW;
    const int hi = ho * stride - pad + kh * dil, wi = wo * stride - pad + kw * dil;
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if (r >= 0 && hi >= 0 && hi < H && wi >= 0 && wi < W)
      v = *(const uint4*)(x + (((long long)img * H + hi) * W + wi) * Cin + cc * EPC);
    dst[c] = v;
  }
}

// Anchor labelling of the RPN (rpn.py:237-293 with detectron2's Matcher(allow_low_quality_matches=True) and
// pairwise_iou restated).  Pass 1: every (image, anchor) takes the best-overlapping pseudo-GT
// box of its image and raises that box's best-IoU cell (IoUs are >= 0, so their bit patterns order like unsigned
// ints).  Pass 2: thresholds -> {0, -1, 1}, then the low-quality rule: an anchor that attains some box's best IoU
// is positive whatever its own maximum.  Both passes evaluate the same fp32 expression, so the equality is exact.
// noinline: both passes must execute the very same instruction sequence for the equality test to be exact.
__device__ __attribute__((noinline)) float box_iou_d2(const float4 g, const float area_g, const float4 a,
                                                      const float area_a) {
  const float w = fmaxf(__fsub_rn(fminf(g.z, a.z), fmaxf(g.x, a.x)), 0.f);
  const float h = fmaxf(__fsub_rn(fminf(g.w, a.w), fmaxf(g.y, a.y)), 0.f);
  const float inter = __fmul_rn(w, h);
  return inter > 0.f ? __fdiv_rn(inter, __fsub_rn(__fadd_rn(area_g, area_a), inter)) : 0.f;
}

__global__ __launch_bounds__(256) void rpn_match_kernel(const float4* __restrict__ anchors, int A,
                                                        const float4* __restrict__ gt, const int* __restrict__ gt_start,
                                                        const int* __restrict__ gt_count,
                                                        float* __restrict__ best_val, int* __restrict__ best_gt,
                                                        unsigned int* __restrict__ gt_best) {
  const int a = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
  if (a >= A) return;
  const float4 an = anchors[a];
  const float area_a = __fmul_rn(__fsub_rn(an.z, an.x), __fsub_rn(an.w, an.y));
  const int g0 = gt_start[b], g1 = g0 + gt_count[b];
  float best = -1.f;
  int arg = -1;
  for (int g = g0; g < g1; ++g) {
    const float4 q = gt[g];
    const float iou = box_iou_d2(q, __fmul_rn(__fsub_rn(q.z, q.x), __fsub_rn(q.w, q.y)), an, area_a);
    if (iou > best) { best = iou; arg = g; }
    atomicMax(gt_best + g, __float_as_uint(iou));
  }
  best_val[(long long)b * A + a] = best;
  best_gt[(long long)b * A + a] = arg;
}

__global__ __launch_bounds__(256) void rpn_label_kernel(const float4* __restrict__ anchors, int A,
                                                        const float4* __restrict__ gt, const int* __restrict__ gt_start,
                                                        const int* __restrict__ gt_count,
                                                        const float* __restrict__ best_val,
                                                        const unsigned int* __restrict__ gt_best, float lo, float hi,
                                                        signed char* __restrict__ labels) {
  const int a = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
  if (a >= A) return;
  const int g0 = gt_start[b], g1 = g0 + gt_count[b];
  const float v = best_val[(long long)b * A + a];
  signed char lab = 0;  // no box at all: background (Matcher's empty-matrix default)
  if (g1 > g0) {
    lab = v >= hi ? 1 : (v >= lo ? -1 : 0);
    const float4 an = anchors[a];
    const float area_a = __fmul_rn(__fsub_rn(an.z, an.x), __fsub_rn(an.w, an.y));
    for (int g = g0; g < g1; ++g) {
      const float4 q = gt[g];
      const float iou = box_iou_d2(q, __fmul_rn(__fsub_rn(q.z, q.x), __fsub_rn(q.w, q.y)), an, area_a);
      if (__float_as_uint(iou) == gt_best[g]) lab = 1;
    }
  }
  labels[(long long)b * A + a] = lab;
}

}  // namespace

extern "C" {

int wsovod_rpn_label_anchors(const float* anchors, int A, const float* gt_boxes, const int* gt_start,
                             const int* gt_count, int num_images, int total_gt, float thr_lo, float thr_hi, float* best_iou, int* best_gt,
                             unsigned int* gt_best_ws, signed char* labels, wsovod_stream_t stream) {
  if (num_images == 0 || A == 0) return WSOVOD_OK;
  WS_CHECK_ARG(anchors && gt_start && gt_count && best_iou && best_gt && labels,
               "wsovod_rpn_label_anchors: null pointer");
  WS_CHECK_ARG(total_gt == 0 || (gt_boxes && gt_best_ws), "wsovod_rpn_label_anchors: null pointer");
  WS_CHECK_ARG((((uintptr_t)anchors | (uintptr_t)gt_boxes) & 15) == 0,
               "wsovod_rpn_label_anchors: anchors/gt_boxes must be 16-byte aligned");
  static int slot = wsovod::prof_slot("rpn_label_anchors");
  hipStream_t s = (hipStream_t)stream;
  wsovod::ProfScope prof(slot, s, 0.0, (double)num_images * A * 25.0);
  if (total_gt > 0 && hipMemsetAsync(gt_best_ws, 0, sizeof(unsigned int) * total_gt, s) != hipSuccess) {
    wsovod::set_error("wsovod_rpn_label_anchors: memset failed");
    return WSOVOD_ERR_HIP;
  }
  const dim3 grid(ceil_div(A, 256), num_images);
  hipLaunchKernelGGL(rpn_match_kernel, grid, dim3(256), 0, s, (const float4*)anchors, A, (const float4*)gt_boxes,
                     gt_start, gt_count, best_iou, best_gt, gt_best_ws);
  hipLaunchKernelGGL(rpn_label_kernel, grid, dim3(256), 0, s, (const float4*)anchors, A, (const float4*)gt_boxes,
                     gt_start, gt_count, best_iou, gt_best_ws, thr_lo, thr_hi, labels);
  WS_CHECK_LAUNCH("wsovod_rpn_label_anchors");
  return WSOVOD_OK;
}

int wsovod_im2col_rows(const void* x, int dtype, const long long* rows, int n_rows, int H, int W, int Cin, int Ho, int Wo,
                       int KH, int KW, int stride, int pad, int dil, void* out, wsovod_stream_t stream) {
  if (n_rows == 0) return WSOVOD_OK;
  WS_CHECK_ARG(x && rows && out, "wsovod_im2col_rows: null pointer");
  const int esz = dtype == WSOVOD_BF16 ? 2 : 4;
  WS_CHECK_ARG(dtype == WSOVOD_BF16 || dtype == WSOVOD_F32, "wsovod_im2col_rows: bad dtype");
  WS_CHECK_ARG(Cin % (16 / esz) == 0, "wsovod_im2col_rows: Cin=%d must be a multiple of %d", Cin, 16 / esz);
  WS_CHECK_ARG((((uintptr_t)x | (uintptr_t)out) & 15) == 0, "wsovod_im2col_rows: x/out must be 16-byte aligned");
  static int slot = wsovod::prof_slot("im2col_rows");
  hipStream_t s = (hipStream_t)stream;
  wsovod::ProfScope prof(slot, s, 0.0, 2.0 * n_rows * KH * KW * Cin * esz);
  if (dtype == WSOVOD_BF16)
    hipLaunchKernelGGL(im2col_rows_kernel<bf16_t>, dim3(n_rows), dim3(256), 0, s, (const bf16_t*)x, rows, H, W, Cin, Ho,
                       Wo, KH, KW, stride, pad, dil, (bf16_t*)out);
  else
    hipLaunchKernelGGL(im2col_rows_kernel<float>, dim3(n_rows), dim3(256), 0, s, (const float*)x, rows, H, W, Cin, Ho,
                       Wo, KH, KW, stride, pad, dil, (float*)out);
  WS_CHECK_LAUNCH("wsovod_im2col_rows");
  return WSOVOD_OK;
}

int wsovod_nms_segments(const float* boxes, const int* seg_offsets, const unsigned char* valid, int G, int N,
                        int max_seg_len, float iou_threshold, int max_keep, unsigned long long* workspace,
                        int* keep_idx, int* keep_count, wsovod_stream_t stream) {
  WS_CHECK_ARG(G >= 0 && N >= 0 && max_seg_len >= 0, "wsovod_nms_segments: negative size");
  if (G == 0) return WSOVOD_OK;
  WS_CHECK_ARG(seg_offsets && keep_idx && keep_count, "wsovod_nms_segments: null pointer");
  WS_CHECK_ARG(N == 0 || (boxes && workspace), "wsovod_nms_segments: null pointer");
  WS_CHECK_ARG(((uintptr_t)boxes & 15) == 0, "wsovod_nms_segments: boxes must be 16-byte aligned");
  const int W = max(1, ceil_div(max_seg_len, 64));
  if (W > 64 * kScanRegs) {
    wsovod::set_error("wsovod_nms_segments: segments of more than %d boxes are not supported (got %d)",
                      64 * 64 * kScanRegs, max_seg_len);
    return WSOVOD_ERR_UNSUPPORTED;
  }
  const size_t lds = (size_t)2 * 64 * W * sizeof(unsigned long long);
  if (lds > 159 * 1024) {
    wsovod::set_error("wsovod_nms_segments: max_seg_len %d needs %zu B of LDS", max_seg_len, lds);
    return WSOVOD_ERR_UNSUPPORTED;
  }
  if (max_keep <= 0) max_keep = 0x7fffffff;
  hipStream_t s = (hipStream_t)stream;
  static int slot_m = wsovod::prof_slot("nms_mask"), slot_s = wsovod::prof_slot("nms_scan");
  if (N > 0) {
    wsovod::ProfScope prof(slot_m, s, 0.0, (double)N * W * 8 + (double)N * 16);
    hipLaunchKernelGGL(nms_mask_kernel, dim3(ceil_div(N, 256), W), dim3(256), 0, s, (const float4*)boxes, seg_offsets,
                       G, N, W, iou_threshold, workspace);
    WS_CHECK_LAUNCH("wsovod_nms_segments(mask)");
  }
  {
    static bool attr_set = false;
    if (!attr_set) {
      WS_CHECK_HIP(hipFuncSetAttribute((const void*)nms_scan_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024), "wsovod_nms_segments: LDS opt-in");
      attr_set = true;
    }
    wsovod::ProfScope prof(slot_s, s, 0.0, (double)N * W * 8);
    hipLaunchKernelGGL(nms_scan_kernel, dim3(G), dim3(128), lds, s, workspace, seg_offsets, valid, W, max_keep,
                       keep_idx, keep_count);
    WS_CHECK_LAUNCH("wsovod_nms_segments(scan)");
  }
  return WSOVOD_OK;
}

int wsovod_rpn_decode(const float* anchors, const float* deltas, const long long* index, int num_images, int per_image,
                      long long anchors_per_image, const float* image_sizes, const float* weights, float scale_clamp,
                      float min_size, float* boxes, unsigned char* valid, wsovod_stream_t stream) {
  const long long total = (long long)num_images * per_image;
  if (total == 0) return WSOVOD_OK;
  WS_CHECK_ARG(total < (1ll << 31), "wsovod_rpn_decode: too many boxes");
  WS_CHECK_ARG(anchors && deltas && image_sizes && weights && boxes && valid, "wsovod_rpn_decode: null pointer");
  WS_CHECK_ARG((((uintptr_t)anchors | (uintptr_t)deltas | (uintptr_t)boxes) & 15) == 0,
               "wsovod_rpn_decode: anchors/deltas/boxes must be 16-byte aligned");
  static int slot = wsovod::prof_slot("rpn_decode");
  hipStream_t s = (hipStream_t)stream;
  wsovod::ProfScope prof(slot, s, 0.0, (double)total * 57);
  hipLaunchKernelGGL(rpn_decode_kernel, dim3(ceil_div((int)total, 256)), dim3(256), 0, s, (const float4*)anchors,
                     (const float4*)deltas, index, per_image, anchors_per_image, image_sizes, weights[0], weights[1],
                     weights[2], weights[3], scale_clamp, min_size, (int)total, (float4*)boxes, valid);
  WS_CHECK_LAUNCH("wsovod_rpn_decode");
  return WSOVOD_OK;
}

}  // extern "C"

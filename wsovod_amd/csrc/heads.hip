// Proposal-concept MIL head kernels: WSDDN softmax product over ragged per-image segments,
// image-level BCE, weighted softmax cross-entropy, weighted smooth-L1 box loss, and the
// no-grad pseudo-ground-truth mining + proposal labelling.  All are tiny HBM/latency-bound
// reductions: one workgroup per image (segment) or one wavefront per proposal row, with
// wavefront shuffles for the reductions; no host synchronisation anywhere.
#include <float.h>

#include "common.h"

// Index results (bins, keep sets, labels) must match the reference bit for bit: no mul+add fusion anywhere in this
// file (HIP's __fmul_rn & co. are plain operators and would still be contracted under the default fp-contract=fast).
#pragma clang fp contract(off)

namespace {

__device__ __forceinline__ float block_reduce(float v, float* sh, bool is_max) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  v = is_max ? wave_reduce_max(v) : wave_reduce_sum(v);
  __syncthreads();
  if (lane == 0) sh[wave] = v;
  __syncthreads();
  float r = sh[0];
  for (int i = 1; i < nw; ++i) r = is_max ? fmaxf(r, sh[i]) : r + sh[i];
  return r;
}

// ---------------------------------------------------------------------------------
// MIL forward (fast_rcnn_open_vocabulary.py:342-354):  per image,
//   P = softmax_k(C[r,:]),  Q = softmax_r(D[:,k]) over the image's proposals,  S = P * Q.
// logits: (M, 2K) row-major [cls | det].
//   rows kernel : wavefront per proposal row (grid-stride)            -> P
//   cols kernel : workgroup (16 wavefronts) per (image, 64-class tile): lane = class (coalesced),
//                 the wavefronts split the image's proposals, LDS combine   -> Q, S = P*Q
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void mil_rows_kernel(const float* __restrict__ logits, long long ld, int M, int K,
                                                       float* __restrict__ P) {
  const int lane = threadIdx.x & 63;
  for (int m = blockIdx.x * 4 + (threadIdx.x >> 6); m < M; m += gridDim.x * 4) {
    const float* c = logits + (long long)m * ld;
    float mx = -FLT_MAX;
    for (int k = lane; k < K; k += 64) mx = fmaxf(mx, c[k]);
    mx = wave_reduce_max(mx);
    float sum = 0.f;
    for (int k = lane; k < K; k += 64) sum += expf(c[k] - mx);
    sum = wave_reduce_sum(sum);
    for (int k = lane; k < K; k += 64) P[(long long)m * K + k] = expf(c[k] - mx) / sum;
  }
}

constexpr int kColWaves = 16;

__device__ __forceinline__ float col_combine(float v, float (*red)[64], bool is_max) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  red[wave][lane] = v;
  __syncthreads();
  float r = red[0][lane];
#pragma unroll
  for (int i = 1; i < kColWaves; ++i) r = is_max ? fmaxf(r, red[i][lane]) : r + red[i][lane];
  return r;
}

__global__ __launch_bounds__(1024) void mil_cols_kernel(const float* __restrict__ logits, long long ld,
                                                        const int* __restrict__ seg, int K,
                                                        const float* __restrict__ P, float* __restrict__ Q,
                                                        float* __restrict__ scores) {
  __shared__ float red[kColWaves][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = blockIdx.y, k = blockIdx.x * 64 + lane;
  const int m0 = seg[g], m1 = seg[g + 1];
  const bool ok = k < K;
  const float* d = logits + K + (ok ? k : 0);
  float mx = -FLT_MAX;
  for (int m = m0 + wave; m < m1; m += kColWaves) mx = fmaxf(mx, d[(long long)m * ld]);
  mx = col_combine(mx, red, true);
  float sum = 0.f;
  for (int m = m0 + wave; m < m1; m += kColWaves) sum += expf(d[(long long)m * ld] - mx);
  sum = col_combine(sum, red, false);
  if (!ok) return;
  for (int m = m0 + wave; m < m1; m += kColWaves) {
    const float q = expf(d[(long long)m * ld] - mx) / sum;
    const long long i = (long long)m * K + k;
    Q[i] = q;
    scores[i] = P[i] * q;
  }
}

// MIL backward: dC = P*(dP - sum_k dP*P), dD = Q*(dQ - sum_r dQ*Q) with dP = dS*Q, dQ = dS*P.
__global__ __launch_bounds__(256) void mil_bwd_rows_kernel(const float* __restrict__ dS, const float* __restrict__ P,
                                                           const float* __restrict__ Q, int M, int K,
                                                           float* __restrict__ dlogits, long long ld) {
  const int lane = threadIdx.x & 63;
  for (int m = blockIdx.x * 4 + (threadIdx.x >> 6); m < M; m += gridDim.x * 4) {
    const long long b = (long long)m * K;
    float dot = 0.f;
    for (int k = lane; k < K; k += 64) dot += dS[b + k] * Q[b + k] * P[b + k];
    dot = wave_reduce_sum(dot);
    for (int k = lane; k < K; k += 64) dlogits[(long long)m * ld + k] = P[b + k] * (dS[b + k] * Q[b + k] - dot);
  }
}

__global__ __launch_bounds__(1024) void mil_bwd_cols_kernel(const float* __restrict__ dS, const float* __restrict__ P,
                                                            const float* __restrict__ Q,
                                                            const int* __restrict__ seg, int K,
                                                            float* __restrict__ dlogits, long long ld) {
  __shared__ float red[kColWaves][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = blockIdx.y, k = blockIdx.x * 64 + lane;
  const int m0 = seg[g], m1 = seg[g + 1];
  const bool ok = k < K;
  float dot = 0.f;
  if (ok)
    for (int m = m0 + wave; m < m1; m += kColWaves) {
      const long long i = (long long)m * K + k;
      dot += dS[i] * P[i] * Q[i];
    }
  dot = col_combine(dot, red, false);
  if (!ok) return;
  for (int m = m0 + wave; m < m1; m += kColWaves) {
    const long long i = (long long)m * K + k;
    dlogits[(long long)m * ld + K + k] = Q[i] * (dS[i] * P[i] - dot);
  }
}

// ---------------------------------------------------------------------------------
// Image-level scores + BCE (predict_probs_img :604-618, binary_cross_entropy :429-437):
//   S[g][k] = sum_r scores;  c = clamp(S, 1e-6, 1-1e-6);  loss = sum BCE(c, y) / norm.
//   sum kernel: workgroup (16 wavefronts) per (image, 64-class tile) -> S;  finalize: one workgroup.
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void image_sum_kernel(const float* __restrict__ scores,
                                                         const int* __restrict__ seg, int K, float* __restrict__ S) {
  __shared__ float red[kColWaves][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = blockIdx.y, k = blockIdx.x * 64 + lane;
  float s = 0.f;
  if (k < K)
    for (int m = seg[g] + wave; m < seg[g + 1]; m += kColWaves) s += scores[(long long)m * K + k];
  s = col_combine(s, red, false);
  if (k < K && wave == 0) S[(long long)g * K + k] = s;
}

__global__ __launch_bounds__(256) void image_bce_kernel(const float* S, int GK, const float* __restrict__ y,
                                                        float norm, float* __restrict__ img_scores, float* dS_img,
                                                        float* __restrict__ loss) {  // S may alias dS_img
  __shared__ float sh[4];
  float local = 0.f;
  for (int i = threadIdx.x; i < GK; i += blockDim.x) {
    const float s = S[i];
    const float c = fminf(fmaxf(s, 1e-6f), 1.0f - 1e-6f);
    img_scores[i] = c;
    const float yy = y[i];
    local += -(yy * fmaxf(logf(c), -100.f) + (1.f - yy) * fmaxf(logf(1.f - c), -100.f));
    const bool pass = s >= 1e-6f && s <= 1.0f - 1e-6f;  // clamp passes gradient inside [min,max]
    dS_img[i] = pass ? (-(yy / c) + (1.f - yy) / (1.f - c)) / norm : 0.f;
  }
  const float tot = block_reduce(local, sh, false);
  if (threadIdx.x == 0) loss[0] = tot / norm;
}

// dscores[m][k] = gout[0] * dS_img[g(m)][k]
__global__ void image_bce_backward_kernel(const float* __restrict__ dS_img, const int* __restrict__ seg, int K,
                                          const float* __restrict__ gout, float* __restrict__ dscores) {
  const int g = blockIdx.x;
  const float go = gout ? gout[0] : 1.f;
  for (long long i = (long long)seg[g] * K + threadIdx.x + (long long)blockIdx.y * blockDim.x;
       i < (long long)seg[g + 1] * K; i += (long long)blockDim.x * gridDim.y)
    dscores[i] = go * dS_img[(long long)g * K + (int)(i % K)];
}

// ---------------------------------------------------------------------------------
// Weighted softmax cross-entropy (fast_rcnn_open_vocabulary.py:799-802,813-820):
//   w'_r = (gt_r == -1) ? 0 : w_r;  loss = sum_r w'_r * CE_r / #{w'_r > 1e-12};  ignore_index -1.
// unweighted mode: mean over non-ignored rows.  Wavefront per row; accum[0]=sum, accum[1]=count
// (zeroed by the caller).  dlogits holds the UN-normalised gradient; backward scales by
// gout/count on the device.
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void weighted_ce_kernel(const float* __restrict__ logits, long long ld, int M,
                                                          int K1, const long long* __restrict__ gt,
                                                          const float* __restrict__ w, int weighted,
                                                          float* __restrict__ dlogits, long long ldd,
                                                          float* __restrict__ accum) {
  __shared__ float sh_sum[4], sh_cnt[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float lsum = 0.f, lcnt = 0.f;
  for (int m = blockIdx.x * 4 + wave; m < M; m += gridDim.x * 4) {
    const long long t = gt[m];
    const bool ignore = t == -1;
    const float wr = weighted ? (ignore ? 0.f : w[m]) : (ignore ? 0.f : 1.f);
    const float* x = logits + (long long)m * ld;
    float mx = -FLT_MAX;
    for (int k = lane; k < K1; k += 64) mx = fmaxf(mx, x[k]);
    mx = wave_reduce_max(mx);
    float sum = 0.f;
    for (int k = lane; k < K1; k += 64) sum += expf(x[k] - mx);
    sum = wave_reduce_sum(sum);
    const float lse = mx + logf(sum);
    for (int k = lane; k < K1; k += 64) {
      const float p = expf(x[k] - lse);
      dlogits[(long long)m * ldd + k] = ignore ? 0.f : wr * (p - (k == t ? 1.f : 0.f));
    }
    if (!ignore) lsum += wr * (lse - x[t]);
    if (weighted ? (wr > 1e-12f) : !ignore) lcnt += 1.0f;
  }
  if (lane == 0) {
    sh_sum[wave] = lsum;
    sh_cnt[wave] = lcnt;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    atomicAdd(accum, sh_sum[0] + sh_sum[1] + sh_sum[2] + sh_sum[3]);
    atomicAdd(accum + 1, sh_cnt[0] + sh_cnt[1] + sh_cnt[2] + sh_cnt[3]);
  }
}
__global__ void ce_finalize_kernel(const float* __restrict__ accum, float* __restrict__ loss) {
  loss[0] = accum[0] / accum[1];
}

// ---------------------------------------------------------------------------------
// Weighted smooth-L1 box regression (fast_rcnn_open_vocabulary.py:822-892, branch :864-878),
// class-agnostic deltas (num_bbox_reg_classes = 1):  fg = 0 <= gt < K;
//   t = Box2BoxTransform(weights).get_deltas(proposal, gt_box);  loss = sum_fg sl1(p - t) * w / max(M,1)
// A NaN in the target deltas zeroes the whole loss (the reference's guard :868-871).
// accum[0] = sum, accum[1] = NaN flag (zeroed by caller).  dpred is un-normalised (divide by M later).
// ---------------------------------------------------------------------------------
__global__ void weighted_l1_kernel(const float* __restrict__ pred, long long ldp, const float* __restrict__ pbox,
                                   const float* __restrict__ gbox, const long long* __restrict__ gt,
                                   const float* __restrict__ w, int M, int K, float wx, float wy, float ww, float wh,
                                   float beta, int weighted, float* __restrict__ dpred, float* __restrict__ accum) {
  float tot = 0.f, bad = 0.f;
  for (int m = blockIdx.x * blockDim.x + threadIdx.x; m < M; m += gridDim.x * blockDim.x) {
    const long long t = gt[m];
    const bool fg = t >= 0 && t < K;
    float local = 0.f;
    float d[4] = {0.f, 0.f, 0.f, 0.f};
    if (fg) {
      const float* s = pbox + (long long)m * 4;
      const float* g = gbox + (long long)m * 4;
      const float sw = s[2] - s[0], sh = s[3] - s[1];
      const float sx = s[0] + 0.5f * sw, sy = s[1] + 0.5f * sh;
      const float tw = g[2] - g[0], th = g[3] - g[1];
      const float tx = g[0] + 0.5f * tw, ty = g[1] + 0.5f * th;
      const float tgt[4] = {wx * (tx - sx) / sw, wy * (ty - sy) / sh, ww * logf(tw / sw), wh * logf(th / sh)};
      const float wr = weighted ? w[m] : 1.f;
      bool nan = false;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        nan |= tgt[j] != tgt[j];
        const float diff = pred[(long long)m * ldp + j] - tgt[j];
        const float a = fabsf(diff);
        if (beta < 1e-5f) {
          local += a * wr;
          d[j] = (diff > 0.f ? 1.f : (diff < 0.f ? -1.f : 0.f)) * wr;
        } else if (a < beta) {
          local += 0.5f * diff * diff / beta * wr;
          d[j] = diff / beta * wr;
        } else {
          local += (a - 0.5f * beta) * wr;
          d[j] = (diff > 0.f ? 1.f : -1.f) * wr;
        }
      }
      if (nan) {
        bad += 1.0f;
        local = 0.f;
      }
      tot += local;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) dpred[(long long)m * 4 + j] = d[j];
  }
  __shared__ float sh[4];
  const float t = block_reduce(tot, sh, false);
  const float b = block_reduce(bad, sh, false);
  if (threadIdx.x == 0) {
    if (t != 0.f) atomicAdd(accum, t);
    if (b != 0.f) atomicAdd(accum + 1, b);
  }
}
__global__ void l1_finalize_kernel(const float* __restrict__ accum, int M, const int* __restrict__ rows_true,
                                   float* __restrict__ loss, float* __restrict__ dpred) {
  const bool bad = accum[1] > 0.f;
  const float inv = 1.0f / (float)max(rows_true ? *rows_true : M, 1);
  if (blockIdx.x == 0 && threadIdx.x == 0) loss[0] = bad ? 0.f : accum[0] * inv;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < (long long)M * 4;
       i += (long long)gridDim.x * blockDim.x)
    dpred[i] = bad ? 0.f : dpred[i] * inv;
}

// ---------------------------------------------------------------------------------
// Pseudo-GT mining + labelling (no grad): get_pgt_top_k (roi_heads.py:1043-1343, top_k=1,
// sam=None) + label_and_sample_proposals_wsl (roi_heads.py:1722-1825) with Matcher([thr],[0,1])
// and subsample_labels keeping everything (R <= BATCH_SIZE_PER_IMAGE, POSITIVE_FRACTION 1.0).
// Workgroup per image.
//   step 1: for every image-level GT class: arg-max of its score column over proposals whose
//           box area > 20 (first maximum); weight = clamped image score of that class.
//   step 2: every proposal: IoU against the mined boxes, first max; label = class if IoU >= thr
//           else K (background); carries the matched box / score / weight.
// ---------------------------------------------------------------------------------
constexpr int kMaxPgt = 128;  // max distinct GT classes per image

__global__ __launch_bounds__(256) void pgt_mine_label_kernel(
    const float* __restrict__ scores, long long lds_, const float* __restrict__ boxes,
    const int* __restrict__ seg, const long long* __restrict__ gt_cls, const int* __restrict__ gt_off,
    const float* __restrict__ img_scores, int K, float iou_thr, float* __restrict__ pgt_boxes,
    long long* __restrict__ pgt_classes, float* __restrict__ pgt_scores, float* __restrict__ pgt_weights,
    int* __restrict__ pgt_index, int* __restrict__ pgt_count, long long* __restrict__ out_classes,
    float* __restrict__ out_boxes, float* __restrict__ out_scores, float* __restrict__ out_weights,
    int* __restrict__ out_matched) {
  __shared__ float s_box[kMaxPgt][4];
  __shared__ float s_w[kMaxPgt], s_sc[kMaxPgt];
  __shared__ long long s_cls[kMaxPgt];
  __shared__ float sh_v[4];
  __shared__ int sh_i[4];
  __shared__ int s_cnt;
  const int g = blockIdx.x;
  const int m0 = seg[g], m1 = seg[g + 1];
  const int c0 = gt_off[g], c1 = gt_off[g + 1];
  const int G = min(c1 - c0, kMaxPgt);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  bool any_kept = false;
  for (int j = 0; j < G; ++j) {
    const long long cls = gt_cls[c0 + j];
    float best = -FLT_MAX;
    int bi = 0x7fffffff;
    for (int m = m0 + threadIdx.x; m < m1; m += blockDim.x) {
      const float* b = boxes + (long long)m * 4;
      const float area = (b[2] - b[0]) * (b[3] - b[1]);
      if (!(area > 20.f)) continue;
      const float v = scores[(long long)m * lds_ + cls];
      if (v > best) {  // ascending m per thread: keeps this thread's first maximum
        best = v;
        bi = m;
      }
    }
    // wavefront then workgroup arg-max with "smaller index wins ties"
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(best, o, 64);
      const int oi = __shfl_xor(bi, o, 64);
      if (ov > best || (ov == best && oi < bi)) {
        best = ov;
        bi = oi;
      }
    }
    __syncthreads();
    if (lane == 0) {
      sh_v[wave] = best;
      sh_i[wave] = bi;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      float bv = sh_v[0];
      int bidx = sh_i[0];
      for (int wv = 1; wv < 4; ++wv)
        if (sh_v[wv] > bv || (sh_v[wv] == bv && sh_i[wv] < bidx)) {
          bv = sh_v[wv];
          bidx = sh_i[wv];
        }
      const bool found = bidx != 0x7fffffff;
      if (found) {
        for (int q = 0; q < 4; ++q) s_box[j][q] = boxes[(long long)bidx * 4 + q];
        s_sc[j] = bv;
        s_w[j] = img_scores[(long long)g * K + cls];
        s_cls[j] = cls;
        sh_i[0] = bidx - m0;
      } else {
        sh_i[0] = -1;
      }
      pgt_index[c0 + j] = sh_i[0];
    }
    __syncthreads();
    any_kept |= sh_i[0] >= 0;
    __syncthreads();
  }
  // Either every GT class found a box (the area filter is class-independent) or none did:
  // the reference then falls back to one dummy target (roi_heads.py:1181-1207).
  if (threadIdx.x == 0) {
    int cnt = G;
    if (!any_kept) {
      cnt = 1;
      s_box[0][0] = -10000.f; s_box[0][1] = -10000.f; s_box[0][2] = 10000.f; s_box[0][3] = 10000.f;
      s_sc[0] = 1.f;
      s_w[0] = 1.f;
      s_cls[0] = 0;
    }
    s_cnt = cnt;
    pgt_count[g] = cnt;
  }
  __syncthreads();
  const int cnt = s_cnt;
  // publish the mined targets; a fallback target with no GT slot lives only in LDS
  for (int j = threadIdx.x; j < cnt && c0 + j < c1; j += blockDim.x) {
    for (int q = 0; q < 4; ++q) pgt_boxes[(long long)(c0 + j) * 4 + q] = s_box[j][q];
    pgt_classes[c0 + j] = s_cls[j];
    pgt_scores[c0 + j] = s_sc[j];
    pgt_weights[c0 + j] = s_w[j];
  }
  for (int m = m0 + threadIdx.x; m < m1; m += blockDim.x) {
    const float* b = boxes + (long long)m * 4;
    const float a1 = (b[2] - b[0]) * (b[3] - b[1]);
    float best = -1.f;
    int bj = 0;
    for (int j = 0; j < cnt; ++j) {
      const float iw = fminf(b[2], s_box[j][2]) - fmaxf(b[0], s_box[j][0]);
      const float ih = fminf(b[3], s_box[j][3]) - fmaxf(b[1], s_box[j][1]);
      const float inter = fmaxf(iw, 0.f) * fmaxf(ih, 0.f);
      const float a2 = (s_box[j][2] - s_box[j][0]) * (s_box[j][3] - s_box[j][1]);
      // detectron2 pairwise_iou(gt, proposals): inter / (area_gt + area_prop - inter)
      const float iou = inter > 0.f ? inter / (a2 + a1 - inter) : 0.f;
      if (iou > best) {
        best = iou;
        bj = j;
      }
    }
    out_classes[m] = best >= iou_thr ? s_cls[bj] : (long long)K;
    for (int q = 0; q < 4; ++q) out_boxes[(long long)m * 4 + q] = s_box[bj][q];
    out_scores[m] = s_sc[bj];
    out_weights[m] = s_w[bj];
    out_matched[m] = bj;
  }
}

// ---------------------------------------------------------------------------------
// DataAwareFeaturesHead (class_heads/data_aware_features_head.py:103-129) on the pooled
// (N,C) GAP vector:  h1 = relu(W1 g + b1) (Hd = C/16);  h2 = tanh(W2 h1 + b2) (P prototypes);
// daf = h2 @ E (P x F).  Workgroup per image forward; the backward (sums over images) is one
// workgroup -- the whole head is a few hundred KFLOP.
// ---------------------------------------------------------------------------------
// first layer: one wavefront per (image, hidden unit) -- N*Hd dot products of length C spread over the chip (a
// workgroup per image walked its Hd/4 rows one after the other: 0.37 ms of pure latency at C = 2048)
__global__ __launch_bounds__(256) void data_aware_h1_kernel(const float* __restrict__ gap, int Cc,
                                                            const float* __restrict__ W1,
                                                            const float* __restrict__ b1, int Hd, int N,
                                                            float* __restrict__ h1) {
  const int lane = threadIdx.x & 63;
  const long long item = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (item >= (long long)N * Hd) return;
  const int n = (int)(item / Hd), j = (int)(item - (long long)n * Hd);
  const float* g = gap + (long long)n * Cc;
  float a = 0.f;
  for (int c = lane; c < Cc; c += 64) a += W1[(long long)j * Cc + c] * g[c];
  a = wave_reduce_sum(a);
  if (lane == 0) h1[item] = fmaxf(a + b1[j], 0.f);
}

__global__ __launch_bounds__(256) void data_aware_fwd_kernel(const float* __restrict__ W2,
                                                             const float* __restrict__ b2, int P,
                                                             const float* __restrict__ E, int F,
                                                             const float* __restrict__ h1, int Hd,
                                                             float* __restrict__ h2, float* __restrict__ daf) {
  extern __shared__ float sm[];
  float* s1 = sm;        // Hd
  float* s2 = sm + Hd;   // P
  const int n = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int j = threadIdx.x; j < Hd; j += blockDim.x) s1[j] = h1[(long long)n * Hd + j];
  __syncthreads();
  for (int p = wave; p < P; p += 4) {
    float a = 0.f;
    for (int j = lane; j < Hd; j += 64) a += W2[(long long)p * Hd + j] * s1[j];
    a = wave_reduce_sum(a);
    if (lane == 0) {
      a = tanhf(a + b2[p]);
      s2[p] = a;
      h2[(long long)n * P + p] = a;
    }
  }
  __syncthreads();
  for (int f = threadIdx.x; f < F; f += blockDim.x) {
    float a = 0.f;
    for (int p = 0; p < P; ++p) a += s2[p] * E[(long long)p * F + f];
    daf[(long long)n * F + f] = a;
  }
}

// stage 1, workgroup per image: dpre2[n][p] = (ddaf[n] . E[p]) * (1 - h2^2);  dh1[n][j] = relu'(h1) * W2^T dpre2
__global__ __launch_bounds__(256) void data_aware_bwd_stage1(const float* __restrict__ ddaf,
                                                             const float* __restrict__ W2,
                                                             const float* __restrict__ E, int F,
                                                             const float* __restrict__ h1, int Hd,
                                                             const float* __restrict__ h2, int P,
                                                             float* __restrict__ dpre2, float* __restrict__ dh1) {
  extern __shared__ float sm[];  // P
  __shared__ float sh[4];
  const int n = blockIdx.x, tid = threadIdx.x;
  const float* dd = ddaf + (long long)n * F;
  for (int p = 0; p < P; ++p) {
    float a = 0.f;
    for (int f = tid; f < F; f += blockDim.x) a += dd[f] * E[(long long)p * F + f];
    a = block_reduce(a, sh, false);
    if (tid == 0) {
      const float t = h2[(long long)n * P + p];
      sm[p] = a * (1.f - t * t);
      dpre2[(long long)n * P + p] = sm[p];
    }
  }
  __syncthreads();
  for (int j = tid; j < Hd; j += blockDim.x) {
    float a = 0.f;
    for (int p = 0; p < P; ++p) a += W2[(long long)p * Hd + j] * sm[p];
    dh1[(long long)n * Hd + j] = h1[(long long)n * Hd + j] > 0.f ? a : 0.f;
  }
}

// stage 2, grid-stride over every gradient element; each sums over the N images in a fixed order
// (deterministic): dE = h2^T ddaf, dW2 = dpre2^T h1, db2, dW1 = dh1^T gap, db1.
__global__ void data_aware_bwd_stage2(const float* __restrict__ ddaf, int N, const float* __restrict__ gap, int Cc,
                                      int F, const float* __restrict__ h1, int Hd, const float* __restrict__ h2,
                                      int P, const float* __restrict__ dpre2, const float* __restrict__ dh1,
                                      float* __restrict__ dW1, float* __restrict__ db1, float* __restrict__ dW2,
                                      float* __restrict__ db2, float* __restrict__ dE) {
  const long long nE = (long long)P * F, nW2 = (long long)P * Hd, nW1 = (long long)Hd * Cc;
  const long long total = nE + nW2 + P + nW1 + Hd;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    float a = 0.f;
    long long j = i;
    if (j < nE) {
      const int p = (int)(j / F), f = (int)(j % F);
      for (int n = 0; n < N; ++n) a += h2[(long long)n * P + p] * ddaf[(long long)n * F + f];
      dE[j] = a;
    } else if ((j -= nE) < nW2) {
      const int p = (int)(j / Hd), q = (int)(j % Hd);
      for (int n = 0; n < N; ++n) a += dpre2[(long long)n * P + p] * h1[(long long)n * Hd + q];
      dW2[j] = a;
    } else if ((j -= nW2) < P) {
      for (int n = 0; n < N; ++n) a += dpre2[(long long)n * P + j];
      db2[j] = a;
    } else if ((j -= P) < nW1) {
      const int q = (int)(j / Cc), c = (int)(j % Cc);
      for (int n = 0; n < N; ++n) a += dh1[(long long)n * Hd + q] * gap[(long long)n * Cc + c];
      dW1[j] = a;
    } else {
      j -= nW1;
      for (int n = 0; n < N; ++n) a += dh1[(long long)n * Hd + j];
      db1[j] = a;
    }
  }
}


// ---------------------------------------------------------------------------------
// Proposal sub-sampling (no grad): _sample_proposals_wsl (roi_heads.py:1566-1603) ->
// detectron2 subsample_labels(labels, num, positive_fraction, bg_label):
//   positives = labels not in {-1, bg};  negatives = labels == bg
//   num_pos = min(#pos, pos_cap)  (pos_cap = int(num * positive_fraction), computed by the host)
//   num_neg = min(#neg, num - num_pos)
//   a uniform sample without replacement of each group keeps its label; every other row becomes -1 (ignored).
// The random permutation is expressed through per-row sort keys: a row is sampled when its rank among
// the rows of its group, ordered by (key, row index), is below the group's quota -- uniform keys give the
// reference's `randperm(...)[:n]`, keys = row index give a deterministic first-n rule (golden fixtures).
// Workgroup per 256 rows of one image; the image's (key, group) pairs stream through LDS.  Index work only:
// bit-exact against the oracle on the same keys.
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void subsample_labels_kernel(const long long* __restrict__ labels,
                                                               const float* __restrict__ keys,
                                                               const int* __restrict__ seg, int num, int pos_cap,
                                                               long long bg, long long* __restrict__ out) {
  __shared__ float sh_key[256];
  __shared__ int sh_grp[256];
  const int g = blockIdx.y;
  const int r0 = seg[g], n = seg[g + 1] - r0;
  if ((int)blockIdx.x * 256 >= n) return;
  const int r = blockIdx.x * 256 + threadIdx.x;
  const bool live = r < n;
  const long long lab = live ? labels[r0 + r] : -1;
  const float key = live ? keys[r0 + r] : 0.f;
  const int grp = lab == bg ? 1 : (lab == -1 ? 2 : 0);
  int rank = 0, n_pos = 0, n_neg = 0;
  for (int t0 = 0; t0 < n; t0 += 256) {
    const int j = t0 + threadIdx.x;
    __syncthreads();
    if (j < n) {
      const long long lj = labels[r0 + j];
      sh_key[threadIdx.x] = keys[r0 + j];
      sh_grp[threadIdx.x] = lj == bg ? 1 : (lj == -1 ? 2 : 0);
    } else {
      sh_grp[threadIdx.x] = 2;
      sh_key[threadIdx.x] = 0.f;
    }
    __syncthreads();
    const int lim = min(256, n - t0);
    for (int q = 0; q < lim; ++q) {
      const int gq = sh_grp[q];
      const float kq = sh_key[q];
      n_pos += gq == 0;
      n_neg += gq == 1;
      rank += (gq == grp) && (kq < key || (kq == key && t0 + q < r));
    }
  }
  if (!live) return;
  const int num_pos = min(n_pos, pos_cap);
  const int num_neg = min(n_neg, num - num_pos);
  const bool keep = (grp == 0 && rank < num_pos) || (grp == 1 && rank < num_neg);
  out[r0 + r] = keep ? lab : -1;
}
}  // namespace

extern "C" {

int wsovod_mil_forward(const float* logits, long long ld, const int* seg_offsets, int G, int K, float* scores,
                       float* P, float* Q, int M, wsovod_stream_t stream) {
  if (G == 0 || M == 0) return WSOVOD_OK;
  WS_CHECK_ARG(logits && seg_offsets && scores && P && Q && K > 0, "wsovod_mil_forward: bad argument");
  static int slot = wsovod::prof_slot("mil_forward");
  hipStream_t s = (hipStream_t)stream;
  wsovod::ProfScope prof(slot, s, 0.0, (double)M * K * 20.0);
  hipLaunchKernelGGL(mil_rows_kernel, dim3(std::min(ceil_div(M, 4), 2048)), dim3(256), 0, s, logits, ld, M, K, P);
  hipLaunchKernelGGL(mil_cols_kernel, dim3(ceil_div(K, 64), G), dim3(1024), 0, s, logits, ld, seg_offsets, K, P, Q,
                     scores);
  WS_CHECK_LAUNCH("wsovod_mil_forward");
  return WSOVOD_OK;
}

int wsovod_mil_backward(const float* dscores, const float* P, const float* Q, const int* seg_offsets, int G, int K,
                        float* dlogits, long long ld, int M, wsovod_stream_t stream) {
  if (G == 0 || M == 0) return WSOVOD_OK;
  WS_CHECK_ARG(dscores && P && Q && seg_offsets && dlogits && K > 0, "wsovod_mil_backward: bad argument");
  static int slot = wsovod::prof_slot("mil_backward");
  hipStream_t s = (hipStream_t)stream;
  wsovod::ProfScope prof(slot, s, 0.0, (double)M * K * 20.0);
  hipLaunchKernelGGL(mil_bwd_rows_kernel, dim3(std::min(ceil_div(M, 4), 2048)), dim3(256), 0, s, dscores, P, Q, M, K,
                     dlogits, ld);
  hipLaunchKernelGGL(mil_bwd_cols_kernel, dim3(ceil_div(K, 64), G), dim3(1024), 0, s, dscores, P, Q, seg_offsets, K,
                     dlogits, ld);
  WS_CHECK_LAUNCH("wsovod_mil_backward");
  return WSOVOD_OK;
}

int wsovod_image_bce_forward(const float* scores, const int* seg_offsets, int G, int K, const float* labels_onehot,
                             float norm, float* img_scores, float* dS_img, float* loss, wsovod_stream_t stream) {
  WS_CHECK_ARG(scores && seg_offsets && labels_onehot && img_scores && dS_img && loss && G > 0 && K > 0,
               "wsovod_image_bce_forward: bad argument");
  static int slot = wsovod::prof_slot("image_bce");
  hipStream_t s = (hipStream_t)stream;
  wsovod::ProfScope prof(slot, s, 0.0, 0.0);
  // dS_img doubles as the scratch for the un-clamped sums (read, then overwritten, per element)
  hipLaunchKernelGGL(image_sum_kernel, dim3(ceil_div(K, 64), G), dim3(1024), 0, s, scores, seg_offsets, K, dS_img);
  hipLaunchKernelGGL(image_bce_kernel, dim3(1), dim3(256), 0, s, dS_img, G * K, labels_onehot, norm, img_scores,
                     dS_img, loss);
  WS_CHECK_LAUNCH("wsovod_image_bce_forward");
  return WSOVOD_OK;
}

int wsovod_image_bce_backward(const float* dS_img, const int* seg_offsets, int G, int K, const float* grad_out,
                              float* dscores, wsovod_stream_t stream) {
  if (G == 0) return WSOVOD_OK;
  WS_CHECK_ARG(dS_img && seg_offsets && dscores, "wsovod_image_bce_backward: bad argument");
  static int slot = wsovod::prof_slot("image_bce_bwd");
  hipStream_t s = (hipStream_t)stream;
  wsovod::ProfScope prof(slot, s, 0.0, 0.0);
  hipLaunchKernelGGL(image_bce_backward_kernel, dim3(G, 8), dim3(256), 0, s, dS_img, seg_offsets, K, grad_out, dscores);
  WS_CHECK_LAUNCH("wsovod_image_bce_backward");
  return WSOVOD_OK;
}

int wsovod_weighted_ce_forward(const float* logits, long long ld, int M, int K1, const long long* gt_classes,
                               const float* weights, int weighted, float* dlogits, long long ldd, float* accum2,
                               float* loss, wsovod_stream_t stream) {
  WS_CHECK_ARG(accum2 && loss, "wsovod_weighted_ce_forward: null pointer");
  static int slot = wsovod::prof_slot("weighted_ce");
  hipStream_t s = (hipStream_t)stream;
  wsovod::ProfScope prof(slot, s, 0.0, 0.0);
  (void)hipMemsetAsync(accum2, 0, 2 * sizeof(float), s);
  if (M > 0) {
    WS_CHECK_ARG(logits && gt_classes && dlogits && K1 > 0 && (!weighted || weights), "wsovod_weighted_ce_forward: bad argument");
    hipLaunchKernelGGL(weighted_ce_kernel, dim3(std::min(ceil_div(M, 4), 256)), dim3(256), 0, s, logits, ld, M, K1, gt_classes,
                       weights, weighted, dlogits, ldd, accum2);
  }
  hipLaunchKernelGGL(ce_finalize_kernel, dim3(1), dim3(1), 0, s, accum2, loss);
  WS_CHECK_LAUNCH("wsovod_weighted_ce_forward");
  return WSOVOD_OK;
}

int wsovod_weighted_l1_box_forward(const float* pred_deltas, long long ldp, const float* proposal_boxes,
                                   const float* gt_boxes, const long long* gt_classes, const float* weights, int M,
                                   int K, const float* bbox_weights_host, float beta, int weighted, float* dpred,
                                   float* accum2, float* loss, const int* rows_true, wsovod_stream_t stream) {
  WS_CHECK_ARG(accum2 && loss && bbox_weights_host, "wsovod_weighted_l1_box_forward: null pointer");
  static int slot = wsovod::prof_slot("weighted_l1_box");
  hipStream_t s = (hipStream_t)stream;
  wsovod::ProfScope prof(slot, s, 0.0, 0.0);
  (void)hipMemsetAsync(accum2, 0, 2 * sizeof(float), s);
  if (M > 0) {
    WS_CHECK_ARG(pred_deltas && proposal_boxes && gt_boxes && gt_classes && dpred && (!weighted || weights),
                 "wsovod_weighted_l1_box_forward: bad argument");
    hipLaunchKernelGGL(weighted_l1_kernel, dim3(ceil_div(M, 256)), dim3(256), 0, s, pred_deltas, ldp, proposal_boxes,
                       gt_boxes, gt_classes, weights, M, K, bbox_weights_host[0], bbox_weights_host[1],
                       bbox_weights_host[2], bbox_weights_host[3], beta, weighted, dpred, accum2);
  }
  hipLaunchKernelGGL(l1_finalize_kernel, dim3(std::max(1, ceil_div(M * 4, 256))), dim3(256), 0, s, accum2, M, rows_true, loss,
                     dpred);
  WS_CHECK_LAUNCH("wsovod_weighted_l1_box_forward");
  return WSOVOD_OK;
}

int wsovod_pgt_mine_and_label(const float* scores, long long ld_scores, const float* boxes, const int* seg_offsets,
                              int G, const long long* gt_classes_img, const int* gt_offsets,
                              const float* img_scores, int K, float iou_threshold, float* pgt_boxes,
                              long long* pgt_classes, float* pgt_scores, float* pgt_weights, int* pgt_index,
                              int* pgt_count, long long* out_classes, float* out_boxes, float* out_scores,
                              float* out_weights, int* out_matched, wsovod_stream_t stream) {
  if (G == 0) return WSOVOD_OK;
  WS_CHECK_ARG(scores && boxes && seg_offsets && gt_classes_img && gt_offsets && img_scores && pgt_boxes &&
                   pgt_classes && pgt_scores && pgt_weights && pgt_index && pgt_count && out_classes && out_boxes &&
                   out_scores && out_weights && out_matched,
               "wsovod_pgt_mine_and_label: null pointer");
  static int slot = wsovod::prof_slot("pgt_mine_label");
  hipStream_t s = (hipStream_t)stream;
  wsovod::ProfScope prof(slot, s, 0.0, 0.0);
  hipLaunchKernelGGL(pgt_mine_label_kernel, dim3(G), dim3(256), 0, s, scores, ld_scores, boxes, seg_offsets,
                     gt_classes_img, gt_offsets, img_scores, K, iou_threshold, pgt_boxes, pgt_classes, pgt_scores,
                     pgt_weights, pgt_index, pgt_count, out_classes, out_boxes, out_scores, out_weights, out_matched);
  WS_CHECK_LAUNCH("wsovod_pgt_mine_and_label");
  return WSOVOD_OK;
}

int wsovod_data_aware_forward(const float* gap, int N, int C, const float* W1, const float* b1, int Hd, const float* W2,
                               const float* b2, int P, const float* E, int F, float* h1, float* h2, float* daf,
                               wsovod_stream_t stream) {
  if (N == 0) return WSOVOD_OK;
  WS_CHECK_ARG(gap && W1 && b1 && W2 && b2 && E && h1 && h2 && daf, "wsovod_data_aware_forward: null pointer");
  static int slot = wsovod::prof_slot("data_aware_fwd");
  hipStream_t s = (hipStream_t)stream;
  wsovod::ProfScope prof(slot, s, 0.0, 0.0);
  hipLaunchKernelGGL(data_aware_h1_kernel, dim3((unsigned)(((long long)N * Hd + 3) / 4)), dim3(256), 0, s, gap, C, W1, b1,
                     Hd, N, h1);
  hipLaunchKernelGGL(data_aware_fwd_kernel, dim3(N), dim3(256), (Hd + P) * sizeof(float), s, W2, b2, P, E, F, h1, Hd, h2,
                     daf);
  WS_CHECK_LAUNCH("wsovod_data_aware_forward");
  return WSOVOD_OK;
}

int wsovod_data_aware_backward(const float* ddaf, int N, const float* gap, int C, const float* W2, const float* E,
                               int F, const float* h1, int Hd, const float* h2, int P, float* dW1, float* db1,
                               float* dW2, float* db2, float* dE, float* scratch, wsovod_stream_t stream) {
  WS_CHECK_ARG(dW1 && db1 && dW2 && db2 && dE && scratch, "wsovod_data_aware_backward: null pointer");
  static int slot = wsovod::prof_slot("data_aware_bwd");
  hipStream_t s = (hipStream_t)stream;
  wsovod::ProfScope prof(slot, s, 0.0, 0.0);
  float* dpre2 = scratch;                 // N*P
  float* dh1 = scratch + (size_t)N * P;   // N*Hd
  if (N > 0)
    hipLaunchKernelGGL(data_aware_bwd_stage1, dim3(N), dim3(256), P * sizeof(float), s, ddaf, W2, E, F, h1, Hd, h2, P,
                       dpre2, dh1);
  const long long total = (long long)P * F + (long long)P * Hd + P + (long long)Hd * C + Hd;
  hipLaunchKernelGGL(data_aware_bwd_stage2, dim3((int)std::min<long long>(ceil_div_ll(total, 256), 1024)), dim3(256),
                     0, s, ddaf, N, gap, C, F, h1, Hd, h2, P, dpre2, dh1, dW1, db1, dW2, db2, dE);
  WS_CHECK_LAUNCH("wsovod_data_aware_backward");
  return WSOVOD_OK;
}

int wsovod_subsample_labels(const long long* labels, const float* keys, const int* seg_offsets, int G, int max_rows,
                            int num_samples, int pos_cap, long long bg_label, long long* out_labels,
                            wsovod_stream_t stream) {
  if (G == 0 || max_rows == 0) return WSOVOD_OK;
  WS_CHECK_ARG(labels && keys && seg_offsets && out_labels && labels != out_labels && num_samples >= 0 && pos_cap >= 0,
               "wsovod_subsample_labels: bad argument");
  static int slot = wsovod::prof_slot("subsample_labels");
  hipStream_t s = (hipStream_t)stream;
  wsovod::ProfScope prof(slot, s, 0.0, 0.0);
  hipLaunchKernelGGL(subsample_labels_kernel, dim3(ceil_div(max_rows, 256), G), dim3(256), 0, s, labels, keys,
                     seg_offsets, num_samples, pos_cap, bg_label, out_labels);
  WS_CHECK_LAUNCH("wsovod_subsample_labels");
  return WSOVOD_OK;
}

}  // extern "C"

/*
 * wsovod_hip.h -- C-ABI of the MI355X-native (gfx950) implementation of WSOVOD's
 * per-image detection hot path.  This is the drop-in boundary: plain pointers and
 * sizes, no torch types.  Every pointer is a DEVICE pointer unless its name ends in
 * `_host`.  `stream` is a hipStream_t (NULL = the legacy default stream).
 *
 * The reference binds its native layer through a pybind module `wsovod._C`
 * (/root/reference/wsovod/layers/vision.cpp:9-13) exporting
 *   roi_loop_pool_forward / roi_loop_pool_backward / csc_forward
 * and leans on ATen (cuDNN/cuBLAS) + torchvision for everything else on the path.
 * Each entry point below names the reference interface it replaces.
 *
 * Error convention (replaces AT_ASSERTM/AT_ERROR -> RuntimeError,
 * wsovod/layers/ROILoopPool/ROILoopPool_cuda.cu:258-265,311,384): functions return
 * WSOVOD_OK (0) or a non-zero wsovod_status; wsovod_last_error() returns the message
 * of the last failure on the calling thread.  Empty problems (zero rows) succeed and
 * launch nothing, like the reference's early returns (ROILoopPool_cuda.cu:288-291).
 *
 * Threading/streams: no internal threads, no global mutable state besides the optional
 * profiling table; kernels are enqueued on the caller's stream and never synchronise.
 */
#ifndef WSOVOD_HIP_H_
#define WSOVOD_HIP_H_

#ifdef __cplusplus
extern "C" {
#endif

typedef void* wsovod_stream_t; /* hipStream_t */

typedef enum {
  WSOVOD_OK = 0,
  WSOVOD_ERR_INVALID_ARGUMENT = 1,
  WSOVOD_ERR_HIP = 2,
  WSOVOD_ERR_UNSUPPORTED = 3
} wsovod_status;

/* WSOVOD_BF16X2 ("bf16x2", MODEL.HIP.PRECISION = "parity"): an fp32-grade value stored as a PAIR of bf16 numbers,
 * hi = bf16(x) and lo = bf16(x - hi) (x = hi + lo up to 2^-17 |x|), in the 4 bytes an fp32 element would take.  A row of
 * K values (K a multiple of 32) is laid out in groups of 32: element k = 32 g + j has its hi at bf16 index 64 g + j and
 * its lo at 64 g + 32 + j, so every 128-byte line holds the hi AND the lo halves of 32 consecutive elements and a
 * contraction kernel stages a bf16x2 operand exactly like a bf16 operand of twice the length.  Leading dimensions of
 * bf16x2 operands are counted in VALUES (4-byte slots), as for fp32.  A contraction over bf16x2 operands evaluates
 * sum_k (ah*bh + ah*bl + al*bh) on the bf16 MFMA pipe with fp32 accumulation: relative error ~2^-16 per product
 * instead of bf16's 2^-8 -- the precision the north star's 1e-3 logit bound needs. */
/* WSOVOD_BF16X2P (round 5, "planar" bf16x2): the same (hi, lo) pairs as TWO bf16 matrices of the tensor's shape, all hi
 * values first, all lo values `numel` elements further -- the 4 bytes per value of WSOVOD_BF16X2 in another order.  It is
 * an OUTPUT format of the RoI poolers (wsovod_roi_pool_forward_ws / wsovod_roi_align_forward_x2hi, out_dtype) and an
 * INPUT format of wsovod_gemm_nt's A operand (wsovod_gemm_desc.a_plane_bytes): the pooled tensor feeds the first FC layer's
 * three-product forward (hi and lo planes) AND its bf16 weight-gradient contraction (the hi plane alone, a plain bf16
 * matrix), which in the interleaved format needed a second, plain-bf16 copy of every pooled value (822 MB per 32 images). */
typedef enum { WSOVOD_F32 = 0, WSOVOD_BF16 = 1, WSOVOD_BF16X2 = 2, WSOVOD_BF16X2P = 3, WSOVOD_F16MX = 4 } wsovod_dtype;
/* Feature-map layout.  NCHW is the reference's layout; NHWC is what the HIP backbone
 * produces (torch.channels_last memory format on the Python side). */
typedef enum { WSOVOD_NCHW = 0, WSOVOD_NHWC = 1 } wsovod_layout;

const char* wsovod_last_error(void);
/* ABI version of this header; bumped on any signature change. */
int wsovod_abi_version(void);

/* ------------------------------------------------------------------------------------
 * Profiling: when enabled every launcher brackets its kernel with hipEvents on the
 * launch stream and accumulates per-kernel launch count, milliseconds, algorithmic
 * FLOPs and algorithmic bytes.  bench.py reads the table for its "roofline" object.
 * ---------------------------------------------------------------------------------- */
typedef struct {
  const char* name; /* stable kernel family name, e.g. "gemm_nt_bf16_128x128" */
  long long launches;
  double ms;    /* sum of event-measured durations */
  double flops; /* sum of algorithmic FLOPs (2*MACs) */
  double bytes; /* sum of algorithmic HBM bytes */
} wsovod_prof_entry;
int wsovod_profile_enable(int on); /* returns previous state */
int wsovod_profile_reset(void);
/* Synchronises pending events, then copies up to `cap` entries; returns the count. */
int wsovod_profile_collect(wsovod_prof_entry* out_host, int cap);

/* convert_boxes_to_pooler_format (wsovod/modeling/poolers.py:74-108) on the concatenated boxes of all images:
 *   boxes (M,4) fp32, seg_offsets (G+1) int32 prefix offsets of the per-image counts
 *   -> rois (M,5) fp32 [image index, x0, y0, x1, y1]
 * and, when objectness (M) / roi_scale (M) are given, roi_scale = objectness + 1 (roi_heads.py:733-739). */
int wsovod_format_rois(const float* boxes, const int* seg_offsets, int G, int M, const float* objectness,
                       float* rois, float* roi_scale, wsovod_stream_t stream);

/* ------------------------------------------------------------------------------------
 * RoI max pooling.  Replaces torchvision.ops.RoIPool as used by
 * wsovod/modeling/poolers.py:183-186,284 and the reference's own native op
 * roi_loop_pool_forward/backward (wsovod/layers/vision.cpp:10-11; algorithm
 * wsovod/layers/ROILoopPool/ROILoopPool_cpu.cpp:13-123).
 *   feat : (N,C,H,W) in `layout`, dtype `dtype`
 *   rois : (R,5) fp32 [batch_idx,x0,y0,x1,y1] (poolers.py:74-108)
 *   roi_scale : optional (R) fp32; out = max * roi_scale[r]  (fuses the objectness
 *               scaling of wsovod/modeling/roi_heads/roi_heads.py:733-739); NULL = 1
 *   out  : (R,C,ph,pw) contiguous, dtype `out_dtype`
 *   argmax : (R,C,ph,pw) int32, h*W+w of the max or -1 for an empty bin; may be NULL
 * Index arithmetic is fp32/int32 exactly as the reference writes it (bit-exact).
 * ---------------------------------------------------------------------------------- */
int wsovod_roi_pool_forward(const void* feat, int dtype, int layout, const float* rois,
                            const float* roi_scale, int R, int N, int C, int H, int W, int ph,
                            int pw, float spatial_scale, void* out, int out_dtype, int* argmax,
                            wsovod_stream_t stream);
/* The same with a bf16x2 output (out_dtype = WSOVOD_BF16X2) AND, in `out_hi` (R,C,ph,pw bf16, may be NULL), the plain bf16
 * rounding of the same values: the operand of the bf16 weight-gradient contraction of the first FC layer in the "parity"
 * precision, which would otherwise fetch the hi halves out of the bf16x2 rows as half lines (wsovod_gemm_tn_ex). */
int wsovod_roi_pool_forward_x2hi(const void* feat, int dtype, int layout, const float* rois, const float* roi_scale,
                                 int R, int N, int C, int H, int W, int ph, int pw, float spatial_scale, void* out,
                                 int out_dtype, int* argmax, void* out_hi, wsovod_stream_t stream);
/* The general form.  `workspace` (may be NULL) is caller-owned scratch of wsovod_roi_pool_workspace_bytes(...) bytes: when
 * that function returns > 0 (NHWC map, 7 bins wide, values only, enough rois to re-read the map many times over) the
 * launcher first writes the map's stride-1 2x2 maxima there and the pooling kernel covers every bin of >= 2 x 2 cells
 * with windows of THAT map -- a quarter of the gather's requests; max is order-free, so the values are the same bit for
 * bit.  With workspace = NULL (or argmax wanted: the first maximum in scan order needs the cells themselves) the cell
 * scan runs. */
long long wsovod_roi_pool_workspace_bytes(int dtype, int layout, int R, int N, int C, int H, int W, int ph, int pw,
                                          int want_argmax);
int wsovod_roi_pool_forward_ws(const void* feat, int dtype, int layout, const float* rois, const float* roi_scale, int R,
                               int N, int C, int H, int W, int ph, int pw, float spatial_scale, void* out, int out_dtype,
                               int* argmax, void* out_hi, void* workspace, long long workspace_bytes,
                               wsovod_stream_t stream);
/* Round 5 (ABI 7): the 2x2-max map of wsovod_roi_pool_forward_ws written TOGETHER with the global average pool of the same
 * NHWC map -- the input of the data-aware head (wsovod/modeling/class_heads.py:36-53, F.adaptive_avg_pool2d of the res5
 * map) -- in one pass over it, and the pooling on a map that is already there.  m2_out: N*H*W*C elements of `dtype`;
 * gap_out: (N, C) fp32 means; gap_workspace: as many floats as the _workspace_floats function below returns (partial sums, added in a fixed
 * order: run-to-run bit-identical).  wsovod_roi_pool_forward_m2 = wsovod_roi_pool_forward_ws without its own pre-pass:
 * `m2` must hold that map of `feat` (same values as _ws bit for bit); where _ws would not use a map (argmax wanted, few
 * rois, ...) it is ignored. */
long long wsovod_max2x2_gap_workspace_floats(int dtype, int N, int C, int H, int W);
int wsovod_max2x2_gap_nhwc(const void* feat, int dtype, int N, int C, int H, int W, void* m2_out, float* gap_out,
                           float* gap_workspace, wsovod_stream_t stream);
int wsovod_roi_pool_forward_m2(const void* feat, int dtype, int layout, const float* rois, const float* roi_scale, int R,
                               int N, int C, int H, int W, int ph, int pw, float spatial_scale, void* out, int out_dtype,
                               int* argmax, void* out_hi, const void* m2, long long m2_bytes, wsovod_stream_t stream);
/* ROILoopPool in the 3-output form of the reference's CUDA op (wsovod/layers/ROILoopPool/ROILoopPool_cuda.cu:9-204,
 * bound as `_C.roi_loop_pool_forward`, wsovod/layers/roi_loop_pool.py:9-22; context_ratio is 1.8 there): out and
 * argmax are (3R, C, ph, pw) = [region | frame | context] fp32 / int32 (NCHW order).  The matching backward is
 * wsovod_roi_pool_backward over the 3R outputs with the rois repeated three times (ROILoopPool_cuda.cu:207-243).
 * The map must be NHWC (wavefront per pooled row, channels per lane; WSOVOD_ERR_UNSUPPORTED otherwise): the Python
 * fronts bring the reference's NCHW tensors into channels_last first. */
int wsovod_roi_loop_pool_forward(const void* feat, int dtype, int layout, const float* rois, int R, int N, int C, int H,
                                 int W, int ph, int pw, float spatial_scale, float context_ratio, float* out,
                                 int* argmax, wsovod_stream_t stream);

/* grad_in (N,C,H,W) in `layout`, fp32, must be zero-filled by the caller; scatter-add
 * through argmax (ROILoopPool_cpu.cpp:82-123).  grad_out is (R,C,ph,pw) contiguous fp32. */
int wsovod_roi_pool_backward(const float* grad_out, const float* rois, const float* roi_scale,
                             const int* argmax, int R, int N, int C, int H, int W, int ph, int pw,
                             int layout, float* grad_in, wsovod_stream_t stream);

/* ROIAlign (aligned=True / False), replaces detectron2.layers.ROIAlign ->
 * torchvision roi_align as built by wsovod/modeling/poolers.py:169-182
 * (POOLER_TYPE ROIAlign / ROIAlignV2).  sampling_ratio<=0 means adaptive
 * ceil(roi_size/pooled_size).  Same tensor conventions as roi_pool. */
int wsovod_roi_align_forward(const void* feat, int dtype, int layout, const float* rois,
                             const float* roi_scale, int R, int N, int C, int H, int W, int ph,
                             int pw, float spatial_scale, int sampling_ratio, int aligned,
                             void* out, int out_dtype, wsovod_stream_t stream);
int wsovod_roi_align_forward_x2hi(const void* feat, int dtype, int layout, const float* rois, const float* roi_scale,
                                  int R, int N, int C, int H, int W, int ph, int pw, float spatial_scale,
                                  int sampling_ratio, int aligned, void* out, int out_dtype, void* out_hi,
                                  wsovod_stream_t stream);
int wsovod_roi_align_backward(const float* grad_out, const float* rois, const float* roi_scale,
                              int R, int N, int C, int H, int W, int ph, int pw,
                              float spatial_scale, int sampling_ratio, int aligned, int layout,
                              float* grad_in, wsovod_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Dense contraction C[M][N] = epilogue(sum_k A[m][k] * B[n][k])  ("NT": both operands
 * K-contiguous) on the matrix cores: bf16 MFMA (fp32 accumulate) or exact-fp32 MFMA.
 * Replaces ATen addmm/linear (cuBLAS) behind nn.Linear in
 * wsovod/modeling/roi_heads/box_head.py:60-75, fast_rcnn_open_vocabulary.py:277-285,681,
 * class_heads/open_vocabulary_classifier.py:39-44,102, data_aware_features_head.py:66-84,
 * and -- with `conv` set -- ATen conv2d (cuDNN) behind detectron2.layers.Conv2d in
 * wsovod/modeling/backbone/resnet_wsl.py:48-79,151-194,375-404 as an implicit GEMM over
 * an NHWC input (A is never materialised).
 *
 * Epilogue order:  v = alpha*acc; v *= row_scale[m]; v += bias[n]; v += residual[m][n];
 *   relu; dropout (inverted, keep-prob 1-p, counter-based RNG on (seed, m, n): one splitmix64
 *   value per quad of columns n & ~3 .. n | 3, 16 bits per element, keep iff bits >= p * 2^16);
 *   v += group_add[row_group[m]][n];  if mask_src: v = mask_src[m][n] > 0 ? v*mask_scale : 0;
 *   if accumulate: v += C_old.   C and the optional transposed copy Ct are then stored.
 * ---------------------------------------------------------------------------------- */
typedef struct {
  int n_img, H, W, Cin; /* input NHWC */
  int Ho, Wo;           /* output spatial size */
  int KH, KW, stride, pad, dil;
  int pool; /* 0, or 2: MaxPool2d(2, 2) of the conv output (stem tail resnet_wsl.py:418-420, BasicBlock tail :85-92)
               applied in the epilogue; C is then the pooled (n_img, Ho/2, Wo/2, N) map and the full-resolution output is
               never written.  Only the bf16 64 -> 64 channel 3x3 stride-1 kernel implements it (error otherwise). */
} wsovod_conv_geom;

typedef struct {
  int dtype_in; /* element type of A and B */
  int M, N, K;
  const void* A;
  long long lda; /* elements */
  const void* B;
  long long ldb;
  void* C; /* may be NULL when only Ct is wanted */
  long long ldc;
  int dtype_c;
  void* Ct; /* optional [N][M] copy */
  long long ldct;
  int dtype_ct;
  float alpha;
  const float* row_scale; /* [M] */
  const float* bias;      /* [N] */
  const void* residual;   /* [M][N] */
  long long ldr;
  int dtype_r;
  int relu;
  float dropout_p;
  unsigned long long dropout_seed;
  const int* row_group;   /* [M] -> group index */
  const float* group_add; /* [G][N] fp32 */
  long long ld_ga;
  const void* mask_src; /* [M][N] */
  long long ldm;
  int dtype_m;
  float mask_scale;
  int accumulate;             /* C (fp32 only) += result */
  int conv;                   /* 0 = plain GEMM, 1 = implicit-GEMM convolution */
  wsovod_conv_geom geom;      /* used when conv != 0; M = n_img*Ho*Wo, K = KH*KW*Cin */
  int tile_hint;              /* 0 = auto; else BM*1000+BN (e.g. 128128, 128064, 64064) */
  int prof_tag;               /* 0 = generic; >0 selects a named profiling slot */
  /* conv only: a second NHWC input A2 (n_img, Ho, Wo, Cin2) contracted 1x1 / stride 1 in the SAME accumulation -- the
   * block's projection shortcut (resnet_wsl.py:94-106: out = conv2(...) + shortcut(x)) folded into its last conv: B rows
   * are [W (KH*KW*Cin) | Wshortcut (Cin2)], K = KH*KW*Cin + Cin2, bias = the sum of the two folded biases.  The shortcut's
   * output is never written, rounded or re-read as a residual.  NULL = off. */
  const void* A2;
  int Cin2;
  /* optional DEVICE scalar added to dropout_seed when the kernel runs: the part of the seed that changes from step to
   * step (box_head.py: the training-step counter) kept in memory, so that a captured HIP graph of the training step draws
   * a new mask at every replay although its kernel arguments are frozen.  NULL = dropout_seed alone. */
  const unsigned long long* dropout_seed_add;
  /* dtype_in = WSOVOD_BF16X2, plain GEMM only: A is PLANAR bf16x2 (WSOVOD_BF16X2P) -- an (M, lda) bf16 matrix of hi halves at
   * A and the matrix of lo halves a_plane_bytes further (lda in values = bf16 elements of a plane row).  0 = the interleaved
   * layout.  Served by the lean two-phase 8-wavefront tile (K a multiple of 32 values; a_plane_bytes + 256 rows < 2 GiB). */
  long long a_plane_bytes;
} wsovod_gemm_desc;

int wsovod_gemm_nt(const wsovod_gemm_desc* desc_host, wsovod_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Image preprocessing.  Replaces GeneralizedRCNN_WSOVOD.preprocess_image
 * (wsovod/modeling/meta_arch/rcnn_wsovod.py:321-328): (x - mean) / std on uint8 CHW BGR
 * images already copied into one (N,3,Hp,Wp) canvas; pixels outside image n's own
 * (sizes[2n], sizes[2n+1]) = (h, w) are ZERO after normalisation (ImageList.from_tensors).
 * mean_host/std_host are HOST arrays of 3 floats.
 * wsovod_stem_im2col fuses that with the im2col of the stem's first conv (3x3, stride 2,
 * pad 1, Cin = 3; resnet_wsl.py:375-383): out is (N*Ho*Wo, 32), column k = (r*3+q)*3+c for
 * k < 27, zero above; Ho = (Hp-1)/2+1, Wo = (Wp-1)/2+1.  It is the A operand of wsovod_gemm_nt.
 * ---------------------------------------------------------------------------------- */
int wsovod_preprocess_image(const unsigned char* img, const int* sizes, const float* mean_host,
                            const float* std_host, int N, int Hp, int Wp, float* out_nchw,
                            wsovod_stream_t stream);
int wsovod_stem_im2col(const unsigned char* img, const int* sizes, const float* mean_host,
                       const float* std_host, int N, int Hp, int Wp, void* out, int out_dtype,
                       wsovod_stream_t stream);

/* The same layer fused end to end for bf16: uint8 images -> normalise -> 3x3/s2 conv (w32: the folded [64][32] bf16
 * weight in wsovod_stem_im2col's k order, bias: folded FrozenBN shift) -> ReLU -> (N, Ho, Wo, 64) bf16 NHWC.  The
 * im2col operand is never materialised; results are bit-identical to wsovod_stem_im2col + wsovod_gemm_nt. */
int wsovod_stem_conv1(const unsigned char* img, const int* sizes, const float* mean_host, const float* std_host, int N,
                      int Hp, int Wp, const void* w32, const float* bias, void* out, wsovod_stream_t stream);

/* 2x2 max pool over NHWC, stride 1 or 2; zero_pad_br=1 first pads one zero row/column at the
 * bottom/right (nn.ZeroPad2d((0,1,0,1)) + MaxPool2d(2, 1)).  Replaces the pools of
 * resnet_wsl.py:85-92,408. */
int wsovod_maxpool2x2_nhwc(const void* in, int dtype, int N, int H, int W, int C, int stride,
                           int zero_pad_br, void* out, wsovod_stream_t stream);
/* Its backward (round 6; a trainable res2 / res3 stage, resnet_wsl.py:85-92,530-552 under autograd): din (N,H,W,C) fp32 =
 * the gradient dout (N,Ho,Wo,C) fp32 routed to the FIRST maximum of every window in torch's scan order ((0,0), (0,1), (1,0),
 * (1,1), strict '>'); the padded zero cells take part in the comparison and drop their share.  `in` is the pool's input as
 * the forward saw it (dtype fp32 / bf16 / WSOVOD_BF16X2: hi + lo is compared). */
int wsovod_maxpool2x2_nhwc_backward(const void* in, int dtype, int N, int H, int W, int C, int stride, int zero_pad_br,
                                    const float* dout, float* din, wsovod_stream_t stream);
/* AdaptiveAvgPool2d(1) over NHWC -> (N,C) fp32 (data_aware_features_head.py:62,124).  workspace: caller-owned fp32
 * scratch sized by wsovod_colsum_workspace_floats for (G, M, N) = (N, N*HW, C) (see wsovod_segment_colsum). */
int wsovod_global_avgpool_nhwc(const void* in, int dtype, int N, int HW, int C, float* out, float* workspace,
                               wsovod_stream_t stream);
/* dst[c][r] = (dst_dtype) src[r][c]; leading dimensions in elements. */
int wsovod_transpose_cast(const void* src, int src_dtype, long long ld_src, int R, int C, void* dst,
                          int dst_dtype, long long ld_dst, wsovod_stream_t stream);
int wsovod_cast(const void* src, int src_dtype, void* dst, int dst_dtype, long long n,
                wsovod_stream_t stream);

/* Cosine-similarity head (open_vocabulary_classifier.py:91-92): row_scale[m] =
 * temperature / max(||x_m||_2, eps), consumed as the row_scale of wsovod_gemm_nt.
 * Backward of zn = row_scale * z (u = dL/dzn), optionally masked by the ReLU that produced z. */
int wsovod_row_l2norm_scale(const void* x, int dtype, long long ld, int M, int D, float temperature,
                            float eps, float* row_scale, wsovod_stream_t stream);
int wsovod_row_l2norm_backward(const void* z, int dtype, long long ldz, const float* u, long long ldu,
                               int M, int D, float temperature, float eps, int relu_mask, float* dz,
                               long long lddz, wsovod_stream_t stream);

/* out[g][n] (+)= sum over rows m in [seg_offsets[g], seg_offsets[g+1]) of x[m][n]  (autograd's bias / broadcast-add
 * gradients: fast_rcnn_open_vocabulary.py:318-367, roi_heads.py:762-763 under autograd).  A fixed-order two-stage
 * reduction: per 128-row chunk one fp32 partial per column (no atomics), then one pass that adds a segment's chunk
 * partials in chunk order -- run-to-run bit-identical, no quantisation, NaN / Inf propagate.  workspace: caller-owned
 * fp32 scratch of as many elements as wsovod_colsum_workspace_floats returns for (G, M, N) (not shared between streams). */
long long wsovod_colsum_workspace_floats(int G, int M, int N);
int wsovod_segment_colsum(const void* x, int dtype, long long ld, const int* seg_offsets, int G, int M,
                          int N, float* out, long long ldo, int accumulate, float* workspace, wsovod_stream_t stream);
/* x *= num[0] / den[0]; either pointer may be NULL (= 1). Device scalars: no host sync. */
int wsovod_scale_by_device_scalar(float* x, long long n, const float* num, const float* den,
                                  wsovod_stream_t stream);

/* Fused SGD step (torch.optim.SGD semantics as configured by wsovod/engine/defaults.py:274-318):
 *   g = grad*grad_scale + weight_decay*p;  buf = momentum*buf + g;  p -= lr*buf;
 * optionally refreshing a bf16 shadow copy of p in the same pass. */
int wsovod_sgd_momentum(float* param, const float* grad, float* momentum_buf, long long n, float lr,
                        float momentum, float weight_decay, float grad_scale, void* bf16_shadow,
                        wsovod_stream_t stream);

/* The same update for up to 32 tensors in ONE launch (every trainable tensor of the path is its own parameter group,
 * engine/defaults.py:274-318: 19 launches per step otherwise).  `tensors` is a HOST array. */
typedef struct wsovod_sgd_tensor {
  float* param;
  const float* grad;
  float* momentum_buf;
  void* bf16_shadow; /* optional bf16 copy of param refreshed in the same pass */
  long long numel;
  float lr, weight_decay;
  int grad_is_bf16; /* `grad` points at bf16 values (gradients that crossed the wire in bf16, see below) */
  int shadow_is_bf16x2; /* `bf16_shadow` is a bf16x2 copy (4 bytes per element, numel a multiple of 32): the "parity"
                         * precision's weight operand refreshed in the update pass instead of by wsovod_bf16x2_encode */
  const float* used_flag; /* optional DEVICE scalar: 0 = no data-parallel rank produced a gradient for this tensor in
                           * this step -> parameter and momentum stay untouched, as torch.optim.SGD skips `grad is None`
                           * under DDP(find_unused_parameters=True) (engine/defaults.py:146-148); NULL = always update */
  const float* grad_coef; /* optional DEVICE scalar multiplied into grad_scale: the norm-clipping coefficient of the
                           * wsovod_grad_clip_coef pass below (engine/defaults.py:292-318).  NULL = 1 */
  float clip_value;       /* > 0: grad * grad_scale is clamped to [-clip_value, clip_value] (detectron2's
                           * SOLVER.CLIP_GRADIENTS.CLIP_TYPE "value" = clip_grad_value_ per parameter); 0 = off */
  const float* lr_dev;    /* optional DEVICE scalar read instead of `lr`: the learning rate of a captured step graph lives
                           * in memory, so that a scheduler (detectron2's WarmupMultiStepLR changes it every iteration of
                           * the warm-up, engine/defaults.py build_lr_scheduler) takes effect at the next replay */
  const unsigned char* mx_scale; /* shadow_is_bf16x2 == 2 (round 6): `bf16_shadow` is an f16mx copy of the parameter (numel a
                           * multiple of 32) encoded with ONE power-of-two scale for the whole tensor, the E8M0 byte this
                           * DEVICE pointer names (wsovod_f16mx_encode_with): the "parity_mx" weight operand refreshed in the
                           * update pass.  NULL otherwise */
} wsovod_sgd_tensor;
int wsovod_sgd_momentum_multi(const wsovod_sgd_tensor* tensors, int count, float momentum, float grad_scale,
                              wsovod_stream_t stream);

/* Gradient-norm clipping coefficients on the device (no host read of the norm).  Replaces
 * torch.nn.utils.clip_grad_norm_(params, max_norm) (L2) as the reference applies it: over ALL parameters at once for
 * SOLVER.CLIP_GRADIENTS.CLIP_TYPE "full_model" (FullModelGradientClippingOptimizer, engine/defaults.py:292-318;
 * per_tensor = 0) or parameter by parameter for detectron2's CLIP_TYPE "norm" (per_tensor = 1).  The norm is taken of
 * grad * grad_scale (the gradient the optimizer sees after the data-parallel average); tensors whose used_flag reads 0
 * do not count (their grad is None in the reference).  coef[k] = min(1, max_norm / (norm + 1e-6)) for tensor k: pass
 * coef + k as that tensor's grad_coef to wsovod_sgd_momentum_multi.  Fixed summation order (run-to-run bit-identical).
 * workspace: wsovod_grad_clip_workspace_floats(tensors, count) fp32 elements; `tensors` is a HOST array (grad, numel,
 * grad_is_bf16 and used_flag are read). */
long long wsovod_grad_clip_workspace_floats(const wsovod_sgd_tensor* tensors, int count);
int wsovod_grad_clip_coef(const wsovod_sgd_tensor* tensors, int count, float grad_scale, float max_norm, int per_tensor,
                          float* workspace, float* coef, wsovod_stream_t stream);

/* bf16 gradient wire format (the reference's counterpart is DDP's fp16 compression hook, engine/defaults.py:149-152):
 * every tensor's fp32 gradient is rounded to bf16 into its slice of one flat buffer, the operand of ONE RCCL
 * all-reduce; wsovod_sgd_momentum_multi then reads the reduced slices (grad_is_bf16).  `tensors` is a HOST array. */
typedef struct wsovod_pack_tensor {
  const float* src;
  void* dst; /* bf16, 8-byte aligned for the vector path */
  long long numel;
} wsovod_pack_tensor;
int wsovod_pack_bf16_multi(const wsovod_pack_tensor* tensors, int count, wsovod_stream_t stream);

/* Direct (all-link) form of that exchange -- the reference has no counterpart, its DDP leaves the algorithm to NCCL
 * (engine/defaults.py:143-152): all-to-all of the wire buffer's shards over the point-to-point xGMI links, then this
 * kernel, then an all-gather.  src holds n_shards bf16 copies of this rank's shard, shard_elems (a multiple of 8) apart;
 * dst[e] = bf16(sum_j float(src[j * shard_elems + e])): fp32 accumulation, ONE rounding of the sum whatever the world
 * size (a ring all-reduce on bf16 rounds the running sum world-1 times). */
int wsovod_sum_shards_bf16(const void* src, int n_shards, long long shard_elems, void* dst, wsovod_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Proposal-concept MIL head.  Per-image segments: proposals of image g are rows
 * [seg_offsets[g], seg_offsets[g+1]).
 * wsovod_mil_forward replaces ObjectMiningOutputLayers.forward's
 *   softmax(C, dim=1) * softmax(D, dim=0) per image   (fast_rcnn_open_vocabulary.py:342-354)
 * on logits (M, 2K) = [C | D]; P, Q (M,K) are saved for the backward.
 * wsovod_image_bce_* replaces predict_probs_img + binary_cross_entropy (:604-618, :429-437):
 *   loss = sum BCE(clamp(sum_r scores, 1e-6, 1-1e-6), y) / norm.
 * ---------------------------------------------------------------------------------- */
int wsovod_mil_forward(const float* logits, long long ld, const int* seg_offsets, int G, int K,
                       float* scores, float* P, float* Q, int M, wsovod_stream_t stream);
int wsovod_mil_backward(const float* dscores, const float* P, const float* Q, const int* seg_offsets,
                        int G, int K, float* dlogits, long long ld, int M, wsovod_stream_t stream);
int wsovod_image_bce_forward(const float* scores, const int* seg_offsets, int G, int K,
                             const float* labels_onehot, float norm, float* img_scores, float* dS_img,
                             float* loss, wsovod_stream_t stream);
int wsovod_image_bce_backward(const float* dS_img, const int* seg_offsets, int G, int K,
                              const float* grad_out, float* dscores, wsovod_stream_t stream);

/* Weighted softmax cross-entropy of the instance-refinement branch
 * (fast_rcnn_open_vocabulary.py:799-802,813-820). gt_classes int64 in {-1, 0..K}; weighted=0
 * gives the plain mean over non-ignored rows.  dlogits receives the UN-normalised gradient;
 * accum2[1] the normaliser (scale with wsovod_scale_by_device_scalar(dlogits, n, gout, accum2+1)). */
int wsovod_weighted_ce_forward(const float* logits, long long ld, int M, int K1,
                               const long long* gt_classes, const float* weights, int weighted,
                               float* dlogits, long long ldd, float* accum2, float* loss,
                               wsovod_stream_t stream);
/* Weighted smooth-L1 box loss, class-agnostic deltas (fast_rcnn_open_vocabulary.py:822-892):
 * dpred receives d loss / d pred_deltas (already divided by the row count).  rows_true: optional DEVICE scalar with the
 * number of REAL rows when the M rows include padding (labels -1) behind the last image's proposals -- a captured step
 * graph runs on a bucketed row count and must normalise by the step's own; NULL = M. */
int wsovod_weighted_l1_box_forward(const float* pred_deltas, long long ldp, const float* proposal_boxes,
                                   const float* gt_boxes, const long long* gt_classes,
                                   const float* weights, int M, int K, const float* bbox_weights_host,
                                   float beta, int weighted, float* dpred, float* accum2, float* loss,
                                   const int* rows_true, wsovod_stream_t stream);

/* Weight-gradient contraction over the SLOW index of two row-major bf16 operands (no transposed copies):
 *   C[i][j] (+)= alpha * sum_m P[m][i] * Q[m][j],   P (Mred, NI) row stride ldp, Q (Mred, NJ) row stride ldq, C fp32.
 * nn.Linear backward dW = dY^T X (box_head.py:60-75, open_vocabulary_classifier.py:60-66 under autograd) with
 * P = dY, Q = X as the forward pass left them.  ldp, ldq, NI, NJ multiples of 8; operands < 2 GiB.
 * accumulate: bit 0 = add to C; bit 1 = keep a partial last round of 256x256 tiles unsplit.  By default such a round
 * (more than 256 tiles in all, at most 128 in the last round) is cut along the reduction into up to 8 slices per tile
 * that meet by fp32 atomic adds, so that it fills the chip; the summation order of those tiles is then not fixed. */
int wsovod_gemm_tn(const void* P, long long ldp, const void* Q, long long ldq, int Mred, int NI, int NJ, float* C,
                   long long ldc, float alpha, int accumulate, wsovod_stream_t stream);
/* The same with Q in `q_dtype` = WSOVOD_BF16 or WSOVOD_BF16X2 (ldq in values): of a bf16x2 Q only the hi halves are read,
 * i.e. Q is taken rounded to bf16 exactly as a cast would -- the bf16 weight gradient of the "parity" precision from the
 * activations its forward pass left in bf16x2, without a cast pass. */
int wsovod_gemm_tn_ex(const void* P, long long ldp, const void* Q, long long ldq, int q_dtype, int Mred, int NI, int NJ,
                      float* C, long long ldc, float alpha, int accumulate, wsovod_stream_t stream);
/* The weight-gradient contraction FUSED with the optimizer step of that weight (round 6): the tile's gradient
 *   g = alpha * sum_m P[m][i] * Q[m][j] * grad_scale
 * never goes to memory; the epilogue applies torch.optim.SGD's update (momentum, weight decay, dampening 0 -- the arithmetic
 * of wsovod_sgd_momentum_multi, engine/defaults.py:274-318 of the reference) to param (NI, NJ) fp32 contiguous in place:
 *   buf = momentum * buf + (g + weight_decay * param);  param -= lr * buf;  shadow refreshed (bf16 or bf16x2 copy).
 * Replaces, for ONE large weight at small batches, the pair (dW = dY^T X under autograd, box_head.py:60-75) + (the
 * optimizer's pass over that tensor, engine/trainer.py:72-84): 16 - 20 bytes per parameter instead of 28 - 32.  Whole tiles
 * only (no reduction slices: fixed summation order).  The caller owns the schedule: no clipping, no accumulation
 * (ITER_SIZE 1), no gradient exchange may be pending on this tensor (HotPathTrainer installs it at world = 1 only). */
typedef struct wsovod_tn_sgd {
  float* param;
  float* momentum_buf;
  void* shadow;          /* optional bf16 / bf16x2 copy of param (NULL = none) */
  int shadow_is_bf16x2;
  float lr, weight_decay, momentum, grad_scale;
  const float* lr_dev;   /* optional DEVICE scalar read instead of lr (captured step graphs) */
  const unsigned char* mx_scale; /* shadow_is_bf16x2 == 2: an f16mx shadow with this per-tensor E8M0 byte (DEVICE), as in
                          * wsovod_sgd_tensor */
} wsovod_tn_sgd;
int wsovod_gemm_tn_sgd(const void* P, long long ldp, const void* Q, long long ldq, int q_dtype, int Mred, int NI, int NJ,
                       float alpha, const wsovod_tn_sgd* update, wsovod_stream_t stream);

/* Round 6 ("parity_mx" precision): the forward contractions of the big layers -- the two FC layers of the box head
 * (box_head.py:60-75, F.linear) and the res4 / res5 convolutions (resnet_wsl.py:94-110) -- with the two cross terms of the
 * three-product forward on gfx950's block-scaled matrix instruction.  Operand format WSOVOD_F16MX ("f16mx"): rows of groups of
 * 32 values = 128 bytes [32 x fp16 hi | 32 x OCP e4m3 q | 32 x e4m3 ql] (4 bytes per value, leading dimensions in VALUES as
 * for WSOVOD_BF16X2),
 *   q = e4m3(x / 2^s),  ql = e4m3((x - hi) / 2^(s - 11))   (saturating at +-448; hi = fp16(x)).
 * ACTIVATIONS carry no scale (s = 0, "unit scale": what wsovod_gemm_f16mx writes with dtype_c = WSOVOD_F16MX, the RoI poolers
 * with out_dtype = WSOVOD_F16MX, wsovod_f16mx_from_bf16x2, and wsovod_f16mx_encode with scales = NULL); WEIGHTS one E8M0 byte
 * per ROW SEGMENT in scales[row][nseg] (nseg equal segments of the row), s = floor(log2(max |hi| of the segment)) - 7,
 * byte = s + 127 (wsovod_f16mx_encode; cols a multiple of 32 * nseg).
 * wsovod_gemm_f16mx: C = epilogue(A B^T) for f16mx A (M, K) -- a_scale NULL = unit scale -- and B (N, K); a scale segment is the
 * whole row or a multiple of 6 groups of 32; `d` as for wsovod_gemm_nt (dtype_in ignored) with the epilogue alpha / bias /
 * residual (fp32, bf16, bf16x2 or unit-scale f16mx) / ReLU / counter dropout and C in fp32, bf16, bf16x2 (N a multiple of 4) or
 * unit-scale f16mx (N a multiple of 16); c_bf16 (may be NULL): a plain bf16 copy of C, the operand of the next layer's weight
 * gradient and the mask source of this layer's backward.  d->conv = 1: the implicit-GEMM convolution of wsovod_gemm_nt on an
 * NHWC unit-scale f16mx map (Cin a multiple of 32; A2 / Cin2: the fused 1x1 projection shortcut, also f16mx).  Per product
 * hi_a hi_b + q_a ql_b + ql_a q_b  (two v_mfma_f32_32x32x16_f16 + one v_mfma_scale_f32_32x32x64_f8f6f4 per 32x32 tile). */
int wsovod_f16mx_encode(const float* src, long long ld_src, int rows, int cols, int nseg, void* dst, long long ld_dst,
                        unsigned char* scales, wsovod_stream_t stream);
/* The same with ONE scale for the whole tensor, given as an E8M0 byte in DEVICE memory (`tensor_scale`; chosen by the caller
 * from the tensor's largest magnitude with headroom); scales[rows] is filled with that byte, so the tensor is an ordinary
 * row-scaled operand of wsovod_gemm_f16mx.  A trained weight keeps its byte from step to step: the optimizer kernels re-encode
 * it element-wise inside the update pass (wsovod_sgd_tensor.mx_scale) instead of a separate two-pass encode per step. */
int wsovod_f16mx_encode_with(const float* src, long long ld_src, int rows, int cols, void* dst, long long ld_dst,
                             unsigned char* scales, const unsigned char* tensor_scale, wsovod_stream_t stream);
/* n values (whole groups of 32) of an interleaved bf16x2 tensor -> unit-scale f16mx (the map that crosses from the bf16x2
 * layers, res3, to the f16mx ones, res4). */
int wsovod_f16mx_from_bf16x2(const void* src, void* dst, long long n, wsovod_stream_t stream);
int wsovod_gemm_f16mx(const wsovod_gemm_desc* d, const unsigned char* a_scale, int a_segments, const unsigned char* b_scale,
                      int b_segments, void* c_bf16, long long ld_c_bf16, wsovod_stream_t stream);

/* Greedy non-maximum suppression over G independent segments of boxes that are already sorted by
 * descending score inside each segment.  Replaces torchvision.ops.nms / batched_nms (un-vendored; SURVEY
 * Appendix A) at the reference's call sites: find_top_rpn_proposals (proposal_utils.py:123; one segment per
 * image, idxs = level) and fast_rcnn_inference_single_image (fast_rcnn_open_vocabulary.py:176; one segment per
 * class = the exact per-class form of batched_nms).  A box suppresses a later one of its segment when
 * inter / (area_a + area_b - inter) > iou_threshold, all in fp32 in that operation order (bit-exact keep sets).
 *   boxes       : (N,4) f32 xyxy, 16-byte aligned; segment g owns rows [seg_offsets[g], seg_offsets[g+1])
 *   valid       : optional (N) bytes; a zero marks a box that was filtered out (never kept, never suppresses)
 *   max_seg_len : longest segment (host value; sizes the bitmap, <= 16384)
 *   max_keep    : stop after this many kept boxes per segment (<= 0: no limit)
 *   workspace   : N * max(1, ceil(max_seg_len / 64)) 64-bit words
 *   keep_idx    : (N) int32; segment g's kept positions (relative to its start, ascending = by score) are
 *                 written to keep_idx[seg_offsets[g] ...]; keep_count : (G) int32 how many. */
int wsovod_nms_segments(const float* boxes, const int* seg_offsets, const unsigned char* valid, int G, int N,
                        int max_seg_len, float iou_threshold, int max_keep, unsigned long long* workspace,
                        int* keep_idx, int* keep_count, wsovod_stream_t stream);

/* RPN proposal decode for the selected anchors of every image: Box2BoxTransform.apply_deltas (detectron2,
 * un-vendored; called from WSOVODRPN_V2._decode_proposals, rpn.py:495-515), then Boxes.clip
 * and the nonempty(threshold = min_size) test of find_top_rpn_proposals (proposal_utils.py:101-121).
 *   anchors (A,4) f32; deltas (num_images, A, 4) f32; index (num_images*per_image) int64 anchor ids per image
 *   (NULL: the first per_image anchors); image_sizes (num_images,2) f32 (h,w) on the device;
 *   weights: HOST array of the 4 Box2BoxTransform weights; boxes (num_images*per_image,4) out;
 *   valid (num_images*per_image) out: finite before clipping and both sides > min_size after. */
int wsovod_rpn_decode(const float* anchors, const float* deltas, const long long* index, int num_images, int per_image,
                      long long anchors_per_image, const float* image_sizes, const float* weights, float scale_clamp,
                      float min_size, float* boxes, unsigned char* valid, wsovod_stream_t stream);

/* RPN anchor labelling: detectron2 pairwise_iou + Matcher(thresholds [lo, hi], labels [0, -1, 1],
 * allow_low_quality_matches=True) (un-vendored) as used by
 * WSOVODRPN_V2.label_and_sample_anchors (rpn.py:237-293) before the random sub-sampling.
 *   anchors (A,4); gt_boxes (total_gt,4) pseudo-GT boxes, image b owns rows
 *   [gt_start[b], gt_start[b] + gt_count[b]) (device arrays: no host round trip); outputs per (image, anchor): best_iou f32, best_gt int32 (row of gt_boxes
 *   or -1), labels int8 in {1 positive, 0 negative, -1 ignore}; gt_best_ws: total_gt 32-bit words of scratch. */
int wsovod_rpn_label_anchors(const float* anchors, int A, const float* gt_boxes, const int* gt_start,
                             const int* gt_count, int num_images, int total_gt, float thr_lo, float thr_hi, float* best_iou, int* best_gt,
                             unsigned int* gt_best_ws, signed char* labels, wsovod_stream_t stream);

/* im2col rows of selected output pixels of an NHWC convolution input (weight gradient of the RPN's 3x3 conv,
 * detectron2 StandardRPNHead.conv, un-vendored; only the sampled anchors of rpn.py:217-233 carry a loss).
 *   x (n_img,H,W,Cin) NHWC in `dtype`; rows (n_rows) int64 flat output-pixel ids (img*Ho*Wo + ho*Wo + wo), a
 *   negative id yields a zero row; out (n_rows, KH*KW*Cin) in `dtype`, tap-major then channel. */
int wsovod_im2col_rows(const void* x, int dtype, const long long* rows, int n_rows, int H, int W, int Cin, int Ho, int Wo,
                       int KH, int KW, int stride, int pad, int dil, void* out, wsovod_stream_t stream);

/* Pseudo-ground-truth mining + proposal labelling, no grad.  Replaces
 * WSOVODROIHeads.get_pgt_top_k (roi_heads.py:1043-1343; top_k=1, sam=None) followed by
 * label_and_sample_proposals_wsl (roi_heads.py:1722-1825) with Matcher([thr],[0,1]) when every
 * proposal is kept (R <= BATCH_SIZE_PER_IMAGE, POSITIVE_FRACTION 1.0).
 *   gt_classes_img : concatenated sorted-unique image-level classes, image g owns
 *                    [gt_offsets[g], gt_offsets[g+1])  (get_image_level_gt, roi_heads.py:158-174)
 *   img_scores     : (G,K) clamped image-level scores (pred_class_img_logits)
 * Outputs per GT slot: pgt_* (pgt_index = proposal index inside its image or -1); per image:
 * pgt_count; per proposal: label, matched box / score / weight / matched slot. */
int wsovod_pgt_mine_and_label(const float* scores, long long ld_scores, const float* boxes,
                              const int* seg_offsets, int G, const long long* gt_classes_img,
                              const int* gt_offsets, const float* img_scores, int K,
                              float iou_threshold, float* pgt_boxes, long long* pgt_classes,
                              float* pgt_scores, float* pgt_weights, int* pgt_index, int* pgt_count,
                              long long* out_classes, float* out_boxes, float* out_scores,
                              float* out_weights, int* out_matched, wsovod_stream_t stream);

/* Proposal sub-sampling for the refinement losses, no grad.  Replaces detectron2 `subsample_labels` as called by
 * WSOVODROIHeads._sample_proposals_wsl (roi_heads.py:1566-1603) when an image has more proposals than
 * WSOVOD.SAMPLING.BATCH_SIZE_PER_IMAGE or POSITIVE_FRACTION < 1 (the shipped RPN configs: up to 4000 loaded + 1024
 * RPN boxes against 4096).  Per image g (rows [seg_offsets[g], seg_offsets[g+1]) of `labels`):
 *   positives = labels not in {-1, bg_label}, negatives = labels == bg_label,
 *   num_pos = min(#pos, pos_cap) with pos_cap = int(num_samples * positive_fraction), num_neg = min(#neg, num_samples - num_pos);
 *   the num_pos / num_neg rows of each group with the smallest (keys[row], row) keep their label, all others get -1.
 * keys ~ U[0,1) reproduces the reference's randperm sampling; keys = row index is the deterministic first-n rule.
 * out_labels must not alias labels.  max_rows = the longest segment (launch geometry only). */
int wsovod_subsample_labels(const long long* labels, const float* keys, const int* seg_offsets, int G, int max_rows,
                            int num_samples, int pos_cap, long long bg_label, long long* out_labels,
                            wsovod_stream_t stream);

/* bf16x3 operand split (MODEL.HIP.PRECISION = bf16x3: the mode that meets the north star's 1e-3 logit bound on the bf16
 * MFMA path).  An fp32 matrix src (rows, cols; row stride ld_src) becomes three bf16 blocks hi = bf16(x) and
 * lo = bf16(x - hi) in the order  side 0 (A operand): hi, hi, lo;  side 1 (B operand): hi, lo, hi  -- so that an
 * UNCHANGED bf16 contraction over the three-times-longer reduction index computes sum ah*bh + ah*bl + al*bh with fp32
 * accumulation (relative error ~2^-16 per product instead of bf16's 2^-8).  Block b of row r is written at
 * dst + b * block_stride + r * ld_dst: block_stride = cols puts the blocks side by side along K (Linear layers:
 * fast_rcnn_open_vocabulary.py:318-367, open_vocabulary_classifier.py:79-105; NHWC conv channels: resnet_wsl.py),
 * block_stride = padded_rows * ld_dst stacks them along the rows (weight-gradient contraction over proposals). */
int wsovod_split3_bf16(const float* src, long long ld_src, int rows, int cols, void* dst, long long ld_dst,
                       long long block_stride, int side, wsovod_stream_t stream);

/* Backward prologue of Linear+ReLU(+Dropout) (box_head.py:60-66): dA = dy * [y > 0] * scale
 * written as [M][N] and/or transposed [N][ldt] (either output may be NULL; y NULL = no mask). */
int wsovod_mask_transpose(const void* dy, long long lddy, const void* y, long long ldy, int in_dtype,
                          int M, int N, float scale, void* dA, long long ldda, void* dAt, long long ldt,
                          int out_dtype, wsovod_stream_t stream);
/* The same pass, also ADDING the column sums of dA (fp32, before rounding to out_dtype) into colsum[N]: the bias
 * gradient of the Linear layer (autograd's sum over rows) without a second read of dA.  The caller zero-fills colsum. */
int wsovod_mask_transpose_colsum(const void* dy, long long lddy, const void* y, long long ldy, int in_dtype,
                                 int M, int N, float scale, void* dA, long long ldda, void* dAt, long long ldt,
                                 int out_dtype, float* colsum, wsovod_stream_t stream);
/* General form: dy in `dy_dtype` (fp32 / bf16), the mask source y in `y_dtype` = dy_dtype, WSOVOD_BF16X2 (the layer's
 * own output as the "parity" forward pass left it; ldy in values) or -- next to an fp32 dy -- WSOVOD_BF16 (the plain bf16
 * rounding that wsovod_gemm_f16mx writes beside an f16mx output; round 6), colsum optional (NULL = none). */
int wsovod_mask_transpose_ex(const void* dy, long long lddy, int dy_dtype, const void* y, long long ldy, int y_dtype, int M,
                             int N, float scale, void* dA, long long ldda, void* dAt, long long ldt, int out_dtype,
                             float* colsum, wsovod_stream_t stream);
/* fp32 (rows, cols; row stride ld_src) <-> bf16x2 (row stride ld_dst / ld_src in values), cols a multiple of 32.  Weights
 * are encoded once per optimizer step; activations are produced in bf16x2 by the kernels themselves
 * (wsovod_gemm_nt with dtype_c = WSOVOD_BF16X2, wsovod_roi_pool_forward / wsovod_roi_align_forward with out_dtype =
 * WSOVOD_BF16X2, wsovod_stem_conv1_x2, wsovod_maxpool2x2_nhwc / wsovod_add_group_rows with dtype = WSOVOD_BF16X2). */
int wsovod_bf16x2_encode(const float* src, long long ld_src, int rows, int cols, void* dst, long long ld_dst,
                         wsovod_stream_t stream);
int wsovod_bf16x2_decode(const void* src, long long ld_src, int rows, int cols, float* dst, long long ld_dst,
                         wsovod_stream_t stream);
/* wsovod_stem_conv1 for the "parity" precision: w32x2 = the bf16x2 encoding of the folded (64, 32) fp32 weight, out =
 * (N, Ho, Wo, 64) bf16x2 NHWC; the normalised image is split into hi / lo in LDS and every product is the three-MFMA
 * sum w_hi*a_hi + w_lo*a_hi + w_hi*a_lo (resnet_wsl.py:375-383,410-413; rcnn_wsovod.py:321-328). */
int wsovod_stem_conv1_x2(const unsigned char* img, const int* sizes, const float* mean_host, const float* std_host, int N,
                         int Hp, int Wp, const void* w32x2, const float* bias, void* out, wsovod_stream_t stream);
/* out[m][:] = x[m][:] + add[row_group[m]][:]  (box_features += data_aware_features,
 * roi_heads.py:762-763, without materialising the per-proposal repeat). */
int wsovod_add_group_rows(const void* x, long long ldx, int dtype, const int* row_group, const float* add,
                          long long ld_add, int M, int N, void* out, long long ldo, wsovod_stream_t stream);
/* out[r][:] = x[r][:] * row_scale[r]  (F.normalize of the class text embeddings,
 * open_vocabulary_classifier.py:59-60,87-89). */
int wsovod_scale_rows(const float* x, long long ldx, const float* row_scale, int R, int C, void* out,
                      long long ldo, int out_dtype, wsovod_stream_t stream);

/* DataAwareFeaturesHead on the GAP vector (data_aware_features_head.py:103-129):
 * h1 = relu(W1 gap + b1) (N,Hd); h2 = tanh(W2 h1 + b2) (N,P); daf = h2 @ E (N,F).  All fp32.
 * The backward takes d loss / d daf (N,F) and OVERWRITES the five parameter gradients. */
int wsovod_data_aware_forward(const float* gap, int N, int C, const float* W1, const float* b1, int Hd,
                              const float* W2, const float* b2, int P, const float* E, int F, float* h1,
                              float* h2, float* daf, wsovod_stream_t stream);
int wsovod_data_aware_backward(const float* ddaf, int N, const float* gap, int C, const float* W2,
                               const float* E, int F, const float* h1, int Hd, const float* h2, int P,
                               float* dW1, float* db1, float* dW2, float* db2, float* dE,
                               float* scratch /* N*(P+Hd) floats */, wsovod_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* WSOVOD_HIP_H_ */

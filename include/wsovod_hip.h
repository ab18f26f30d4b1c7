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

typedef enum { WSOVOD_F32 = 0, WSOVOD_BF16 = 1 } wsovod_dtype;
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
 *   relu; dropout (inverted, keep-prob 1-p, counter-based RNG on (seed,m,n));
 *   v += group_add[row_group[m]][n];  if mask_src: v = mask_src[m][n] > 0 ? v*mask_scale : 0;
 *   if accumulate: v += C_old.   C and the optional transposed copy Ct are then stored.
 * ---------------------------------------------------------------------------------- */
typedef struct {
  int n_img, H, W, Cin; /* input NHWC */
  int Ho, Wo;           /* output spatial size */
  int KH, KW, stride, pad, dil;
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
} wsovod_gemm_desc;

int wsovod_gemm_nt(const wsovod_gemm_desc* desc_host, wsovod_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* WSOVOD_HIP_H_ */

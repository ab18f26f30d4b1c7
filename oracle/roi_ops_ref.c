/*
 * ORACLE -- test infrastructure only.  Plain-C CPU restatement of the reference's RoI
 * pooling arithmetic.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may call this; the product path (wsovod_amd/) never does.
 *
 * roi_pool_forward / roi_pool_backward follow
 *   /root/reference/wsovod/layers/ROILoopPool/ROILoopPool_cpu.cpp:13-80 (forward)
 *   /root/reference/wsovod/layers/ROILoopPool/ROILoopPool_cpu.cpp:82-123 (backward)
 * which is the same algorithm as torchvision.ops.RoIPool used by
 *   /root/reference/wsovod/modeling/poolers.py:183-186.
 * roi_align_forward / roi_align_backward restate torchvision 0.13 `roi_align`
 * (un-vendored third-party dependency: detectron2.layers.ROIAlign -> torchvision.ops.roi_align,
 * call site /root/reference/wsovod/modeling/poolers.py:169-182); see SURVEY.md Appendix A.
 *
 * Layout: input (N,C,H,W) fp32 contiguous, rois (R,5) fp32, output (R,C,PH,PW).
 * Pinned against oracle/_ref (the reference's own ROILoopPool_cpu.cpp compiled where it
 * lies) by tests/test_oracle_golden.py and against the golden vectors in tests/golden/; roi_align_* is checked
 * against an independent fp64 evaluator written from the operator's definition (tests/test_oracle_properties.py).
 */
#include <float.h>
#include <math.h>
#include <string.h>

static int imin(int a, int b) { return a < b ? a : b; }
static int imax(int a, int b) { return a > b ? a : b; }

void roi_pool_forward(const float* input, float spatial_scale, int channels, int height, int width,
                      int pooled_height, int pooled_width, const float* rois, int num_rois,
                      float* output, int* argmax_data) {
  for (int n = 0; n < num_rois; ++n) {
    const float* roi = rois + n * 5;
    int roi_batch_ind = (int)roi[0];
    int roi_start_w = (int)roundf(roi[1] * spatial_scale);
    int roi_start_h = (int)roundf(roi[2] * spatial_scale);
    int roi_end_w = (int)roundf(roi[3] * spatial_scale);
    int roi_end_h = (int)roundf(roi[4] * spatial_scale);
    /* malformed ROIs become 1x1 */
    int roi_width = imax(roi_end_w - roi_start_w + 1, 1);
    int roi_height = imax(roi_end_h - roi_start_h + 1, 1);
    float bin_size_h = (float)roi_height / (float)pooled_height;
    float bin_size_w = (float)roi_width / (float)pooled_width;
    for (int ph = 0; ph < pooled_height; ++ph) {
      for (int pw = 0; pw < pooled_width; ++pw) {
        int hstart = (int)floorf((float)ph * bin_size_h);
        int wstart = (int)floorf((float)pw * bin_size_w);
        int hend = (int)ceilf((float)(ph + 1) * bin_size_h);
        int wend = (int)ceilf((float)(pw + 1) * bin_size_w);
        hstart = imin(imax(hstart + roi_start_h, 0), height);
        hend = imin(imax(hend + roi_start_h, 0), height);
        wstart = imin(imax(wstart + roi_start_w, 0), width);
        wend = imin(imax(wend + roi_start_w, 0), width);
        int is_empty = (hend <= hstart) || (wend <= wstart);
        for (int c = 0; c < channels; ++c) {
          float maxval = is_empty ? 0.f : -FLT_MAX;
          int maxidx = -1;
          const float* plane = input + ((long)roi_batch_ind * channels + c) * height * width;
          for (int h = hstart; h < hend; ++h)
            for (int w = wstart; w < wend; ++w) {
              int idx = h * width + w;
              if (plane[idx] > maxval) {
                maxval = plane[idx];
                maxidx = idx;
              }
            }
          long o = (((long)n * channels + c) * pooled_height + ph) * pooled_width + pw;
          output[o] = maxval;
          argmax_data[o] = maxidx;
        }
      }
    }
  }
}

/* grad_input (N,C,H,W) must be zero-filled by the caller. */
void roi_pool_backward(const float* grad_output, const int* argmax_data, int num_rois, int channels,
                       int height, int width, int pooled_height, int pooled_width,
                       float* grad_input, const float* rois) {
  for (int n = 0; n < num_rois; ++n) {
    int roi_batch_ind = (int)rois[n * 5];
    for (int c = 0; c < channels; ++c) {
      float* gi = grad_input + ((long)roi_batch_ind * channels + c) * height * width;
      long base = ((long)n * channels + c) * pooled_height * pooled_width;
      for (int i = 0; i < pooled_height * pooled_width; ++i) {
        int am = argmax_data[base + i];
        if (am != -1) gi[am] += grad_output[base + i];
      }
    }
  }
}

/* ---- ROIAlign (torchvision roi_align) ---- */
typedef struct {
  int yl, yh, xl, xh, valid;
  float w1, w2, w3, w4;
} bilin_t;

static bilin_t bilinear_setup(float y, float x, int height, int width) {
  bilin_t s;
  s.valid = !(y < -1.0f || y > (float)height || x < -1.0f || x > (float)width);
  if (y <= 0) y = 0;
  if (x <= 0) x = 0;
  s.yl = (int)y;
  s.xl = (int)x;
  if (s.yl >= height - 1) {
    s.yh = s.yl = height - 1;
    y = (float)s.yl;
  } else {
    s.yh = s.yl + 1;
  }
  if (s.xl >= width - 1) {
    s.xh = s.xl = width - 1;
    x = (float)s.xl;
  } else {
    s.xh = s.xl + 1;
  }
  float ly = y - s.yl, lx = x - s.xl, hy = 1.f - ly, hx = 1.f - lx;
  s.w1 = hy * hx;
  s.w2 = hy * lx;
  s.w3 = ly * hx;
  s.w4 = ly * lx;
  return s;
}

void roi_align_forward(const float* input, float spatial_scale, int channels, int height, int width,
                       int pooled_height, int pooled_width, int sampling_ratio, int aligned,
                       const float* rois, int num_rois, float* output) {
  for (int n = 0; n < num_rois; ++n) {
    const float* roi = rois + n * 5;
    int b = (int)roi[0];
    float offset = aligned ? 0.5f : 0.0f;
    float sw = roi[1] * spatial_scale - offset, sh = roi[2] * spatial_scale - offset;
    float ew = roi[3] * spatial_scale - offset, eh = roi[4] * spatial_scale - offset;
    float rw = ew - sw, rh = eh - sh;
    if (!aligned) {
      rw = fmaxf(rw, 1.f);
      rh = fmaxf(rh, 1.f);
    }
    float bin_h = rh / (float)pooled_height, bin_w = rw / (float)pooled_width;
    int gh = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(rh / (float)pooled_height);
    int gw = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(rw / (float)pooled_width);
    float inv_count = 1.0f / (float)imax(gh * gw, 1);
    for (int c = 0; c < channels; ++c) {
      const float* plane = input + ((long)b * channels + c) * height * width;
      for (int ph = 0; ph < pooled_height; ++ph)
        for (int pw = 0; pw < pooled_width; ++pw) {
          float acc = 0.f;
          for (int iy = 0; iy < gh; ++iy) {
            float y = sh + (float)ph * bin_h + ((float)iy + .5f) * bin_h / (float)gh;
            for (int ix = 0; ix < gw; ++ix) {
              float x = sw + (float)pw * bin_w + ((float)ix + .5f) * bin_w / (float)gw;
              bilin_t s = bilinear_setup(y, x, height, width);
              if (!s.valid) continue;
              acc += s.w1 * plane[s.yl * width + s.xl] + s.w2 * plane[s.yl * width + s.xh] +
                     s.w3 * plane[s.yh * width + s.xl] + s.w4 * plane[s.yh * width + s.xh];
            }
          }
          output[(((long)n * channels + c) * pooled_height + ph) * pooled_width + pw] = acc * inv_count;
        }
    }
  }
}

void roi_align_backward(const float* grad_output, float spatial_scale, int channels, int height,
                        int width, int pooled_height, int pooled_width, int sampling_ratio,
                        int aligned, const float* rois, int num_rois, float* grad_input) {
  for (int n = 0; n < num_rois; ++n) {
    const float* roi = rois + n * 5;
    int b = (int)roi[0];
    float offset = aligned ? 0.5f : 0.0f;
    float sw = roi[1] * spatial_scale - offset, sh = roi[2] * spatial_scale - offset;
    float ew = roi[3] * spatial_scale - offset, eh = roi[4] * spatial_scale - offset;
    float rw = ew - sw, rh = eh - sh;
    if (!aligned) {
      rw = fmaxf(rw, 1.f);
      rh = fmaxf(rh, 1.f);
    }
    float bin_h = rh / (float)pooled_height, bin_w = rw / (float)pooled_width;
    int gh = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(rh / (float)pooled_height);
    int gw = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(rw / (float)pooled_width);
    float inv_count = 1.0f / (float)imax(gh * gw, 1);
    for (int c = 0; c < channels; ++c) {
      float* gi = grad_input + ((long)b * channels + c) * height * width;
      for (int ph = 0; ph < pooled_height; ++ph)
        for (int pw = 0; pw < pooled_width; ++pw) {
          float g = grad_output[(((long)n * channels + c) * pooled_height + ph) * pooled_width + pw] * inv_count;
          for (int iy = 0; iy < gh; ++iy) {
            float y = sh + (float)ph * bin_h + ((float)iy + .5f) * bin_h / (float)gh;
            for (int ix = 0; ix < gw; ++ix) {
              float x = sw + (float)pw * bin_w + ((float)ix + .5f) * bin_w / (float)gw;
              bilin_t s = bilinear_setup(y, x, height, width);
              if (!s.valid) continue;
              gi[s.yl * width + s.xl] += g * s.w1;
              gi[s.yl * width + s.xh] += g * s.w2;
              gi[s.yh * width + s.xl] += g * s.w3;
              gi[s.yh * width + s.xh] += g * s.w4;
            }
          }
        }
    }
  }
}

/* ROILoopPool, 3-output form: restated from the reference's CUDA kernel
 * (wsovod/layers/ROILoopPool/ROILoopPool_cuda.cu:9-204; launched with context_ratio 1.8 at :311) -- the reference's
 * CPU source implements the first output only, so this part of the oracle is pinned by source reading, not by a
 * compiled reference ("parity unpinned" for frame / context; the region output equals roi_pool_forward on
 * non-negative inputs, which IS pinned against oracle/_ref).
 * out, argmax: (3R, C, PH, PW) = [region | frame | context]. */
void roi_loop_pool_forward(const float* input, float spatial_scale, int channels, int height, int width,
                           int pooled_height, int pooled_width, const float* rois, int num_rois, float context_ratio,
                           float* output, int* argmax_data) {
  const long part = (long)num_rois * channels * pooled_height * pooled_width;
  for (int n = 0; n < num_rois; ++n) {
    const float* roi = rois + n * 5;
    const int b = (int)roi[0];
    const float x1 = roi[1], y1 = roi[2], x2 = roi[3], y2 = roi[4];
    const float rois_w = x2 - x1, rois_h = y2 - y1;
    const float rois_inner_w = rois_w / context_ratio, rois_inner_h = rois_h / context_ratio;
    const float rois_outer_w = rois_w * context_ratio, rois_outer_h = rois_h * context_ratio;
    const float inner_residual_w = rois_w - rois_inner_w, inner_residual_h = rois_h - rois_inner_h;
    const float outer_residual_w = rois_outer_w - rois_w, outer_residual_h = rois_outer_h - rois_h;
    const float xmax = (float)(1.0 * width / spatial_scale), ymax = (float)(1.0 * height / spatial_scale);
#define CLAMPF(v, hi) fminf(fmaxf((v), 0.f), (hi))
    const float x1_inner = CLAMPF(x1 + inner_residual_w / 2, xmax), y1_inner = CLAMPF(y1 + inner_residual_h / 2, ymax);
    const float x2_inner = CLAMPF(x2 - inner_residual_w / 2, xmax), y2_inner = CLAMPF(y2 - inner_residual_h / 2, ymax);
    const float x1_outer = CLAMPF(x1 - outer_residual_w / 2, xmax), y1_outer = CLAMPF(y1 - outer_residual_h / 2, ymax);
    const float x2_outer = CLAMPF(x2 + outer_residual_w / 2, xmax), y2_outer = CLAMPF(y2 + outer_residual_h / 2, ymax);
#undef CLAMPF
    const int sw = (int)roundf(x1 * spatial_scale), sh = (int)roundf(y1 * spatial_scale);
    const int ew = (int)roundf(x2 * spatial_scale), eh = (int)roundf(y2 * spatial_scale);
    const int sw_in = (int)roundf(x1_inner * spatial_scale), sh_in = (int)roundf(y1_inner * spatial_scale);
    const int ew_in = (int)roundf(x2_inner * spatial_scale), eh_in = (int)roundf(y2_inner * spatial_scale);
    const int sw_out = (int)roundf(x1_outer * spatial_scale), sh_out = (int)roundf(y1_outer * spatial_scale);
    const int ew_out = (int)roundf(x2_outer * spatial_scale), eh_out = (int)roundf(y2_outer * spatial_scale);
    for (int c = 0; c < channels; ++c) {
      const float* plane = input + ((long)b * channels + c) * height * width;
      for (int ph = 0; ph < pooled_height; ++ph)
        for (int pw = 0; pw < pooled_width; ++pw) {
          const long index = (((long)n * channels + c) * pooled_height + ph) * pooled_width + pw;
          {
            const int roi_width = imax(ew - sw + 1, 1), roi_height = imax(eh - sh + 1, 1);
            const float bin_h = (float)roi_height / (float)pooled_height, bin_w = (float)roi_width / (float)pooled_width;
            const int hstart = imin(imax((int)floorf((float)ph * bin_h) + sh, 0), height);
            const int hend = imin(imax((int)ceilf((float)(ph + 1) * bin_h) + sh, 0), height);
            const int wstart = imin(imax((int)floorf((float)pw * bin_w) + sw, 0), width);
            const int wend = imin(imax((int)ceilf((float)(pw + 1) * bin_w) + sw, 0), width);
            float maxval = 0, maxval_F = 0;
            int maxidx = -1, maxidx_F = -1;
            for (int h = hstart; h < hend; ++h)
              for (int w = wstart; w < wend; ++w) {
                const int ii = h * width + w;
                if (plane[ii] > maxval) { maxval = plane[ii]; maxidx = ii; }
                if (h > sh_in && h < eh_in && w > sw_in && w < ew_in) continue;
                if (plane[ii] > maxval_F) { maxval_F = plane[ii]; maxidx_F = ii; }
              }
            output[index] = maxval;
            argmax_data[index] = maxidx;
            output[index + part] = maxval_F;
            argmax_data[index + part] = maxidx_F;
          }
          {
            const int roi_width = imax(ew_out - sw_out + 1, 1), roi_height = imax(eh_out - sh_out + 1, 1);
            const float bin_h = (float)roi_height / (float)pooled_height, bin_w = (float)roi_width / (float)pooled_width;
            const int hstart = imin(imax((int)floorf((float)ph * bin_h) + sh_out, 0), height);
            const int hend = imin(imax((int)ceilf((float)(ph + 1) * bin_h) + sh_out, 0), height);
            const int wstart = imin(imax((int)floorf((float)pw * bin_w) + sw_out, 0), width);
            const int wend = imin(imax((int)ceilf((float)(pw + 1) * bin_w) + sw_out, 0), width);
            float maxval = 0;
            int maxidx = -1;
            for (int h = hstart; h < hend; ++h) {
              const int in_h = h > sh && h < eh;
              for (int w = wstart; w < wend; ++w) {
                if (in_h && w > sw && w < ew) continue;
                const int ii = h * width + w;
                if (plane[ii] > maxval) { maxval = plane[ii]; maxidx = ii; }
              }
            }
            output[index + 2 * part] = maxval;
            argmax_data[index + 2 * part] = maxidx;
          }
        }
    }
  }
}

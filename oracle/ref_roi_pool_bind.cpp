// ORACLE -- test infrastructure only.
// C-ABI shim around the REFERENCE's own CPU RoIPool, compiled from the reference sources
// where they lie (/root/reference/wsovod/layers/ROILoopPool/ROILoopPool_cpu.cpp; the
// header's Python-facing dispatcher raises on CPU, ROILoopPool.h:62,95, so the *_cpu
// functions are bound directly).  Built by oracle/build.py into oracle/_ref/ (git-ignored);
// used to pin oracle/roi_ops_ref.c and, optionally, as the `reference` CPU baseline.
#include <ATen/ATen.h>

#include "ROILoopPool/ROILoopPool.h"

extern "C" {

void ref_roi_pool_forward(const float* input, int N, int C, int H, int W, const float* rois, int R,
                          float spatial_scale, int ph, int pw, float* out, int* argmax) {
  at::Tensor in_t = at::from_blob((void*)input, {N, C, H, W}, at::kFloat);
  at::Tensor rois_t = at::from_blob((void*)rois, {R, 5}, at::kFloat);
  auto res = wsovod::ROILoopPool_forward_cpu(in_t, rois_t, spatial_scale, ph, pw);
  at::Tensor o = std::get<0>(res).contiguous();
  at::Tensor a = std::get<1>(res).contiguous();
  memcpy(out, o.data_ptr<float>(), sizeof(float) * o.numel());
  memcpy(argmax, a.data_ptr<int>(), sizeof(int) * a.numel());
}

void ref_roi_pool_backward(const float* grad, const float* rois, const int* argmax, int R,
                           float spatial_scale, int ph, int pw, int N, int C, int H, int W,
                           float* grad_input) {
  at::Tensor g_t = at::from_blob((void*)grad, {R, C, ph, pw}, at::kFloat);
  at::Tensor rois_t = at::from_blob((void*)rois, {R, 5}, at::kFloat);
  at::Tensor a_t = at::from_blob((void*)argmax, {R, C, ph, pw}, at::kInt);
  at::Tensor gi = wsovod::ROILoopPool_backward_cpu(g_t, rois_t, a_t, spatial_scale, ph, pw, N, C, H, W).contiguous();
  memcpy(grad_input, gi.data_ptr<float>(), sizeof(float) * gi.numel());
}
}

/* ORACLE -- test infrastructure only (never linked into the product library).
 *
 * Plain-C restatement of the greedy NMS the reference calls through detectron2.layers.batched_nms ->
 * torchvision.ops.nms (torchvision 0.13.1, README.md:34 -- UN-VENDORED, absent from /root/reference and not
 * installable here, so this function is "parity unpinned": it restates the published CPU kernel
 * torchvision/csrc/ops/cpu/nms_kernel.cpp and is anchored on the reference's call sites
 * proposal_utils.py:123 and fast_rcnn_open_vocabulary.py:176).
 *
 * Boxes come in already sorted by descending score inside each segment (the caller sorts with torch.sort,
 * as torchvision does); boxes of different segments never interact -- the exact per-category form of
 * batched_nms (torchvision's `_batched_nms_vanilla`; its coordinate-offset form is an approximation of it).
 * Compile with -ffp-contract=off: the comparison `ovr > thr` must see the same fp32 roundings as the kernel. */
#include <stdlib.h>

static float fmax2(float a, float b) { return a > b ? a : b; }
static float fmin2(float a, float b) { return a < b ? a : b; }

/* keep_idx: per segment, kept positions relative to the segment start, written from seg_offsets[g];
 * keep_count[g] = how many.  valid (optional): 0 = filtered out.  max_keep <= 0: unlimited. */
void nms_segments(const float* boxes, const int* seg_offsets, const unsigned char* valid, int G, float thr, int max_keep,
                  int* keep_idx, int* keep_count) {
  for (int g = 0; g < G; ++g) {
    const int s0 = seg_offsets[g], n = seg_offsets[g + 1] - s0;
    unsigned char* suppressed = (unsigned char*)calloc(n > 0 ? n : 1, 1);
    int kept = 0;
    for (int i = 0; i < n && (max_keep <= 0 || kept < max_keep); ++i) {
      if (suppressed[i] || (valid && !valid[s0 + i])) continue;
      keep_idx[s0 + kept++] = i;
      const float* a = boxes + 4 * (long)(s0 + i);
      const float iarea = (a[2] - a[0]) * (a[3] - a[1]);
      for (int j = i + 1; j < n; ++j) {
        if (suppressed[j]) continue;
        const float* b = boxes + 4 * (long)(s0 + j);
        const float xx1 = fmax2(a[0], b[0]), yy1 = fmax2(a[1], b[1]);
        const float xx2 = fmin2(a[2], b[2]), yy2 = fmin2(a[3], b[3]);
        const float w = fmax2(0.f, xx2 - xx1), h = fmax2(0.f, yy2 - yy1);
        const float inter = w * h;
        const float barea = (b[2] - b[0]) * (b[3] - b[1]);
        const float ovr = inter / (iarea + barea - inter);
        if (ovr > thr) suppressed[j] = 1;
      }
    }
    keep_count[g] = kept;
    free(suppressed);
  }
}

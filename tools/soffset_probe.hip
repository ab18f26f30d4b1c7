// Does the scalar `soffset` of a raw buffer load take part in the hardware range check on gfx950?  (It does: lanes whose
// voffset + soffset reach num_records read zeros; a voffset of 2^31 stays out of range whatever the soffset adds.)  The
// lean forms of gemm8.hip / gemm_tn8.hip rely on it: a DMA instruction = per-lane constant offset + scalar K offset.
//   hipcc -O2 --offload-arch=gfx950 tools/soffset_probe.hip -o tools/soffset_probe && gpurun -- ./tools/soffset_probe
// Measured (round 5): records=1024: voff 0 / soff 1016 -> 2 values then zeros; voff 8 / soff 1016 -> zeros;
// voff 0x80000000 / soff 16 -> zeros.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const int* buf, int nbytes, int voff, int soff, int* out) {
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)buf, 0, nbytes, 0x00020000);
  out[threadIdx.x] = __builtin_amdgcn_raw_buffer_load_b32(r, voff + 4 * threadIdx.x, soff, 0);
}
int main() {
  int *buf, *out;
  if (hipMalloc(&buf, 4096) != hipSuccess || hipMalloc(&out, 256) != hipSuccess) return 1;
  int h[1024];
  for (int i = 0; i < 1024; ++i) h[i] = 1000 + i;
  (void)hipMemcpy(buf, h, 4096, hipMemcpyHostToDevice);
  const int cases[][3] = {{1024, 0, 0}, {1024, 1016, 0}, {1024, 0, 1016}, {1024, 512, 504}, {1024, 8, 1016},
                          {1024, 2048, 0}, {1024, 0, 2048}, {1024, (int)0x80000000, 16}};
  for (auto& c : cases) {
    hipLaunchKernelGGL(k, dim3(1), dim3(4), 0, 0, buf, c[0], c[1], c[2], out);
    int o[4];
    (void)hipMemcpy(o, out, 16, hipMemcpyDeviceToHost);
    printf("records=%d voff=%d soff=%d -> %d %d %d %d\n", c[0], c[1], c[2], o[0], o[1], o[2], o[3]);
  }
  return 0;
}

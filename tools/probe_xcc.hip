// Probe: which XCD does workgroup b land on, across consecutive launches of different grid sizes?
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(int* out) {
  if (threadIdx.x == 0) {
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    out[blockIdx.x] = (int)(x & 0xf);
  }
}
int main() {
  int* d; (void)hipMalloc(&d, 4096 * 4);
  int h[4096];
  int grids[] = {256, 256, 1024, 100, 256, 7, 256, 264, 256, 2048, 256};
  for (int gi = 0; gi < (int)(sizeof(grids) / sizeof(int)); ++gi) {
    int g = grids[gi];
    hipLaunchKernelGGL(k, dim3(g), dim3(256), 0, 0, d);
    (void)hipMemcpy(h, d, g * 4, hipMemcpyDeviceToHost);
    int rr = 1;
    for (int b = 0; b < g; ++b) if (h[b] != (h[0] + b) % 8) rr = 0;
    printf("grid %4d: block0 -> XCC %d ; first 16:", g, h[0]);
    for (int b = 0; b < 16 && b < g; ++b) printf(" %d", h[b]);
    printf(" ; strict round-robin: %s\n", rr ? "yes" : "NO");
  }
  // back-to-back without host sync in between
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(k, dim3(256), dim3(256), 0, 0, d);
    hipLaunchKernelGGL(k, dim3(100), dim3(256), 0, 0, d + 1024);
    hipLaunchKernelGGL(k, dim3(256), dim3(256), 0, 0, d + 2048);
    (void)hipMemcpy(h, d, 4096 * 4, hipMemcpyDeviceToHost);
    printf("async seq: 256-grid block0 XCC %d, then 100-grid block0 XCC %d, then 256-grid block0 XCC %d\n", h[0], h[1024], h[2048]);
  }
  return 0;
}

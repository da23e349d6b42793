// Probe: what does ds_read_b64_tr_b16 return?  LDS holds u16 value = row*64 + col for a [64][64] tile (128-B rows).
// Experiment A: lane L supplies the address of (row L, col 0).   -> shows which lanes' addresses feed lane X.
// Experiment B: my assumed usage: lane i=4q+p in group g supplies (row 8g+q, col 4p).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) short s16x4;
__global__ void k(int mode, unsigned short* out) {
  __shared__ __attribute__((aligned(16))) unsigned short lds[64 * 64];
  int t = threadIdx.x;
  for (int i = t; i < 4096; i += 64) lds[i] = (unsigned short)i;
  __syncthreads();
  int row, col;
  if (mode == 0) { row = t; col = 0; }
  else { int g = t >> 4, i = t & 15, q = i >> 2, p = i & 3; row = 8 * g + q; col = 4 * p; }
  auto p = (__attribute__((address_space(3))) s16x4*)(lds + row * 64 + col);
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(p);
  for (int e = 0; e < 4; ++e) out[t * 4 + e] = (unsigned short)v[e];
}
int main() {
  unsigned short* d; hipMalloc(&d, 64 * 4 * 2);
  unsigned short h[256];
  for (int mode = 0; mode < 2; ++mode) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, mode, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("mode %d\n", mode);
    for (int t = 0; t < 64; ++t) {
      printf("lane %2d:", t);
      for (int e = 0; e < 4; ++e) printf(" (r%2d,c%2d)", h[t * 4 + e] / 64, h[t * 4 + e] % 64);
      printf("\n");
    }
  }
  return 0;
}

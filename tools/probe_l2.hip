// Probe: does data written by kernel 1 on XCD x stay in x's L2 for kernel 2?  Compare a consumer that reads with the
// SAME block->chunk affinity as the producer against one shifted by one XCD, for several per-XCD footprints.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) float f4;
__global__ __launch_bounds__(256) void producer(f4* buf, int chunk_f4, int nper, int shift) {
  const int x = (blockIdx.x % 8 + shift) % 8, i = blockIdx.x / 8;
  f4* p = buf + ((size_t)x * nper + i) * chunk_f4;
  for (int j = threadIdx.x; j < chunk_f4; j += 256) p[j] = f4{1.f, 2.f, 3.f, (float)j};
}
__global__ __launch_bounds__(256) void consumer(const f4* buf, int chunk_f4, int nper, int shift, float* out) {
  const int x = (blockIdx.x % 8 + shift) % 8, i = blockIdx.x / 8;
  const f4* p = buf + ((size_t)x * nper + i) * chunk_f4;
  f4 a = {0, 0, 0, 0};
  for (int j = threadIdx.x; j < chunk_f4; j += 256) a += p[j];
  if (a[0] + a[1] + a[2] + a[3] == -1.f) out[0] = 1.f;
}
int main() {
  float* out; (void)hipMalloc(&out, 4);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int nper = 32;  // 32 blocks per XCD = one per CU
  for (int per_xcd_kb : {512, 1024, 2048, 4096, 8192}) {
    const int chunk_bytes = per_xcd_kb * 1024 / nper, chunk_f4 = chunk_bytes / 16;
    f4* buf; (void)hipMalloc(&buf, (size_t)8 * nper * chunk_bytes);
    for (int shift = 0; shift < 2; ++shift) {
      float best = 1e9f;
      for (int rep = 0; rep < 10; ++rep) {
        hipLaunchKernelGGL(producer, dim3(8 * nper), dim3(256), 0, 0, buf, chunk_f4, nper, 0);
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL(consumer, dim3(8 * nper), dim3(256), 0, 0, buf, chunk_f4, nper, shift, out);
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
      }
      printf("per-XCD %5d KB  consumer reads %s data: %7.1f us  (%6.1f GB/s per CU, %5.2f TB/s total)\n", per_xcd_kb,
             shift ? "ANOTHER XCD's" : "its OWN XCD's", best * 1e3, chunk_bytes / (best * 1e-3) / 1e9, 8.0 * nper * chunk_bytes / (best * 1e-3) / 1e12);
    }
    (void)hipFree(buf);
  }
  return 0;
}

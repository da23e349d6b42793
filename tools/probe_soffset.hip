// Probe: how does the SGPR offset (`soffset`) of a raw buffer load take part in the descriptor's range check on gfx950?
// Round 1 recorded a GPU memory access fault (address = first byte past a 2-MiB hipMalloc) from tools/probe_fill.hip run
// with a 2-MiB window: its wave-uniform step offset (up to ~5 MiB) went into `soffset` of raw_ptr_buffer_load_lds while
// num_records was 2 MiB.  This probe never leaves its allocation: a 16-MiB buffer holds word i = i, the descriptor covers
// only the first NR bytes, and a table of (voffset, soffset) pairs is loaded through the register form (4 B) and the
// LDS-DMA form (4 B and 16 B per lane); a clamped (out-of-range) load returns 0, an unchecked one the word at that offset.
// hipcc --offload-arch=gfx950 -O3 tools/probe_soffset.hip -o /tmp/probe_soffset && /tmp/probe_soffset
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

__global__ void fill(unsigned* p, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = (unsigned)i;
}

struct Case { unsigned nr, voff, soff; };

__global__ __launch_bounds__(64) void probe(const unsigned* base, const Case* cases, int n, unsigned* out) {
  __shared__ __attribute__((aligned(16))) unsigned lds[64 * 4 * 2];
  const int lane = threadIdx.x;
  for (int c = 0; c < n; ++c) {
    const unsigned nr = __builtin_amdgcn_readfirstlane(cases[c].nr), vo = cases[c].voff, so = __builtin_amdgcn_readfirstlane(cases[c].soff);
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>(base), 0, (int)nr, 0x00020000);
    const unsigned a = __builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)(vo + 4 * lane), (int)so, 0);
    for (int k = 0; k < 8; ++k) lds[lane + 64 * k] = 0xdeadbeefu;
    __syncthreads();
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds, 4, (int)(vo + 4 * lane), (int)so, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(lds + 256), 16, (int)(vo + 16 * lane), (int)so, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (lane == 0) { out[4 * c] = a; out[4 * c + 1] = lds[0]; out[4 * c + 2] = lds[256]; out[4 * c + 3] = lds[256 + 4 * 63]; }
    __syncthreads();
  }
}

int main() {
  unsigned* buf; unsigned* out; Case* dc;
  const size_t bytes = 16 << 20;
  const unsigned MiB = 1u << 20;
  Case cases[] = {
      {1 * MiB, 0, 0},                    // in range
      {1 * MiB, 2 * MiB, 0},              // voffset beyond num_records
      {1 * MiB, 0, 2 * MiB},              // soffset beyond num_records
      {1 * MiB, 0, 1 * MiB},              // soffset == num_records
      {1 * MiB, 0, 1 * MiB - 4096},       // soffset just inside, voffset keeps the sum inside
      {1 * MiB, 8192, 1 * MiB - 4096},    // each inside, the SUM beyond num_records
      {1 * MiB, 1 * MiB - 4096, 8192},    // same, roles swapped
      {2 * MiB, 0, 3 * MiB},              // the round-1 probe's case: window 2 MiB, step offset 3 MiB
      {2 * MiB, 0, 5 * MiB},              //                                         ... 5 MiB
      {2 * MiB, 0, 2 * MiB + 1024},       // just past the window
      {2 * MiB, 1024, 2 * MiB - 1024},    // sum == num_records exactly (first byte past)
      {2 * MiB, 0, 2 * MiB - 1024},       // the 16-B/lane form: lanes 0..63 cover 1 KiB ending exactly at num_records
      {2 * MiB, 0, 2 * MiB - 512},        // the 16-B/lane form straddles the end: its upper lanes are out of range
  };
  const int n = sizeof(cases) / sizeof(cases[0]);
  if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&out, n * 16) != hipSuccess || hipMalloc(&dc, sizeof(cases)) != hipSuccess) return 2;
  (void)hipMemcpy(dc, cases, sizeof(cases), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(fill, dim3(256), dim3(256), 0, 0, buf, bytes / 4);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, buf, dc, n, out);
  unsigned h[64 * 4];
  if (hipMemcpy(h, out, n * 16, hipMemcpyDeviceToHost) != hipSuccess) return 3;
  printf("16-MiB allocation, word i holds i; a clamped load reads 0.  'want' = the word an UNCHECKED load would return.\n");
  printf("%10s %10s %10s | %10s | %12s %12s %14s %16s\n", "num_rec", "voffset", "soffset", "want", "reg 4B", "lds 4B", "lds 16B lane0", "lds 16B lane63");
  int unchecked = 0;
  for (int c = 0; c < n; ++c) {
    const unsigned want = (cases[c].voff + cases[c].soff) / 4;
    const bool beyond = (uint64_t)cases[c].voff + cases[c].soff >= cases[c].nr;
    printf("%10u %10u %10u | %10u | %12u %12u %14u %16u   %s\n", cases[c].nr, cases[c].voff, cases[c].soff, want, h[4 * c], h[4 * c + 1], h[4 * c + 2],
           h[4 * c + 3], beyond ? (h[4 * c] == 0 && h[4 * c + 1] == 0 && h[4 * c + 2] == 0 ? "beyond: clamped" : "beyond: NOT CLAMPED") : "inside");
    if (beyond && (h[4 * c] || h[4 * c + 1] || h[4 * c + 2])) ++unchecked;
  }
  printf("verdict: %d of the out-of-range cases were NOT clamped\n", unchecked);
  return 0;
}

// What does a grid-wide barrier cost inside one launch (256 workgroups, one per CU)?  Variants: (0) one counter, every workgroup's thread 0
// adds 1 and spins until the counter reaches round x 256; (1) hierarchical: one counter per XCD (32 arrivals), the last arrival of an XCD
// adds to the global counter, everyone spins on the global one; (2) as (1) but every workgroup spins on its own XCD's release word, which
// the XCD's last-arriver sets after seeing the global count (fewer pollers per address).  All spins bounded; a timeout sets a word that
// makes everyone leave.  hipcc --offload-arch=gfx950 -O3 tools/probe_gridbar.hip -o /tmp/gb && /tmp/gb
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(1))) unsigned gu32;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
__device__ __forceinline__ unsigned ld(const unsigned* p) { return __hip_atomic_load((gu32*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st(unsigned* p, unsigned v) { __hip_atomic_store((gu32*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned add(unsigned* p, unsigned v) { return __hip_atomic_fetch_add((gu32*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ __forceinline__ bool spin_ge(const unsigned* p, unsigned target, unsigned* status) {
  for (unsigned s = 0; s < 4000000u; ++s) {
    if (ld(p) >= target) return true;
    if ((s & 255u) == 255u && ld(status) != 0) return false;
    __builtin_amdgcn_s_sleep(1);
  }
  st(status, 1u);
  return false;
}

template <int MODE>
__global__ __launch_bounds__(256) void gridbar(unsigned* ctr /* [0] global, [32*(1+x)] per XCD arrive, [32*(9+x)] per XCD release */, unsigned* status,
                                               int rounds, unsigned long long* t_out, float* sink) {
  const int xcd = blockIdx.x & 7, nx = gridDim.x >> 3;
  __shared__ int give_up;
  if (threadIdx.x == 0) give_up = 0;
  __syncthreads();
  float acc = threadIdx.x;
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  for (int r = 1; r <= rounds; ++r) {
    acc = acc * 1.0001f + 1.f;   // a token of work between barriers
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
      bool ok;
      if (MODE == 0) {
        add(ctr, 1u);
        ok = spin_ge(ctr, (unsigned)r * gridDim.x, status);
      } else {
        const unsigned prev = add(ctr + 32 * (1 + xcd), 1u);
        const bool last = (prev + 1u == (unsigned)r * nx);
        if (last) add(ctr, 1u);
        if (MODE == 1) ok = spin_ge(ctr, (unsigned)r * 8u, status);
        else {
          if (last) { ok = spin_ge(ctr, (unsigned)r * 8u, status); st(ctr + 32 * (9 + xcd), (unsigned)r); }
          else ok = spin_ge(ctr + 32 * (9 + xcd), (unsigned)r, status);
        }
      }
      if (!ok) give_up = 1;
    }
    __syncthreads();
    if (give_up) break;
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) t_out[blockIdx.x] = t1 - t0;
  if (acc == 12345.678f) sink[0] = acc;
}

int main() {
  const int rounds = 500;
  unsigned *ctr, *status; unsigned long long* t_out; float* sink;
  CK(hipMalloc(&ctr, 32 * 20 * 4)); CK(hipMalloc(&status, 64)); CK(hipMalloc(&t_out, 256 * 8)); CK(hipMalloc(&sink, 4));
  std::vector<unsigned long long> t(256);
  for (int mode = 0; mode < 3; ++mode) {
    CK(hipMemset(ctr, 0, 32 * 20 * 4)); CK(hipMemset(status, 0, 64));
    if (mode == 0) hipLaunchKernelGGL(gridbar<0>, dim3(256), dim3(256), 0, 0, ctr, status, rounds, t_out, sink);
    if (mode == 1) hipLaunchKernelGGL(gridbar<1>, dim3(256), dim3(256), 0, 0, ctr, status, rounds, t_out, sink);
    if (mode == 2) hipLaunchKernelGGL(gridbar<2>, dim3(256), dim3(256), 0, 0, ctr, status, rounds, t_out, sink);
    CK(hipDeviceSynchronize());
    unsigned stw; CK(hipMemcpy(&stw, status, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(t.data(), t_out, 256 * 8, hipMemcpyDeviceToHost));
    double mean = 0; for (auto v : t) mean += (double)v; mean /= 256;
    printf("grid barrier over 256 workgroups, variant %d (%s): %.2f us per barrier   timeout word %u\n", mode,
           mode == 0 ? "one counter" : mode == 1 ? "per-XCD arrive, global spin" : "per-XCD arrive and release", mean / 100.0 / rounds, stw);
  }
  return 0;
}

// Probe: how fast can ONE CU pull operand tiles from L2 into LDS (or registers), and what does it depend on?
// Every GEMM main loop of this library sits at ~46 GB/s of LDS fill per CU (0.65-0.8 us per 32-KiB K tile) whatever the
// ring depth; this measures the fill alone (no MFMA, no consumer) for: the number of issuing waves, bytes in flight, the
// access shape (128-B row segments at a 1-KiB pitch, as a K-contiguous tile, or fully contiguous 1-KiB pieces), LDS-DMA
// against plain register loads, and how many CUs stream at once.  hipcc --offload-arch=gfx950 -O3 tools/probe_fill.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(4))) unsigned int u4;

// mode 0: LDS-DMA (buffer_load_dwordx4 ... lds); mode 1: global_load_dwordx4 into registers (then dropped)
// shape 0: K-contiguous tile rows: lane l of piece p reads 16 B at row (8p + l/8), chunk l%8, pitch `pitch` bytes
// shape 1: contiguous 1-KiB pieces
template <int MODE, int SHAPE>
__global__ __launch_bounds__(1024) void fill_kernel(const char* base, size_t span_bytes, int pitch, int pieces_per_step, int steps,
                                                    int inflight_steps, unsigned long long* out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
  __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(base), 0, (int)span_bytes, 0x00020000);
  // this workgroup streams tiles from its own window of the source (all windows together stay L2 / MALL resident)
  const size_t wg_off = ((size_t)blockIdx.x * 2654435761u) % (span_bytes / 2) & ~(size_t)1023;
  const int ppw = pieces_per_step / nwaves;  // pieces per wave per step (host makes it whole)
  u4 sink = {0, 0, 0, 0};
  const unsigned long long t0 = wall_clock64();
  for (int s = 0; s < steps; ++s) {
    // keep at most inflight_steps steps outstanding: wait until only (inflight_steps-1)*ppw of this wave's loads remain
    if (s >= inflight_steps) {
      const int keep = (inflight_steps - 1) * ppw;
      if (keep >= 24) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
      else if (keep >= 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
      else if (keep >= 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
      else if (keep >= 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else if (keep >= 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else if (keep >= 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
      else if (keep >= 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    // address arithmetic kept to one scalar add per step and one per piece (a 64-bit modulo here made the first version
    // of this probe VALU-bound at ~10 GB/s per wave)
    const unsigned step_off = (unsigned)wg_off + (unsigned)(s & 63) * (SHAPE == 0 ? 128u : (unsigned)pieces_per_step * 1024u);
    const unsigned lane_off = SHAPE == 0 ? (unsigned)(lane >> 3) * (unsigned)pitch + (unsigned)(lane & 7) * 16u : (unsigned)lane * 16u;
    const unsigned piece_stride = SHAPE == 0 ? 8u * (unsigned)pitch : 1024u;
    int slot = s - (s / inflight_steps) * inflight_steps;
    for (int i = 0; i < ppw; ++i) {
      const int p = wave * ppw + i;
      const unsigned soff = step_off + (unsigned)p * piece_stride;  // wave-uniform
      if (MODE == 0 || (MODE == 2 && (wave & 1) == 0)) {
        char* dst = smem + (slot * pieces_per_step + p) * 1024;
        // offsets wrap inside the window and travel in the VECTOR offset: `soffset` is not covered by the descriptor's
        // range check (tools/probe_soffset.hip), which is how the first version of this probe faulted with a 2-MiB window
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)dst, 16,
                                                 (int)((soff + lane_off) & (unsigned)(span_bytes - 1) & ~15u), 0, 0, 0);
      } else {
        u4 v = *reinterpret_cast<const u4*>(base + ((size_t)(soff + lane_off) & (span_bytes - 1) & ~(size_t)15));  // span is 2^k: in bounds
        asm volatile("" : "+v"(v));
        sink |= v;
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const unsigned long long t1 = wall_clock64();
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
  if (sink[0] == 0x12345678u && sink[1] == 1u) out[0] = 0;
}

template <int MODE, int SHAPE>
double run(const char* src, size_t span, int nblocks, int nwaves, int pieces_per_step, int inflight, int pitch, unsigned long long* dout) {
  const int steps = 512;
  const int lds = MODE != 1 ? inflight * pieces_per_step * 1024 : 0;
  auto k = fill_kernel<MODE, SHAPE>;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  double best = 0;
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(k, dim3(nblocks), dim3(nwaves * 64), lds, 0, src, span, pitch, pieces_per_step, steps, inflight, dout);
    (void)hipDeviceSynchronize();
    static unsigned long long h[1024];
    (void)hipMemcpy(h, dout, nblocks * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    double t = 0;
    for (int b = 0; b < nblocks; ++b) t += h[b] * 0.01;  // us
    t /= nblocks;
    const double gbs = (double)steps * pieces_per_step * 1024 / (t * 1e-6) / 1e9;
    if (gbs > best) best = gbs;
  }
  return best;
}

int main(int argc, char** argv) {
  // source window (MiB, a power of two >= 2; default 64 = Infinity-Cache resident, mostly L2 misses)
  size_t mib = argc > 1 ? (size_t)atoi(argv[1]) : 64;
  if (mib < 2 || (mib & (mib - 1))) { fprintf(stderr, "window must be a power of two >= 2 MiB\n"); return 2; }
  const size_t span = mib << 20;
  char* src; (void)hipMalloc(&src, span); (void)hipMemset(src, 1, span);
  unsigned long long* dout; (void)hipMalloc(&dout, 1024 * sizeof(unsigned long long));
  printf("LDS fill / register load rate per CU (GB/s), 32 KiB per step unless noted; source window %zu MiB\n", span >> 20);
  for (int nblocks : {1, 256}) {
    printf("-- %d workgroup(s), one per CU\n", nblocks);
    for (int nwaves : {1, 2, 4, 8, 16}) {
      printf("  %2d issuing waves:", nwaves);
      for (int inflight : {2, 3, 4}) {
        const double a = run<0, 0>(src, span, nblocks, nwaves, 32, inflight, 1024, dout);
        const double b = run<0, 1>(src, span, nblocks, nwaves, 32, inflight, 1024, dout);
        printf("  DMA %d stages in flight: rows %5.1f contiguous %5.1f |", inflight, a, b);
      }
      const double c = run<1, 0>(src, span, nblocks, nwaves, 32, 2, 1024, dout), d = run<1, 1>(src, span, nblocks, nwaves, 32, 2, 1024, dout);
      printf("  registers (2 steps): rows %5.1f contiguous %5.1f\n", c, d);
    }
  }
  // are the two paths additive?  16 waves, even ones fill LDS by DMA, odd ones load into registers
  for (int nblocks : {1, 256})
    printf("mixed, 16 waves (8 DMA + 8 register), %3d workgroups: rows %5.1f contiguous %5.1f GB/s per CU  (8 DMA waves alone: see above)\n", nblocks,
           run<2, 0>(src, span, nblocks, 16, 32, 3, 1024, dout), run<2, 1>(src, span, nblocks, 16, 32, 3, 1024, dout));
  // the 256-tile kernel's step: 64 KiB, 8 waves, 2 stages
  printf("64-KiB steps, 8 waves, 2 in flight, 256 workgroups: rows %5.1f contiguous %5.1f GB/s per CU\n",
         run<0, 0>(src, span, 256, 8, 64, 2, 1024, dout), run<0, 1>(src, span, 256, 8, 64, 2, 1024, dout));
  return 0;
}

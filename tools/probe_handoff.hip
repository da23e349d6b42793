// What does a hand-off between the CUs of a small CLUSTER cost inside one launch?  (a cluster of 4 CUs owning
// one sentence through a whole stack exchanges slices of 1 KB .. 128 KB five times per layer.)  256 workgroups of 512 threads, one
// per CU, in clusters of 4; per round every workgroup publishes a slice (16-byte sc1 stores, every wave drains, one lane stores the
// flag: MI355X guide, Guideline 16 R1) and reads the three peers' slices (one wave polls the peers' flags relaxed, barrier, every
// load sc1).  Slices are double-buffered by round parity: a peer's flag for round r + 1 proves it is done reading round r.  Every
// spin is bounded; a timeout sets a word that makes every workgroup leave.  Checks every word it reads.
//   hipcc --offload-arch=gfx950 -O3 tools/probe_handoff.hip -o /tmp/ph && /tmp/ph
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((address_space(1))) unsigned gu32;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__device__ __forceinline__ unsigned ld_rlx(const unsigned* p) { return __hip_atomic_load((gu32*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_rlx(unsigned* p, unsigned v) { __hip_atomic_store((gu32*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// mode 0: cluster = 4 consecutive workgroup ids (four different XCDs under round-robin placement); 1: same id % 8 (one XCD)
// LOCAL: the cluster shares one XCD's L2, which is coherent for its own CUs: plain stores (they stay in L2), loads that only skip the
// CU's L1 (sc0) -- correct ONLY if the four workgroups really are on one XCD (checked with the XCC_ID register)
template <bool LOCAL>
__global__ __launch_bounds__(512) void handoff(char* slices, unsigned* flags, unsigned* status, int slice_bytes, int rounds, int mode,
                                               unsigned long long* t_out, unsigned* xcc) {
  const int b = blockIdx.x;
  int cluster, member;
  if (mode == 0) { cluster = b >> 2; member = b & 3; }
  else           { const int x = b & 7, q = b >> 3; cluster = x * 8 + (q >> 2); member = q & 3; }
  const int wg = cluster * 4 + member;                         // slot of this workgroup's slices and flag
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __shared__ int give_up;
  if (threadIdx.x == 0) give_up = 0;
  __syncthreads();
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(slices, 0, 0x7fffffff, 0x00020000);
  const int chunks = slice_bytes / 16;
  unsigned bad = 0;
  if (threadIdx.x == 0) {
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    xcc[wg] = x & 15u;
  }
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  for (int r = 0; r < rounds; ++r) {
    const unsigned epoch = (unsigned)r + 1;
    const int64_t mine = ((int64_t)(r & 1) * 1024 + wg) * slice_bytes;
    for (int c = threadIdx.x; c < chunks; c += 512) {
      const u32x4 v = {epoch, (unsigned)wg, (unsigned)c, epoch ^ (unsigned)c};
      __builtin_amdgcn_raw_buffer_store_b128(v, rs, (int)(mine + (int64_t)c * 16), 0, LOCAL ? 0 : 16);  // aux 16 = sc1: write-through
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every storing wave drains
    __syncthreads();
    if (threadIdx.x == 0) st_rlx(flags + wg * 32, epoch);   // one lane publishes (flags on lines of their own)
    if (wave == 0) {                                        // one wave polls the three peers, relaxed, bounded
      bool ok = false;
      for (unsigned spins = 0; spins < 4000000u; ++spins) {
        bool all = true;
        if (lane < 4 && lane != member) all = ld_rlx(flags + (cluster * 4 + lane) * 32) >= epoch;
        if (__all(all)) { ok = true; break; }
        if (ld_rlx(status) != 0) break;
        __builtin_amdgcn_s_sleep(2);
      }
      if (!ok && lane == 0) { st_rlx(status, 0x1000u + (unsigned)r); give_up = 1; }
    }
    __syncthreads();
    if (give_up) break;
    // every load of handed-off bytes is an sc1 load to registers (no acquire needed: Guideline 16, the all-sc1 form)
    for (int p = 1; p < 4; ++p) {
      const int peer = cluster * 4 + ((member + p) & 3);
      const int64_t theirs = ((int64_t)(r & 1) * 1024 + peer) * slice_bytes;
      for (int c = threadIdx.x; c < chunks; c += 512) {
        const u32x4 v = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(theirs + (int64_t)c * 16), 0, LOCAL ? 1 : 16));
        bad |= (v[0] != epoch) | (v[1] != (unsigned)peer) | (v[2] != (unsigned)c) | (v[3] != (epoch ^ (unsigned)c));
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
  if (__any(bad != 0) && lane == 0) atomicOr(status + 1, 1u);
  if (threadIdx.x == 0) t_out[b] = t1 - t0;
}

int main() {
  const int rounds = 400;
  char* slices; unsigned *flags, *status; unsigned long long* t_out;
  CK(hipMalloc(&slices, 2ll * 1024 * 131072)); CK(hipMalloc(&flags, 1024 * 32 * 4)); CK(hipMalloc(&status, 64)); CK(hipMalloc(&t_out, 256 * 8));
  std::vector<unsigned long long> t(256);
  unsigned* xcc; CK(hipMalloc(&xcc, 1024 * 4));
  std::vector<unsigned> hx(256);
  for (int mode = 0; mode < 3; ++mode)
    for (int bytes : {1024, 32768, 131072}) {
      CK(hipMemset(flags, 0, 1024 * 32 * 4)); CK(hipMemset(status, 0, 64));
      if (mode == 2) hipLaunchKernelGGL(handoff<true>, dim3(256), dim3(512), 0, 0, slices, flags, status, bytes, rounds, 1, t_out, xcc);
      else hipLaunchKernelGGL(handoff<false>, dim3(256), dim3(512), 0, 0, slices, flags, status, bytes, rounds, mode, t_out, xcc);
      CK(hipDeviceSynchronize());
      unsigned st[2]; CK(hipMemcpy(st, status, 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(t.data(), t_out, 256 * 8, hipMemcpyDeviceToHost));
      CK(hipMemcpy(hx.data(), xcc, 256 * 4, hipMemcpyDeviceToHost));
      int split = 0;   // clusters whose four members are NOT on one XCD
      for (int c = 0; c < 64; ++c) split += !(hx[4 * c] == hx[4 * c + 1] && hx[4 * c] == hx[4 * c + 2] && hx[4 * c] == hx[4 * c + 3]);
      unsigned long long mx = 0; double mean = 0;
      for (auto v : t) { mx = v > mx ? v : mx; mean += (double)v; }
      mean /= 256;
      printf("cluster of 4 on %s, slice %6d B: %.2f us per round (mean over workgroups; slowest %.2f)  timeout word 0x%x  mismatches %u  clusters split over XCDs %d\n",
             mode == 0 ? "four XCDs" : mode == 1 ? "ONE XCD  " : "ONE XCD, through its L2 (plain stores, sc0 loads)", bytes, mean / 100.0 / rounds,
             (double)mx / 100.0 / rounds, st[0], st[1], split);
    }
  return 0;
}

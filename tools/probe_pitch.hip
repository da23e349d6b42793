// Does the row pitch of a streamed operand matter to HBM?  dX through the vocabulary reads dlogits [8128][30000] bf16 as K tiles of
// 256 rows x 128 B at a 60-KB pitch (one 128-B line per DRAM page); a K-blocked layout [K/64][rows][64] would make the same tile one
// contiguous 32-KiB block.  256 workgroups (one per CU, 512 threads) each stream "their" tiles with 4 x 16-B loads per lane in
// flight; also the write side of the cross-entropy kernel (one workgroup per row, 128-B lines at a row pitch vs a 1-MB block pitch).
//   hipcc --offload-arch=gfx950 -O3 tools/probe_pitch.hip -o /tmp/pp && /tmp/pp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

// tile (rb, kt): rows 256*rb .. +255, bytes 128*kt .. +127 of each row.  mode 0: row-major, pitch bytes; mode 1: blocked.
__global__ __launch_bounds__(512) void read_tiles(const char* base, int64_t pitch, int rows, int ktiles, int mode, int rbs, unsigned* sink) {
  const int wg = blockIdx.x, nwg = gridDim.x;
  const int lane16 = threadIdx.x & 7, r0 = threadIdx.x >> 3;  // 8 lanes cover a 128-B row segment; 64 rows per pass
  u32x4 acc = {0, 0, 0, 0};
  // workgroup wg owns row block wg % rbs and every (nwg / rbs)-th K tile, like the 4-way K split of the real launch
  const int rb = wg % rbs, ks = wg / rbs, nks = nwg / rbs;
  for (int kt = ks; kt < ktiles; kt += nks) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int row = rb * 256 + p * 64 + r0;
      const int64_t off = mode == 0 ? (int64_t)row * pitch + (int64_t)kt * 128 + lane16 * 16
                                    : ((int64_t)kt * rows + row) * 128 + lane16 * 16;
      const u32x4 v = *reinterpret_cast<const u32x4*>(base + off);
      acc += v;
    }
  }
  if (acc[0] == 0x12345678u) sink[0] = acc[1] + acc[2] + acc[3];
}

// one workgroup (256 threads) per row writes the row's `kt` 128-B segments; mode 0 contiguous row, mode 1 blocked
__global__ __launch_bounds__(256) void write_rows(char* base, int64_t pitch, int rows, int ktiles, int mode) {
  const int row = blockIdx.x;
  const u32x4 v = {(unsigned)row, 1u, 2u, 3u};
  for (int q = threadIdx.x; q < ktiles * 8; q += 256) {
    const int kt = q >> 3, l = q & 7;
    const int64_t off = mode == 0 ? (int64_t)row * pitch + (int64_t)kt * 128 + l * 16 : ((int64_t)kt * rows + row) * 128 + l * 16;
    *reinterpret_cast<u32x4*>(base + off) = v;
  }
}

int main() {
  const int rows = 8192, ktiles = 469;
  const int64_t pitch = 60032;  // >= 469 * 128, the 16-B-aligned row pitch of [*, 30000] bf16 (60000 rounded up)
  const int64_t bytes = (int64_t)rows * pitch + 4096;
  char *a, *flush; unsigned* sink;
  CK(hipMalloc(&a, bytes)); CK(hipMalloc(&flush, 600ll << 20)); CK(hipMalloc(&sink, 64));
  CK(hipMemset(a, 1, bytes));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int mode = 0; mode < 2; ++mode)
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipMemset(flush, rep, 600ll << 20));  // push the operand out of the Infinity Cache
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(read_tiles, dim3(256), dim3(512), 0, 0, a, pitch, rows, ktiles, mode, 32, sink);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      printf("read  %-9s : %7.1f us  %6.2f TB/s\n", mode ? "blocked" : "row-major", ms * 1e3, (double)rows * ktiles * 128 / ms / 1e9);
    }
  for (int mode = 0; mode < 2; ++mode)
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipMemset(flush, rep, 600ll << 20));
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(write_rows, dim3(rows), dim3(256), 0, 0, a, pitch, rows, ktiles, mode);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      printf("write %-9s : %7.1f us  %6.2f TB/s\n", mode ? "blocked" : "row-major", ms * 1e3, (double)rows * ktiles * 128 / ms / 1e9);
    }
  return 0;
}

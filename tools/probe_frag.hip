// Probe the K-strided bf16 fragment loader of mma.hpp against its specification.
#include "../imagetranslate_amd/csrc/mma.hpp"
void imt_set_error(const char*, ...) {}
template <int RB>
__global__ void k(int krow0, int col0, int perm, unsigned short* out) {
  __shared__ __attribute__((aligned(16))) char tile[64 * RB];
  constexpr int COLS = RB / 2;
  for (int i = threadIdx.x; i < 64 * COLS; i += 64) {
    int kr = i / COLS, c = i % COLS;
    *reinterpret_cast<unsigned short*>(tile + tile_off<RB>(kr, c >> 3) + ((c & 7) << 1)) = (unsigned short)(kr * 256 + c);
  }
  __syncthreads();
  bf16x8 v = perm ? lds_frag_kperm_bf16<RB>(tile, krow0, col0) : lds_frag_kstrided_bf16<RB>(tile, krow0, col0);
  typedef __attribute__((ext_vector_type(8))) unsigned short u16x8;
  *reinterpret_cast<u16x8*>(out + threadIdx.x * 8) = __builtin_bit_cast(u16x8, v);  // whole-vector cast (see mma.hpp)
}
template <int RB> int run(int perm) {
  unsigned short* d; (void)hipMalloc(&d, 64 * 8 * 2);
  unsigned short h[512];
  int bad = 0;
  for (int krow0 = 0; krow0 < 64; krow0 += 32)
    for (int col0 = 0; col0 < RB / 2; col0 += 16) {
      hipLaunchKernelGGL(k<RB>, dim3(1), dim3(64), 0, 0, krow0, col0, perm, d);
      (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
      for (int l = 0; l < 64; ++l)
        for (int j = 0; j < 8; ++j) {
          int r = l & 15, g = l >> 4;
          int kr = perm ? krow0 + 16 * (j >> 2) + 4 * g + (j & 3) : krow0 + 8 * g + j;
          int want = kr * 256 + col0 + r;
          if (h[l * 8 + j] != want) {
            if (bad < 12) printf("RB %d perm %d krow0 %d col0 %d lane %d j %d: got (k%d,c%d) want (k%d,c%d)\n", RB, perm, krow0, col0, l, j, h[l*8+j] / 256, h[l*8+j] % 256, kr, col0 + r);
            ++bad;
          }
        }
    }
  printf("RB %d perm %d: %d mismatches\n", RB, perm, bad);
  return bad;
}
int main() { int b = run<256>(0) + run<128>(0) + run<64>(0) + run<256>(1) + run<128>(1) + run<64>(1); return b != 0; }

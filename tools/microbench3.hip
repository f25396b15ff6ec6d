// Does a 2-way interleaved Blake2s (two independent states per lane) raise VALU utilisation at the
// low occupancies the Merkle kernel runs at?  hipcc --offload-arch=gfx950 -O3 tools/microbench3.hip -o /tmp/mb3
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../luminair_amd/csrc/blake2s.h"
using namespace lmn;

__global__ void k_c1(uint32_t* out, int reps) {
  uint32_t h[8], m[16];
  uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  for (int k = 0; k < 16; ++k) m[k] = t * 2654435761u + k;
  b2_init(h);
  for (int r = 0; r < reps; ++r) {
    b2_compress(h, m, 64u, 0xffffffffu);
    m[r & 15] ^= h[0];
  }
  out[t] = h[0] ^ h[7];
}
__global__ void k_c2(uint32_t* out, int reps) {
  uint32_t h[8], m[16], g[8], n[16];
  uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  for (int k = 0; k < 16; ++k) { m[k] = t * 2654435761u + k; n[k] = t * 40503u + 3 * k; }
  b2_init(h);
  b2_init(g);
  for (int r = 0; r < reps; ++r) {
    b2_compress2(h, m, g, n, 64u, 0xffffffffu);
    m[r & 15] ^= h[0];
    n[r & 15] ^= g[0];
  }
  out[t] = h[0] ^ h[7] ^ g[0] ^ g[7];
}
template <class F> float timeit(F f) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  f(); hipDeviceSynchronize();
  hipEventRecord(a); f(); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); return ms;
}
int main() {
  uint32_t* out; hipMalloc(&out, 64u << 20);
  for (int wpc : {1, 2, 3, 4, 8}) {
    int blocks = 256 * wpc, reps = 256;
    float ms = timeit([&] { hipLaunchKernelGGL(k_c1, dim3(blocks), dim3(256), 0, 0, out, reps); });
    printf("1-way: %d waves/SIMD: %.3f ms, %.2f Gcompress/s\n", wpc, ms, (double)blocks * 256 * reps / ms / 1e6);
    ms = timeit([&] { hipLaunchKernelGGL(k_c2, dim3(blocks), dim3(256), 0, 0, out, reps); });
    printf("2-way: %d waves/SIMD: %.3f ms, %.2f Gcompress/s\n", wpc, ms, 2.0 * blocks * 256 * reps / ms / 1e6);
  }
  return 0;
}

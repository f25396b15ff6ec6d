// Micro-benchmarks for the ALU-side ceilings quoted in DESIGN.md: Blake2s compressions/s and
// M31 multiplications/s on gfx950.  hipcc --offload-arch=gfx950 -O3 tools/microbench.hip -o /tmp/mb
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../luminair_amd/csrc/blake2s.h"
#include "../luminair_amd/csrc/field.h"
using namespace lmn;

__global__ void k_compress(uint32_t* out, int reps) {
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
__global__ void k_mul(uint32_t* out, int reps) {
  uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t a = (t * 2654435761u) % P31, b = (t * 40503u + 12345u) % P31, c = a ^ 5, d = b ^ 9;
  for (int r = 0; r < reps; ++r) {
    a = m_mul(a, b); b = m_mul(b, c); c = m_mul(c, d); d = m_mul(d, a);
  }
  out[t] = a ^ b ^ c ^ d;
}
__global__ void k_addsub(uint32_t* out, int reps) {
  uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t a = (t * 2654435761u) % P31, b = (t * 40503u + 12345u) % P31, c = a ^ 5, d = b ^ 9;
  for (int r = 0; r < reps; ++r) {
    a = m_add(a, b); b = m_sub(b, c); c = m_add(c, d); d = m_sub(d, a);
  }
  out[t] = a ^ b ^ c ^ d;
}
template <class F> float timeit(F f) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  f(); hipDeviceSynchronize();
  hipEventRecord(a); f(); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); return ms;
}
int main() {
  uint32_t* out; hipMalloc(&out, 64u << 20);
  for (int wpc : {1, 2, 4, 8}) {   // blocks of 256 per CU
    int blocks = 256 * wpc, reps = 256;
    float ms = timeit([&] { hipLaunchKernelGGL(k_compress, dim3(blocks), dim3(256), 0, 0, out, reps); });
    double n = (double)blocks * 256 * reps;
    printf("compress: %d blocks/CU: %.3f ms, %.2f Gcompress/s\n", wpc, ms, n / ms / 1e6);
  }
  {
    int blocks = 256 * 8, reps = 4096;
    float ms = timeit([&] { hipLaunchKernelGGL(k_mul, dim3(blocks), dim3(256), 0, 0, out, reps); });
    printf("m31 mul: %.3f ms, %.2f Gmul/s\n", ms, (double)blocks * 256 * reps * 4 / ms / 1e6);
    ms = timeit([&] { hipLaunchKernelGGL(k_addsub, dim3(blocks), dim3(256), 0, 0, out, reps); });
    printf("m31 add/sub: %.3f ms, %.2f Gop/s\n", ms, (double)blocks * 256 * reps * 4 / ms / 1e6);
  }
  return 0;
}

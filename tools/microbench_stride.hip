// Does a power-of-two column stride cost HBM bandwidth on MI355X?  The prover keeps a tree's columns back to back
// (column c at base + c * 2^k words), and its kernels read the SAME row of many columns at once (leaf hashing,
// constraints, quotients).  Reads of 16 columns x 2^21 rows with the column stride padded by 0 / 256 B / 4 KiB / 4.25 KiB,
// and the 64-byte-chunk pattern of the strided transform pass (256 chunks per tile at a 16 KiB stride) with and without
// padding.  Build: hipcc --offload-arch=gfx950 -O3 -o tools/bin/mb_stride tools/microbench_stride.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void __launch_bounds__(256) k_cols(const uint32_t* __restrict__ base, uint64_t stride, int ncols, uint32_t* out) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t acc = 0;
#pragma unroll 16
  for (int c = 0; c < ncols; ++c) acc ^= base[(uint64_t)c * stride + i];
  if (acc == 0x12345678u) out[0] = acc;
}

// tile t reads rows r = 0..255, each 16 words (64 B) at word offset r * row_stride + t * 16
__global__ void __launch_bounds__(256) k_chunks(const uint32_t* __restrict__ base, uint64_t row_stride, uint32_t* out) {
  const uint32_t t = blockIdx.x;
  uint32_t acc = 0;
  for (int k = threadIdx.x; k < 4096; k += 256) {
    const uint32_t r = k >> 4, w = k & 15;
    acc ^= base[(uint64_t)r * row_stride + (uint64_t)t * 16 + w];
  }
  if (acc == 0x12345678u) out[0] = acc;
}

int main() {
  const uint64_t N = 1ull << 21;
  uint32_t *buf, *out;
  const uint64_t words = 16 * (N + 4096) + (1ull << 22);
  CHECK(hipMalloc(&buf, words * 4));
  CHECK(hipMalloc(&out, 64));
  CHECK(hipMemset(buf, 1, words * 4));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (uint64_t pad : {0ull, 64ull, 1024ull, 1088ull}) {
    float best = 1e9;
    for (int rep = 0; rep < 6; ++rep) {
      CHECK(hipEventRecord(e0, 0));
      hipLaunchKernelGGL(k_cols, dim3(N / 256), dim3(256), 0, 0, buf, N + pad, 15, out);
      CHECK(hipEventRecord(e1, 0));
      CHECK(hipDeviceSynchronize());
      float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
      if (rep > 0 && ms < best) best = ms;
    }
    printf("15 columns x 2^21 rows, column stride 2^21 + %4llu words: %.1f us  %.0f GB/s\n", (unsigned long long)pad, best * 1e3, 15.0 * N * 4 / (best * 1e-3) / 1e9);
  }
  // strided-pass pattern on a 2^20-word column: 256 rows of 4096 words; tile t = 16-word chunk t of every row
  for (uint64_t pad : {0ull, 16ull, 64ull}) {
    float best = 1e9;
    for (int rep = 0; rep < 6; ++rep) {
      CHECK(hipEventRecord(e0, 0));
      hipLaunchKernelGGL(k_chunks, dim3(256 * 15), dim3(256), 0, 0, buf, 4096 + pad, out);  // 15 columns' worth of tiles over one region (L2-resident reuse avoided by size: 15 * 256 tiles * 16 KiB = 60 MiB)
      CHECK(hipEventRecord(e1, 0));
      CHECK(hipDeviceSynchronize());
      float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
      if (rep > 0 && ms < best) best = ms;
    }
    printf("64-byte chunks, 256 per tile at a row stride of 4096 + %2llu words: %.1f us  %.0f GB/s\n", (unsigned long long)pad, best * 1e3, 15.0 * 256 * 4096 * 4 / (best * 1e-3) / 1e9);
  }
  return 0;
}

// Single-hash LATENCY: one lane per Blake2s compression vs four lanes per compression (column/diagonal
// G functions across a quad, DPP quad_perm rotations, message words fetched from LDS by per-lane offsets).
// hipcc --offload-arch=gfx950 -O3 tools/microbench4.hip -o tools/bin/mb4
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../luminair_amd/csrc/blake2s.h"
using namespace lmn;

__device__ __constant__ unsigned char SIGMA[10][16] = {
    {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3},
    {11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4}, {7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8},
    {9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13}, {2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9},
    {12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11}, {13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10},
    {6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5}, {10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0}};

#define QP(x, ctrl) (uint32_t) __builtin_amdgcn_mov_dpp((int)(x), ctrl, 0xf, 0xf, true)

// chain of dependent 64-byte hashes: m = prev_hash || prev_hash
__global__ void k_chain1(uint32_t* out, int n) {
  __shared__ uint32_t sh[16];
  uint32_t h[8];
  if (threadIdx.x == 0) {
    for (int k = 0; k < 16; ++k) sh[k] = k * 2654435761u;
  }
  __syncthreads();
  for (int r = 0; r < n; ++r) {
    if (threadIdx.x == 0) {
      uint32_t m[16];
      for (int k = 0; k < 16; ++k) m[k] = sh[k];
      b2_init(h);
      b2_compress(h, m, 64u, 0xffffffffu);
      for (int k = 0; k < 8; ++k) { sh[k] = h[k]; sh[8 + k] = h[k]; }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) for (int k = 0; k < 8; ++k) out[k] = sh[k];
}

__global__ void k_chain4(uint32_t* out, int n) {
  __shared__ uint32_t sh[16];
  const uint32_t q = threadIdx.x & 3u;
  if (threadIdx.x == 0) {
    for (int k = 0; k < 16; ++k) sh[k] = k * 2654435761u;
  }
  // per-lane message offsets (words) for the 10 rounds: column step x/y, diagonal step x/y
  uint32_t off[40];
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    off[4 * r + 0] = SIGMA[r][2 * q];
    off[4 * r + 1] = SIGMA[r][2 * q + 1];
    off[4 * r + 2] = SIGMA[r][8 + 2 * q];
    off[4 * r + 3] = SIGMA[r][9 + 2 * q];
  }
  const uint32_t IV[8] = {0x6A09E667u, 0xBB67AE85u, 0x3C6EF372u, 0xA54FF53Au, 0x510E527Fu, 0x9B05688Cu, 0x1F83D9ABu, 0x5BE0CD19u};
  const uint32_t iv_lo = q == 0 ? IV[0] : q == 1 ? IV[1] : q == 2 ? IV[2] : IV[3];
  const uint32_t iv_hi = q == 0 ? IV[4] : q == 1 ? IV[5] : q == 2 ? IV[6] : IV[7];
  const uint32_t h_lo = q == 0 ? (IV[0] ^ 0x01010020u) : iv_lo;
  const uint32_t d0 = iv_hi ^ (q == 0 ? 64u : q == 2 ? 0xffffffffu : 0u);
  __syncthreads();
  for (int it = 0; it < n; ++it) {
    if (threadIdx.x < 4) {
      uint32_t mw[40];
#pragma unroll
      for (int k = 0; k < 40; ++k) mw[k] = sh[off[k]];
      uint32_t a = h_lo, b = iv_hi, c = iv_lo, d = d0;
#pragma unroll
      for (int r = 0; r < 10; ++r) {
        LMN_B2_G(a, b, c, d, mw[4 * r], mw[4 * r + 1])
        b = QP(b, 0x39);
        c = QP(c, 0x4E);
        d = QP(d, 0x93);
        LMN_B2_G(a, b, c, d, mw[4 * r + 2], mw[4 * r + 3])
        b = QP(b, 0x93);
        c = QP(c, 0x4E);
        d = QP(d, 0x39);
      }
      const uint32_t o_lo = h_lo ^ a ^ c, o_hi = iv_hi ^ b ^ d;
      sh[q] = o_lo; sh[4 + q] = o_hi; sh[8 + q] = o_lo; sh[12 + q] = o_hi;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) for (int k = 0; k < 8; ++k) out[k] = sh[k];
}

template <class F> float timeit(F f) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  f(); hipDeviceSynchronize();
  hipEventRecord(a); f(); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); return ms;
}
int main() {
  uint32_t *o1, *o4; hipMalloc(&o1, 64); hipMalloc(&o4, 64);
  const int n = 4096;
  float t1 = timeit([&] { hipLaunchKernelGGL(k_chain1, dim3(1), dim3(64), 0, 0, o1, n); });
  float t4 = timeit([&] { hipLaunchKernelGGL(k_chain4, dim3(1), dim3(64), 0, 0, o4, n); });
  uint32_t h1[8], h4[8];
  hipMemcpy(h1, o1, 32, hipMemcpyDeviceToHost); hipMemcpy(h4, o4, 32, hipMemcpyDeviceToHost);
  bool same = true; for (int k = 0; k < 8; ++k) same = same && h1[k] == h4[k];
  printf("1-lane: %.3f us/hash   4-lane: %.3f us/hash   digests %s (%08x vs %08x)\n", 1e3 * t1 / n, 1e3 * t4 / n,
         same ? "equal" : "DIFFER", h1[0], h4[0]);
  return 0;
}

// Cost of the "last workgroup finishes the job" pattern on MI355X: every workgroup of a memory-writing, ALU-heavy launch
// ends with an agent-scope release (buffer_wbl2 sc1 on gfx950: the 8 XCDs' L2s are not coherent with each other inside a
// launch) and one atomic; the workgroup that arrives last acquires and reduces what the others left.  Compared with the
// same launch without the pattern followed by a second, single-workgroup launch (what the prover's tree tops are today).
//   hipcc --offload-arch=gfx950 -O3 tools/microbench_fence.hip -o /tmp/mb_fence && /tmp/mb_fence
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int MODE>   // 0: plain, 1: wave 0 releases + atomic, last workgroup reduces, 2: every wave releases
__global__ void k_work(unsigned* __restrict__ big, unsigned* __restrict__ nodes, unsigned* counter, unsigned* result, int iters,
                       int words_per_thread) {
  unsigned x = threadIdx.x * 2654435761u + blockIdx.x;
  for (int i = 0; i < iters; ++i) x = (x << 7 | x >> 25) + (x ^ 0x9e3779b9u) * 5u;
  for (int w = 0; w < words_per_thread; ++w) big[((size_t)blockIdx.x * words_per_thread + w) * blockDim.x + threadIdx.x] = x + w;
  if (threadIdx.x < 8) nodes[blockIdx.x * 8 + threadIdx.x] = x;
  if (MODE == 0) return;
  __shared__ unsigned last;
  if (MODE == 2 || threadIdx.x < 64) __threadfence();
  __syncthreads();
  if (threadIdx.x == 0) last = atomicAdd(counter, 1u) == gridDim.x - 1 ? 1u : 0u;
  __syncthreads();
  if (!last) return;
  __threadfence();
  unsigned s = 0;
  for (unsigned i = threadIdx.x; i < gridDim.x * 8; i += blockDim.x) s += __builtin_nontemporal_load(nodes + i);
  atomicAdd(result, s);
  if (threadIdx.x == 0) *counter = 0u;
}
__global__ void k_top(const unsigned* __restrict__ nodes, unsigned* result, unsigned n) {
  unsigned s = 0;
  for (unsigned i = threadIdx.x; i < n; i += blockDim.x) s += nodes[i];
  atomicAdd(result, s);
}

int main() {
  const int blocks[3] = {512, 2048, 8192};
  unsigned *big, *nodes, *counter, *result;
  CK(hipMalloc(&big, (size_t)8192 * 256 * 64 * 4));
  CK(hipMalloc(&nodes, 8192 * 8 * 4));
  CK(hipMalloc(&counter, 4));
  CK(hipMalloc(&result, 4));
  CK(hipMemset(counter, 0, 4));
  hipStream_t s;
  CK(hipStreamCreate(&s));
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  for (int nb : blocks)
    for (int wpt : {8, 32})
      for (int iters : {200, 2000}) {
        float ms[4] = {0, 0, 0, 0};
        unsigned res[4] = {0, 0, 0, 0};
        for (int mode = 0; mode < 4; ++mode) {
          for (int rep = 0; rep < 12; ++rep) {
            CK(hipMemsetAsync(result, 0, 4, s));
            if (rep == 2) CK(hipEventRecord(a, s));
            if (mode == 0) {
              hipLaunchKernelGGL(k_work<0>, dim3(nb), dim3(256), 0, s, big, nodes, counter, result, iters, wpt);
              hipLaunchKernelGGL(k_top, dim3(1), dim3(256), 0, s, nodes, result, (unsigned)nb * 8);
            } else if (mode == 1) {
              hipLaunchKernelGGL(k_work<1>, dim3(nb), dim3(256), 0, s, big, nodes, counter, result, iters, wpt);
            } else if (mode == 2) {
              hipLaunchKernelGGL(k_work<2>, dim3(nb), dim3(256), 0, s, big, nodes, counter, result, iters, wpt);
            } else {
              hipLaunchKernelGGL(k_work<0>, dim3(nb), dim3(256), 0, s, big, nodes, counter, result, iters, wpt);
            }
          }
          CK(hipEventRecord(b, s));
          CK(hipStreamSynchronize(s));
          CK(hipEventElapsedTime(&ms[mode], a, b));
          CK(hipMemcpy(&res[mode], result, 4, hipMemcpyDeviceToHost));
        }
        printf("blocks %5d  words/thread %2d  iters %4d :  work+top launches %7.2f us   last-workgroup (wave 0 releases) %7.2f us   (every wave releases) %7.2f us   work alone %7.2f us   sums %s\n",
               nb, wpt, iters, ms[0] * 100, ms[1] * 100, ms[2] * 100, ms[3] * 100, (res[0] == res[1] && res[1] == res[2]) ? "equal" : "DIFFER");
      }
  return 0;
}

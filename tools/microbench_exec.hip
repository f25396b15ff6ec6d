// Does a gfx950 SIMD skip the 16-lane passes of a wave64 VALU instruction whose lanes are all inactive in EXEC?  A dependent
// chain of v_add / v_xor / v_alignbit (the Blake2s mix) is timed with 64, 32, 16 and 4 active lanes of ONE wave, and with the
// same lanes spread over the four 16-lane groups.  If passes were skipped, the narrow contiguous cases would run faster.
//   hipcc --offload-arch=gfx950 -O3 tools/microbench_exec.hip -o /tmp/mb_exec && /tmp/mb_exec
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void k_chain(unsigned* out, int iters, unsigned active, unsigned stride, long long* cycles) {
  const unsigned lane = threadIdx.x;
  unsigned a = lane * 2654435761u + 1u, b = lane ^ 0x9e3779b9u, c = lane + 77u, d = ~lane;
  const bool on = stride ? (lane % stride == 0 && lane / stride < active) : lane < active;
  long long t0 = 0, t1 = 0;
  if (on) {
    t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
      a = a + b + c;
      d = __builtin_amdgcn_alignbit(d ^ a, d ^ a, 16);
      c = c + d;
      b = __builtin_amdgcn_alignbit(b ^ c, b ^ c, 12);
      a = a + b + d;
      d = __builtin_amdgcn_alignbit(d ^ a, d ^ a, 8);
      c = c + d;
      b = __builtin_amdgcn_alignbit(b ^ c, b ^ c, 7);
    }
    t1 = __builtin_readcyclecounter();
    out[lane] = a ^ b ^ c ^ d;
    if (lane == 0) *cycles = t1 - t0;
  }
}

int main() {
  unsigned* out;
  long long* cyc;
  CK(hipMalloc(&out, 256));
  CK(hipMalloc(&cyc, 8));
  const int iters = 20000;
  struct { unsigned active, stride; const char* what; } cases[] = {
      {64, 0, "64 lanes"}, {32, 0, "lanes 0..31"}, {16, 0, "lanes 0..15"}, {4, 0, "lanes 0..3"},
      {4, 16, "4 lanes, one per 16-lane group"}, {16, 4, "16 lanes, every fourth"}};
  for (auto& c : cases) {
    long long h = 0;
    for (int rep = 0; rep < 3; ++rep) {
      hipLaunchKernelGGL(k_chain, dim3(1), dim3(64), 0, 0, out, iters, c.active, c.stride, cyc);
      CK(hipDeviceSynchronize());
      CK(hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost));
    }
    printf("%-34s %8.2f cycles per 12-instruction step (s_memtime/readcyclecounter units), %.3f per instruction\n", c.what,
           (double)h / iters, (double)h / iters / 12.0);
  }
  return 0;
}

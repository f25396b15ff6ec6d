// Does the gfx950 VALU co-issue two full-rate ops inside one issue window, and under which conditions?
// (follow-up of tools/microbench_issue.hip: full-rate ops reach ~0.41 wave-instr/clk/SIMD alone but a Blake2s stream
// made ONLY of full-rate ops stays at ~0.25.)   hipcc --offload-arch=gfx950 -O3 tools/microbench_pair.hip -o tools/bin/mb_pair
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr int ITERS = 4096;
#define X(dst, src) "v_xor_b32 " dst ", " dst ", " src "\n"
#define A(dst, src) "v_add_u32 " dst ", " dst ", " src "\n"
#define R(dst) "v_alignbit_b32 " dst ", " dst ", " dst ", 12\n"
#define T(dst, s1, s2) "v_add3_u32 " dst ", " dst ", " s1 ", " s2 "\n"

template <int MODE>
__global__ void __launch_bounds__(1024) k_pair(uint32_t* out, uint64_t* cycles, int iters) {
  uint32_t v0 = threadIdx.x, v1 = v0 * 3, v2 = v0 * 5, v3 = v0 * 7, v4 = v0 * 11, v5 = v0 * 13, v6 = v0 * 17, v7 = v0 * 19;
  uint32_t w0 = v0 + 1, w1 = v0 + 2, w2 = v0 + 3, w3 = v0 + 4, w4 = v0 + 5, w5 = v0 + 6, w6 = v0 + 7, w7 = v0 + 8;
  __syncthreads();
  const uint64_t t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#define OPS "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) : "v"(w0), "v"(w1), "v"(w2), "v"(w3), "v"(w4), "v"(w5), "v"(w6), "v"(w7)
#define REP4(s) s s s s
    if (MODE == 0)  // 8 independent chains, common second operand
      asm volatile(REP4(X("%0", "%8") X("%1", "%8") X("%2", "%8") X("%3", "%8") X("%4", "%8") X("%5", "%8") X("%6", "%8") X("%7", "%8")) : OPS);
    if (MODE == 1)  // 8 independent chains, distinct second operands
      asm volatile(REP4(X("%0", "%8") X("%1", "%9") X("%2", "%10") X("%3", "%11") X("%4", "%12") X("%5", "%13") X("%6", "%14") X("%7", "%15")) : OPS);
    if (MODE == 2)  // each op depends on the previous one (pairs on the same chain)
      asm volatile(REP4(X("%0", "%8") X("%0", "%9") X("%1", "%8") X("%1", "%9") X("%2", "%8") X("%2", "%9") X("%3", "%8") X("%3", "%9")) : OPS);
    if (MODE == 3)  // F F H H on independent chains
      asm volatile(REP4(X("%0", "%8") X("%1", "%9") R("%2") R("%3") X("%4", "%12") X("%5", "%13") R("%6") R("%7")) : OPS);
    if (MODE == 4)  // F H F H on independent chains
      asm volatile(REP4(X("%0", "%8") R("%2") X("%1", "%9") R("%3") X("%4", "%12") R("%6") X("%5", "%13") R("%7")) : OPS);
    if (MODE == 5)  // F F F F H H H H
      asm volatile(REP4(X("%0", "%8") X("%1", "%9") X("%4", "%12") X("%5", "%13") R("%2") R("%3") R("%6") R("%7")) : OPS);
    if (MODE == 6)  // xor feeding an independent add: x a x a (different opcodes, independent)
      asm volatile(REP4(X("%0", "%8") A("%1", "%9") X("%2", "%10") A("%3", "%11") X("%4", "%12") A("%5", "%13") X("%6", "%14") A("%7", "%15")) : OPS);
    if (MODE == 7)  // the second op of each pair reads the first one's result (true dependence inside the pair)
      asm volatile(REP4(X("%0", "%8") A("%1", "%0") X("%2", "%10") A("%3", "%2") X("%4", "%12") A("%5", "%4") X("%6", "%14") A("%7", "%6")) : OPS);
    if (MODE == 8)  // Blake2s-like dependent pairs from two interleaved states: (add c0+=d0, add c1+=d1), (xor b0^=c0, xor b1^=c1), H H
      asm volatile(REP4(A("%0", "%1") A("%4", "%5") X("%2", "%0") X("%6", "%4") R("%2") R("%6") T("%3", "%2", "%8") T("%7", "%6", "%9")) : OPS);
    if (MODE == 9)  // same with the two states NOT interleaved
      asm volatile(REP4(A("%0", "%1") X("%2", "%0") R("%2") T("%3", "%2", "%8") A("%4", "%5") X("%6", "%4") R("%6") T("%7", "%6", "%9")) : OPS);
  }
  const uint64_t t1 = __builtin_readcyclecounter();
  uint32_t acc = v0 ^ v1 ^ v2 ^ v3 ^ v4 ^ v5 ^ v6 ^ v7;
  if (acc == 0x12345678u) out[0] = acc;
  if ((threadIdx.x & 63u) == 0) cycles[(uint64_t)blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int MODE>
void bench(const char* label, uint32_t* d_out, uint64_t* d_cyc, double clk_hz) {
  printf("%-44s", label);
  for (int w : {1, 2, 4, 8}) {
    const int threads = w >= 4 ? 1024 : 256 * w, grid = 256 * (w >= 4 ? w / 4 : 1);
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    hipLaunchKernelGGL(k_pair<MODE>, dim3(grid), dim3(threads), 0, 0, d_out, d_cyc, ITERS);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a);
    hipLaunchKernelGGL(k_pair<MODE>, dim3(grid), dim3(threads), 0, 0, d_out, d_cyc, ITERS);
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms;
    (void)hipEventElapsedTime(&ms, a, b);
    const double instr = (double)ITERS * 32, nw = (double)grid * threads / 64;
    printf("  w=%d: %.3f", w, instr * nw / (1024.0 * clk_hz * ms * 1e-3));
  }
  printf("\n");
}

int main() {
  hipDeviceProp_t p;
  (void)hipGetDeviceProperties(&p, 0);
  const double clk_hz = (double)p.clockRate * 1e3;
  printf("wave-instructions / nominal clk / SIMD (wall time), 32 instructions per loop iteration\n");
  uint32_t* d_out;
  uint64_t* d_cyc;
  (void)hipMalloc(&d_out, 64);
  (void)hipMalloc(&d_cyc, 8 * 256 * 2 * 16);
  bench<0>("F x8 independent, common operand", d_out, d_cyc, clk_hz);
  bench<1>("F x8 independent, distinct operands", d_out, d_cyc, clk_hz);
  bench<2>("F dependent pairs (same chain twice)", d_out, d_cyc, clk_hz);
  bench<3>("F F H H (independent chains)", d_out, d_cyc, clk_hz);
  bench<4>("F H F H (independent chains)", d_out, d_cyc, clk_hz);
  bench<5>("F F F F H H H H (independent chains)", d_out, d_cyc, clk_hz);
  bench<6>("xor add xor add (independent)", d_out, d_cyc, clk_hz);
  bench<7>("xor add (add reads the xor result)", d_out, d_cyc, clk_hz);
  bench<8>("Blake-like, two states interleaved FF FF HH HH", d_out, d_cyc, clk_hz);
  bench<9>("Blake-like, two states sequential  F F H H ...", d_out, d_cyc, clk_hz);
  return 0;
}

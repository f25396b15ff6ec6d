// VALU issue-rate microbenchmark for gfx950: wave-instructions per clock per SIMD for the
// integer ops the prover's hot kernels are made of, as a function of resident waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/microbench_issue.hip -o tools/bin/mb_issue && tools/bin/mb_issue
// Every lane runs ILP independent dependency chains of one instruction (inline asm, so the compiler can neither
// fuse nor drop them); cycles are read with s_memtime around the loop of the wave itself, so the result does not
// depend on the clock the chip happens to run at.  grid = 256 workgroups (one per CU) x (256 * W) threads, i.e.
// W waves on each of the 4 SIMDs of every CU (W = 8: two 1024-thread workgroups per CU).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <vector>

constexpr int ILP = 8;
constexpr int UNROLL = 8;     // instructions per chain per loop iteration
constexpr int ITERS = 2048;

#define OP3(name, asmstr)                                                                      \
  struct name {                                                                                \
    static __device__ __forceinline__ void op(uint32_t& a, uint32_t b, uint32_t c) {           \
      asm volatile(asmstr : "+v"(a) : "v"(b), "v"(c));                                         \
    }                                                                                          \
    static const char* label() { return #name; }                                               \
  };

OP3(add_u32, "v_add_u32 %0, %0, %1")
OP3(xor_b32, "v_xor_b32 %0, %0, %1")
OP3(add3_u32, "v_add3_u32 %0, %0, %1, %2")
OP3(alignbit_b32, "v_alignbit_b32 %0, %0, %0, 12")
OP3(xad_u32, "v_xad_u32 %0, %0, %1, %2")
OP3(lshl_add_u32, "v_lshl_add_u32 %0, %0, 3, %1")
OP3(and_or_b32, "v_and_or_b32 %0, %0, %1, %2")
OP3(perm_b32, "v_perm_b32 %0, %0, %0, %1")
OP3(mul_lo_u32, "v_mul_lo_u32 %0, %0, %1")
OP3(mul_hi_u32, "v_mul_hi_u32 %0, %0, %1")
OP3(min_u32, "v_min_u32 %0, %0, %1")
OP3(sub_u32, "v_sub_u32 %0, %0, %1")
OP3(xor_dpp_quad, "v_xor_b32_dpp %0, %0, %1 quad_perm:[1,2,3,0] row_mask:0xf bank_mask:0xf")
OP3(mov_dpp_quad, "v_mov_b32_dpp %0, %1 quad_perm:[1,2,3,0] row_mask:0xf bank_mask:0xf")
OP3(fma_f32, "v_fma_f32 %0, %0, %1, %2")
OP3(bfe_u32, "v_bfe_u32 %0, %0, 3, 20")
OP3(lshrrev_b32, "v_lshrrev_b32 %0, 1, %0")
OP3(mad_dummy, "v_mad_u32_u24 %0, %0, %1, %2")
OP3(xor_sdwa, "v_xor_b32_sdwa %0, %0, %1 dst_sel:WORD_0 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:WORD_1")

// 64-bit multiply-add: the accumulator is a register pair
struct mad_u64_u32 {
  static __device__ __forceinline__ void op64(uint64_t& acc, uint32_t b, uint32_t c) {
    asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(b), "v"(c) : "vcc");
  }
  static const char* label() { return "mad_u64_u32"; }
};

template <class OP>
__global__ void __launch_bounds__(1024) k_issue(uint32_t* out, uint64_t* cycles, int iters) {
  uint32_t v[ILP];
#pragma unroll
  for (int k = 0; k < ILP; ++k) v[k] = threadIdx.x * 2654435761u + k;
  const uint32_t b = threadIdx.x | 1u, c = blockIdx.x + 7u;
  __syncthreads();
  const uint64_t r0 = __builtin_amdgcn_s_memrealtime();
  const uint64_t t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
#pragma unroll
      for (int k = 0; k < ILP; ++k) OP::op(v[k], b, c);
    }
  }
  const uint64_t t1 = __builtin_readcyclecounter();
  const uint64_t r1 = __builtin_amdgcn_s_memrealtime();
  uint32_t acc = 0;
#pragma unroll
  for (int k = 0; k < ILP; ++k) acc ^= v[k];
  if (acc == 0x12345678u) out[0] = acc;
  if ((threadIdx.x & 63u) == 0) {
    const uint64_t w = (uint64_t)blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
    cycles[2 * w] = t1 - t0;
    cycles[2 * w + 1] = r1 - r0;
  }
}

__global__ void __launch_bounds__(1024) k_issue_mad64(uint32_t* out, uint64_t* cycles, int iters) {
  uint64_t v[ILP];
#pragma unroll
  for (int k = 0; k < ILP; ++k) v[k] = threadIdx.x * 2654435761u + k;
  const uint32_t b = threadIdx.x | 1u, c = blockIdx.x + 7u;
  __syncthreads();
  const uint64_t r0 = __builtin_amdgcn_s_memrealtime();
  const uint64_t t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
#pragma unroll
      for (int k = 0; k < ILP; ++k) mad_u64_u32::op64(v[k], b, c);
    }
  }
  const uint64_t t1 = __builtin_readcyclecounter();
  const uint64_t r1 = __builtin_amdgcn_s_memrealtime();
  uint64_t acc = 0;
#pragma unroll
  for (int k = 0; k < ILP; ++k) acc ^= v[k];
  if (acc == 0x12345678u) out[0] = (uint32_t)acc;
  if ((threadIdx.x & 63u) == 0) {
    const uint64_t w = (uint64_t)blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
    cycles[2 * w] = t1 - t0;
    cycles[2 * w + 1] = r1 - r0;
  }
}

// the Blake2s quarter round as the Merkle kernel issues it (add3, xor, alignbit x4 each), ILP independent states
__global__ void __launch_bounds__(1024) k_issue_blake_g(uint32_t* out, uint64_t* cycles, int iters) {
  uint32_t a[4], b[4], c[4], d[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    a[k] = threadIdx.x + k;
    b[k] = threadIdx.x * 3u + k;
    c[k] = threadIdx.x * 5u + k;
    d[k] = threadIdx.x * 7u + k;
  }
  const uint32_t mx = threadIdx.x | 1u, my = blockIdx.x + 7u;
  __syncthreads();
  const uint64_t r0 = __builtin_amdgcn_s_memrealtime();
  const uint64_t t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        asm volatile(
            "v_add3_u32 %0, %0, %1, %4\n v_xor_b32 %3, %3, %0\n v_alignbit_b32 %3, %3, %3, 16\n"
            "v_add_u32 %2, %2, %3\n v_xor_b32 %1, %1, %2\n v_alignbit_b32 %1, %1, %1, 12\n"
            "v_add3_u32 %0, %0, %1, %5\n v_xor_b32 %3, %3, %0\n v_alignbit_b32 %3, %3, %3, 8\n"
            "v_add_u32 %2, %2, %3\n v_xor_b32 %1, %1, %2\n v_alignbit_b32 %1, %1, %1, 7\n"
            : "+v"(a[k]), "+v"(b[k]), "+v"(c[k]), "+v"(d[k])
            : "v"(mx), "v"(my));
      }
    }
  }
  const uint64_t t1 = __builtin_readcyclecounter();
  const uint64_t r1 = __builtin_amdgcn_s_memrealtime();
  uint32_t acc = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) acc ^= a[k] ^ b[k] ^ c[k] ^ d[k];
  if (acc == 0x12345678u) out[0] = acc;
  if ((threadIdx.x & 63u) == 0) {
    const uint64_t w = (uint64_t)blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
    cycles[2 * w] = t1 - t0;
    cycles[2 * w + 1] = r1 - r0;
  }
}

// the same quarter round with every v_add3_u32 split into two v_add_u32 (14 instructions, 10 of them full rate)
__global__ void __launch_bounds__(1024) k_issue_blake_g_adds(uint32_t* out, uint64_t* cycles, int iters) {
  uint32_t a[4], b[4], c[4], d[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    a[k] = threadIdx.x + k;
    b[k] = threadIdx.x * 3u + k;
    c[k] = threadIdx.x * 5u + k;
    d[k] = threadIdx.x * 7u + k;
  }
  const uint32_t mx = threadIdx.x | 1u, my = blockIdx.x + 7u;
  __syncthreads();
  const uint64_t r0 = __builtin_amdgcn_s_memrealtime();
  const uint64_t t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        asm volatile(
            "v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %4\n v_xor_b32 %3, %3, %0\n v_alignbit_b32 %3, %3, %3, 16\n"
            "v_add_u32 %2, %2, %3\n v_xor_b32 %1, %1, %2\n v_alignbit_b32 %1, %1, %1, 12\n"
            "v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %5\n v_xor_b32 %3, %3, %0\n v_alignbit_b32 %3, %3, %3, 8\n"
            "v_add_u32 %2, %2, %3\n v_xor_b32 %1, %1, %2\n v_alignbit_b32 %1, %1, %1, 7\n"
            : "+v"(a[k]), "+v"(b[k]), "+v"(c[k]), "+v"(d[k])
            : "v"(mx), "v"(my));
      }
    }
  }
  const uint64_t t1 = __builtin_readcyclecounter();
  const uint64_t r1 = __builtin_amdgcn_s_memrealtime();
  uint32_t acc = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) acc ^= a[k] ^ b[k] ^ c[k] ^ d[k];
  if (acc == 0x12345678u) out[0] = acc;
  if ((threadIdx.x & 63u) == 0) {
    const uint64_t w = (uint64_t)blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
    cycles[2 * w] = t1 - t0;
    cycles[2 * w + 1] = r1 - r0;
  }
}

// the same quarter round with every v_add3_u32 split into two v_add_u32 (14 instructions, 10 of them full rate)
__global__ void __launch_bounds__(1024) k_issue_blake_g_allf(uint32_t* out, uint64_t* cycles, int iters) {
  uint32_t a[4], b[4], c[4], d[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    a[k] = threadIdx.x + k;
    b[k] = threadIdx.x * 3u + k;
    c[k] = threadIdx.x * 5u + k;
    d[k] = threadIdx.x * 7u + k;
  }
  const uint32_t mx = threadIdx.x | 1u, my = blockIdx.x + 7u;
  uint32_t tmp = 0;
  __syncthreads();
  const uint64_t r0 = __builtin_amdgcn_s_memrealtime();
  const uint64_t t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        asm volatile(
            "v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %4\n v_xor_b32 %3, %3, %0\n v_lshrrev_b32 %6, 16, %3\n v_lshlrev_b32 %3, 16, %3\n v_or_b32 %3, %3, %6\n"
            "v_add_u32 %2, %2, %3\n v_xor_b32 %1, %1, %2\n v_lshrrev_b32 %6, 12, %1\n v_lshlrev_b32 %1, 20, %1\n v_or_b32 %1, %1, %6\n"
            "v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %5\n v_xor_b32 %3, %3, %0\n v_lshrrev_b32 %6, 8, %3\n v_lshlrev_b32 %3, 24, %3\n v_or_b32 %3, %3, %6\n"
            "v_add_u32 %2, %2, %3\n v_xor_b32 %1, %1, %2\n v_lshrrev_b32 %6, 7, %1\n v_lshlrev_b32 %1, 25, %1\n v_or_b32 %1, %1, %6\n"
            : "+v"(a[k]), "+v"(b[k]), "+v"(c[k]), "+v"(d[k])
            : "v"(mx), "v"(my), "v"(tmp));
      }
    }
  }
  const uint64_t t1 = __builtin_readcyclecounter();
  const uint64_t r1 = __builtin_amdgcn_s_memrealtime();
  uint32_t acc = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) acc ^= a[k] ^ b[k] ^ c[k] ^ d[k];
  if (acc == 0x12345678u) out[0] = acc;
  if ((threadIdx.x & 63u) == 0) {
    const uint64_t w = (uint64_t)blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
    cycles[2 * w] = t1 - t0;
    cycles[2 * w + 1] = r1 - r0;
  }
}

// two instruction kinds alternating over independent chains (is a mixed stream priced per class?)
template <class OPA, class OPB>
__global__ void __launch_bounds__(1024) k_issue_mix(uint32_t* out, uint64_t* cycles, int iters) {
  uint32_t v[ILP];
#pragma unroll
  for (int k = 0; k < ILP; ++k) v[k] = threadIdx.x * 2654435761u + k;
  const uint32_t b = threadIdx.x | 1u, c = blockIdx.x + 7u;
  __syncthreads();
  const uint64_t r0 = __builtin_amdgcn_s_memrealtime();
  const uint64_t t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
#pragma unroll
      for (int k = 0; k < ILP; ++k) {
        if ((k + u) & 1)
          OPA::op(v[k], b, c);
        else
          OPB::op(v[k], b, c);
      }
    }
  }
  const uint64_t t1 = __builtin_readcyclecounter();
  const uint64_t r1 = __builtin_amdgcn_s_memrealtime();
  uint32_t acc = 0;
#pragma unroll
  for (int k = 0; k < ILP; ++k) acc ^= v[k];
  if (acc == 0x12345678u) out[0] = acc;
  if ((threadIdx.x & 63u) == 0) {
    const uint64_t w = (uint64_t)blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
    cycles[2 * w] = t1 - t0;
    cycles[2 * w + 1] = r1 - r0;
  }
}

// Blake2s G with the four independent states interleaved at instruction granularity inside ONE asm block
__global__ void __launch_bounds__(1024) k_issue_blake_g4(uint32_t* out, uint64_t* cycles, int iters) {
  uint32_t a[4], b[4], c[4], d[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    a[k] = threadIdx.x + k;
    b[k] = threadIdx.x * 3u + k;
    c[k] = threadIdx.x * 5u + k;
    d[k] = threadIdx.x * 7u + k;
  }
  const uint32_t mx = threadIdx.x | 1u, my = blockIdx.x + 7u;
  __syncthreads();
  const uint64_t r0 = __builtin_amdgcn_s_memrealtime();
  const uint64_t t0 = __builtin_readcyclecounter();
#define STEP4(INS0, INS1, INS2, INS3) INS0 "\n" INS1 "\n" INS2 "\n" INS3 "\n"
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      asm volatile(
          STEP4("v_add3_u32 %0, %0, %4, %16", "v_add3_u32 %1, %1, %5, %16", "v_add3_u32 %2, %2, %6, %16", "v_add3_u32 %3, %3, %7, %16")
          STEP4("v_xor_b32 %12, %12, %0", "v_xor_b32 %13, %13, %1", "v_xor_b32 %14, %14, %2", "v_xor_b32 %15, %15, %3")
          STEP4("v_alignbit_b32 %12, %12, %12, 16", "v_alignbit_b32 %13, %13, %13, 16", "v_alignbit_b32 %14, %14, %14, 16", "v_alignbit_b32 %15, %15, %15, 16")
          STEP4("v_add_u32 %8, %8, %12", "v_add_u32 %9, %9, %13", "v_add_u32 %10, %10, %14", "v_add_u32 %11, %11, %15")
          STEP4("v_xor_b32 %4, %4, %8", "v_xor_b32 %5, %5, %9", "v_xor_b32 %6, %6, %10", "v_xor_b32 %7, %7, %11")
          STEP4("v_alignbit_b32 %4, %4, %4, 12", "v_alignbit_b32 %5, %5, %5, 12", "v_alignbit_b32 %6, %6, %6, 12", "v_alignbit_b32 %7, %7, %7, 12")
          STEP4("v_add3_u32 %0, %0, %4, %17", "v_add3_u32 %1, %1, %5, %17", "v_add3_u32 %2, %2, %6, %17", "v_add3_u32 %3, %3, %7, %17")
          STEP4("v_xor_b32 %12, %12, %0", "v_xor_b32 %13, %13, %1", "v_xor_b32 %14, %14, %2", "v_xor_b32 %15, %15, %3")
          STEP4("v_alignbit_b32 %12, %12, %12, 8", "v_alignbit_b32 %13, %13, %13, 8", "v_alignbit_b32 %14, %14, %14, 8", "v_alignbit_b32 %15, %15, %15, 8")
          STEP4("v_add_u32 %8, %8, %12", "v_add_u32 %9, %9, %13", "v_add_u32 %10, %10, %14", "v_add_u32 %11, %11, %15")
          STEP4("v_xor_b32 %4, %4, %8", "v_xor_b32 %5, %5, %9", "v_xor_b32 %6, %6, %10", "v_xor_b32 %7, %7, %11")
          STEP4("v_alignbit_b32 %4, %4, %4, 7", "v_alignbit_b32 %5, %5, %5, 7", "v_alignbit_b32 %6, %6, %6, 7", "v_alignbit_b32 %7, %7, %7, 7")
          : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(c[0]),
            "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3])
          : "v"(mx), "v"(my));
    }
  }
  const uint64_t t1 = __builtin_readcyclecounter();
  const uint64_t r1 = __builtin_amdgcn_s_memrealtime();
  uint32_t acc = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) acc ^= a[k] ^ b[k] ^ c[k] ^ d[k];
  if (acc == 0x12345678u) out[0] = acc;
  if ((threadIdx.x & 63u) == 0) {
    const uint64_t w = (uint64_t)blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
    cycles[2 * w] = t1 - t0;
    cycles[2 * w + 1] = r1 - r0;
  }
}

struct Result {
  double ipc;       // wave-instructions per shader clock per SIMD (s_memtime of the waves themselves)
  double wall_ipc;  // the same from HIP-event wall time, per NOMINAL clock (clockRate)
  double sclk_mhz;  // shader clock the waves saw: s_memtime ticks per s_memrealtime tick (100 MHz)
};

static int g_cus = 256;     // workgroups per "one per CU" wave of the launch (fewer: part of the chip stays idle)
static int g_iters = 2048;

template <class Launch>
Result run(Launch launch, int waves_per_simd, double instr_per_wave, double clk_hz, uint32_t* d_out, uint64_t* d_cyc) {
  const int threads = waves_per_simd >= 4 ? 1024 : 256 * waves_per_simd;
  const int blocks_per_cu = waves_per_simd >= 4 ? waves_per_simd / 4 : 1;
  const int grid = g_cus * blocks_per_cu;
  hipEvent_t a, b;
  (void)hipEventCreate(&a);
  (void)hipEventCreate(&b);
  launch(grid, threads);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(a);
  launch(grid, threads);
  (void)hipEventRecord(b);
  (void)hipEventSynchronize(b);
  float ms;
  (void)hipEventElapsedTime(&ms, a, b);
  const int nw = grid * threads / 64;
  std::vector<uint64_t> cyc(2 * nw);
  (void)hipMemcpy(cyc.data(), d_cyc, 2 * nw * 8, hipMemcpyDeviceToHost);
  double mean = 0, real = 0;
  for (int w = 0; w < nw; ++w) {
    mean += (double)cyc[2 * w];
    real += (double)cyc[2 * w + 1];
  }
  mean /= nw;
  real /= nw;
  Result r;
  r.ipc = instr_per_wave * waves_per_simd / mean;
  r.wall_ipc = instr_per_wave * nw / (4.0 * g_cus * clk_hz * ms * 1e-3);
  r.sclk_mhz = mean / real * 100.0;
  (void)hipEventDestroy(a);
  (void)hipEventDestroy(b);
  return r;
}

static void show(const Result& r, int w) { printf("  w=%d: %.3f (%.3f @%4.0f)", w, r.ipc, r.wall_ipc, r.sclk_mhz); }

template <class OPA, class OPB>
void bench_mix(double clk_hz, uint32_t* d_out, uint64_t* d_cyc) {
  char name[64];
  snprintf(name, sizeof name, "%s+%s", OPA::label(), OPB::label());
  printf("%-22s", name);
  for (int w : {1, 2, 4, 8}) {
    Result r = run([&](int g, int t) { hipLaunchKernelGGL((k_issue_mix<OPA, OPB>), dim3(g), dim3(t), 0, 0, d_out, d_cyc, g_iters); },
                   w, (double)g_iters * UNROLL * ILP, clk_hz, d_out, d_cyc);
    show(r, w);
  }
  printf("\n");
}

template <class OP>
void bench(double clk_hz, uint32_t* d_out, uint64_t* d_cyc) {
  printf("%-14s", OP::label());
  for (int w : {1, 2, 4, 8}) {
    Result r = run([&](int g, int t) { hipLaunchKernelGGL(k_issue<OP>, dim3(g), dim3(t), 0, 0, d_out, d_cyc, g_iters); }, w,
                   (double)g_iters * UNROLL * ILP, clk_hz, d_out, d_cyc);
    show(r, w);
  }
  printf("\n");
}

int main(int argc, char** argv) {
  if (argc > 1) g_cus = atoi(argv[1]);
  if (argc > 2) g_iters = atoi(argv[2]);
  hipDeviceProp_t p;
  (void)hipGetDeviceProperties(&p, 0);
  const double clk_hz = (double)p.clockRate * 1e3;
  printf("device %s  CUs %d  clockRate %.0f MHz  workgroups-per-round %d  iters %d\n", p.name, p.multiProcessorCount,
         clk_hz / 1e6, g_cus, g_iters);
  printf("wave-instructions / shader-clk / SIMD from s_memtime; in brackets: from wall time per NOMINAL clock, and the\n"
         "shader clock (MHz) the waves saw (s_memtime ticks per 100 MHz s_memrealtime tick); ILP=%d chains per lane\n", ILP);
  uint32_t* d_out;
  uint64_t* d_cyc;
  (void)hipMalloc(&d_out, 64);
  (void)hipMalloc(&d_cyc, 16 * 256 * 2 * 16);
  bench<add_u32>(clk_hz, d_out, d_cyc);
  bench<sub_u32>(clk_hz, d_out, d_cyc);
  bench<min_u32>(clk_hz, d_out, d_cyc);
  bench<xor_b32>(clk_hz, d_out, d_cyc);
  bench<lshrrev_b32>(clk_hz, d_out, d_cyc);
  bench<add3_u32>(clk_hz, d_out, d_cyc);
  bench<alignbit_b32>(clk_hz, d_out, d_cyc);
  bench<xad_u32>(clk_hz, d_out, d_cyc);
  bench<lshl_add_u32>(clk_hz, d_out, d_cyc);
  bench<and_or_b32>(clk_hz, d_out, d_cyc);
  bench<bfe_u32>(clk_hz, d_out, d_cyc);
  bench<perm_b32>(clk_hz, d_out, d_cyc);
  bench<mul_lo_u32>(clk_hz, d_out, d_cyc);
  bench<mul_hi_u32>(clk_hz, d_out, d_cyc);
  bench<xor_dpp_quad>(clk_hz, d_out, d_cyc);
  bench<mov_dpp_quad>(clk_hz, d_out, d_cyc);
  bench<fma_f32>(clk_hz, d_out, d_cyc);
  bench<mad_dummy>(clk_hz, d_out, d_cyc);
  bench<xor_sdwa>(clk_hz, d_out, d_cyc);
  printf("%-14s", "mad_u64_u32");
  for (int w : {1, 2, 4, 8}) {
    Result r = run([&](int g, int t) { hipLaunchKernelGGL(k_issue_mad64, dim3(g), dim3(t), 0, 0, d_out, d_cyc, g_iters); }, w,
                   (double)g_iters * UNROLL * ILP, clk_hz, d_out, d_cyc);
    show(r, w);
  }
  printf("\n%-14s", "blake2s_G");
  for (int w : {1, 2, 4, 8}) {
    Result r = run([&](int g, int t) { hipLaunchKernelGGL(k_issue_blake_g, dim3(g), dim3(t), 0, 0, d_out, d_cyc, g_iters); }, w,
                   (double)g_iters * 4 * 4 * 12, clk_hz, d_out, d_cyc);
    show(r, w);
  }
  printf("\n%-14s", "blake2s_G adds");
  for (int w : {1, 2, 4, 8}) {
    Result r = run([&](int g, int t) { hipLaunchKernelGGL(k_issue_blake_g_adds, dim3(g), dim3(t), 0, 0, d_out, d_cyc, g_iters); }, w,
                   (double)g_iters * 4 * 4 * 14, clk_hz, d_out, d_cyc);
    show(r, w);
  }
  printf("  [14 instr per G: compare G-rounds/clk = value/14 with blake2s_G value/12]");
  printf("\n%-14s", "blake2s_G allF");
  for (int w : {1, 2, 4, 8}) {
    Result r = run([&](int g, int t) { hipLaunchKernelGGL(k_issue_blake_g_allf, dim3(g), dim3(t), 0, 0, d_out, d_cyc, g_iters); }, w,
                   (double)g_iters * 4 * 4 * 22, clk_hz, d_out, d_cyc);
    show(r, w);
  }
  printf("  [22 instr per G]");
  printf("\n%-14s", "blake2s_G x4il");
  for (int w : {1, 2, 4, 8}) {
    Result r = run([&](int g, int t) { hipLaunchKernelGGL(k_issue_blake_g4, dim3(g), dim3(t), 0, 0, d_out, d_cyc, g_iters); }, w,
                   (double)g_iters * 4 * 4 * 12, clk_hz, d_out, d_cyc);
    show(r, w);
  }
  printf("\n");
  bench_mix<xor_b32, alignbit_b32>(clk_hz, d_out, d_cyc);
  bench_mix<add_u32, add3_u32>(clk_hz, d_out, d_cyc);
  bench_mix<xor_b32, add_u32>(clk_hz, d_out, d_cyc);
  bench_mix<mul_lo_u32, add_u32>(clk_hz, d_out, d_cyc);
  bench_mix<mad_dummy, add_u32>(clk_hz, d_out, d_cyc);
  return 0;
}

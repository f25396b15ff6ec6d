// Third, independent measurement of the gfx950 VALU issue rate: round 2's microbenchmark printed
// two figures that disagree by 1.8x (e.g. blake2s_G, 8 waves per SIMD: 0.446 wave-instructions/clk/SIMD from the
// waves' own s_memtime stamps, 0.249 from wall time).  This tool records, for EVERY wave, its start and end shader
// clock (s_memtime), its 100 MHz real-time stamps and WHERE it ran (HW_ID: SE / CU / SIMD, XCC_ID), and derives per
// SIMD:
//   per-wave   = instructions of the SIMD's waves / MEAN elapsed cycles of those waves      (round 2's first figure)
//   makespan   = instructions of the SIMD's waves / (last end - first start) on that SIMD   (what the SIMD delivered)
//   wall       = all instructions / (HIP-event time x measured shader clock) / SIMDs used   (round 2's second figure)
// plus the placement proof (distinct XCCs / CUs / SIMDs, waves per SIMD min..max) and how far apart the waves of one
// SIMD finish (oldest-first arbitration makes co-resident waves finish one after the other, so the MEAN elapsed time
// of a SIMD's waves is shorter than the time the SIMD was busy: that is the whole discrepancy).
//   hipcc --offload-arch=gfx950 -O3 tools/microbench_reconcile.hip -o tools/bin/mb_reconcile && tools/bin/mb_reconcile
// Under rocprofv3 (--pmc SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace) run it with
// `mb_reconcile pmc`: one launch per stream at 8 waves per SIMD only, so that kernel names map to streams.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

constexpr int ILP = 8;
constexpr int UNROLL = 8;

struct WaveRec {
  uint64_t t0, t1, r0, r1;
  uint32_t hw_id, xcc_id;
};

#define HWREG_ALL(id) ((31 << 11) | (id))
__device__ __forceinline__ void stamp_begin(WaveRec& w) {
  w.hw_id = __builtin_amdgcn_s_getreg(HWREG_ALL(4));    // HW_REG_HW_ID
  w.xcc_id = __builtin_amdgcn_s_getreg(HWREG_ALL(20));  // HW_REG_XCC_ID
  w.r0 = __builtin_amdgcn_s_memrealtime();
  w.t0 = __builtin_readcyclecounter();
}
__device__ __forceinline__ void stamp_end(WaveRec& w, WaveRec* out) {
  w.t1 = __builtin_readcyclecounter();
  w.r1 = __builtin_amdgcn_s_memrealtime();
  if ((threadIdx.x & 63u) == 0) out[(uint64_t)blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = w;
}

#define STREAM_KERNEL(name, BODY, INSTR_PER_ITER)                                                  \
  __global__ void __launch_bounds__(256) name(uint32_t* out, WaveRec* recs, int iters) {           \
    uint32_t v[ILP], w[ILP];                                                                       \
    _Pragma("unroll") for (int k = 0; k < ILP; ++k) {                                              \
      v[k] = threadIdx.x * 2654435761u + k;                                                        \
      w[k] = threadIdx.x * 40503u + 977u * k + blockIdx.x;                                         \
    }                                                                                              \
    const uint32_t b = threadIdx.x | 1u, c = blockIdx.x + 7u;                                      \
    (void)b; (void)c;                                                                              \
    WaveRec rec;                                                                                   \
    __syncthreads();                                                                               \
    stamp_begin(rec);                                                                              \
    for (int it = 0; it < iters; ++it) {                                                           \
      _Pragma("unroll") for (int u = 0; u < UNROLL; ++u) {                                         \
        _Pragma("unroll") for (int k = 0; k < ILP; ++k) { BODY; }                                  \
      }                                                                                            \
    }                                                                                              \
    stamp_end(rec, recs);                                                                          \
    uint32_t acc = 0;                                                                              \
    _Pragma("unroll") for (int k = 0; k < ILP; ++k) acc ^= v[k] ^ w[k];                            \
    if (acc == 0x12345678u) out[0] = acc;                                                          \
  }                                                                                                \
  static const int name##_ipi = (INSTR_PER_ITER);

#define STREAM_KERNEL_U(name, BODY, INSTR_PER_ITER, UNR)                                                  \
  __global__ void __launch_bounds__(256) name(uint32_t* out, WaveRec* recs, int iters) {           \
    uint32_t v[ILP], w[ILP];                                                                       \
    _Pragma("unroll") for (int k = 0; k < ILP; ++k) {                                              \
      v[k] = threadIdx.x * 2654435761u + k;                                                        \
      w[k] = threadIdx.x * 40503u + 977u * k + blockIdx.x;                                         \
    }                                                                                              \
    const uint32_t b = threadIdx.x | 1u, c = blockIdx.x + 7u;                                      \
    (void)b; (void)c;                                                                              \
    WaveRec rec;                                                                                   \
    __syncthreads();                                                                               \
    stamp_begin(rec);                                                                              \
    for (int it = 0; it < iters; ++it) {                                                           \
      _Pragma("unroll") for (int u = 0; u < (UNR); ++u) {                                         \
        _Pragma("unroll") for (int k = 0; k < ILP; ++k) { BODY; }                                  \
      }                                                                                            \
    }                                                                                              \
    stamp_end(rec, recs);                                                                          \
    uint32_t acc = 0;                                                                              \
    _Pragma("unroll") for (int k = 0; k < ILP; ++k) acc ^= v[k] ^ w[k];                            \
    if (acc == 0x12345678u) out[0] = acc;                                                          \
  }                                                                                                \
  static const int name##_ipi = (INSTR_PER_ITER);

// one-instruction streams: ILP independent chains; second operands are loop constants shared by all chains
STREAM_KERNEL(k_add_u32, asm volatile("v_add_u32 %0, %0, %1" : "+v"(v[k]) : "v"(b)), ILP* UNROLL)
STREAM_KERNEL(k_fma_f32, asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[k]) : "v"(b), "v"(c)), ILP* UNROLL)
STREAM_KERNEL(k_add3_u32, asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(v[k]) : "v"(b), "v"(c)), ILP* UNROLL)
STREAM_KERNEL(k_alignbit_b32, asm volatile("v_alignbit_b32 %0, %0, %0, 12" : "+v"(v[k])), ILP* UNROLL)
// the plain two-operand op with a DIFFERENT live register as second source in every chain (operand-cache question)
STREAM_KERNEL(k_xor_distinct, asm volatile("v_xor_b32 %0, %0, %1" : "+v"(v[k]) : "v"(w[k])), ILP* UNROLL)
STREAM_KERNEL(k_add3_distinct, asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(v[k]) : "v"(w[k]), "v"(w[(k + 1) % ILP])),
              ILP* UNROLL)

// the Blake2s quarter round as the Merkle kernel issues it: 12 instructions, 4 independent states per lane
__global__ void __launch_bounds__(256) k_blake2s_G(uint32_t* out, WaveRec* recs, int iters) {
  uint32_t a[4], b[4], c[4], d[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    a[k] = threadIdx.x + k;
    b[k] = threadIdx.x * 3u + k;
    c[k] = threadIdx.x * 5u + k;
    d[k] = threadIdx.x * 7u + k;
  }
  const uint32_t mx = threadIdx.x | 1u, my = blockIdx.x + 7u;
  WaveRec rec;
  __syncthreads();
  stamp_begin(rec);
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
  stamp_end(rec, recs);
  uint32_t acc = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) acc ^= a[k] ^ b[k] ^ c[k] ^ d[k];
  if (acc == 0x12345678u) out[0] = acc;
}
static const int k_blake2s_G_ipi = 4 * 4 * 12;

// the same quarter round made of plain two-operand ops only (add3 -> 2 adds, rotation -> shift, shift, or; every
// state has its own temporary): 22 instructions
__global__ void __launch_bounds__(256) k_blake2s_G_plain(uint32_t* out, WaveRec* recs, int iters) {
  uint32_t a[4], b[4], c[4], d[4], t[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    a[k] = threadIdx.x + k;
    b[k] = threadIdx.x * 3u + k;
    c[k] = threadIdx.x * 5u + k;
    d[k] = threadIdx.x * 7u + k;
    t[k] = 0;
  }
  const uint32_t mx = threadIdx.x | 1u, my = blockIdx.x + 7u;
  WaveRec rec;
  __syncthreads();
  stamp_begin(rec);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        asm volatile(
            "v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %5\n v_xor_b32 %3, %3, %0\n v_lshrrev_b32 %4, 16, %3\n v_lshlrev_b32 %3, 16, %3\n v_or_b32 %3, %3, %4\n"
            "v_add_u32 %2, %2, %3\n v_xor_b32 %1, %1, %2\n v_lshrrev_b32 %4, 12, %1\n v_lshlrev_b32 %1, 20, %1\n v_or_b32 %1, %1, %4\n"
            "v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %6\n v_xor_b32 %3, %3, %0\n v_lshrrev_b32 %4, 8, %3\n v_lshlrev_b32 %3, 24, %3\n v_or_b32 %3, %3, %4\n"
            "v_add_u32 %2, %2, %3\n v_xor_b32 %1, %1, %2\n v_lshrrev_b32 %4, 7, %1\n v_lshlrev_b32 %1, 25, %1\n v_or_b32 %1, %1, %4\n"
            : "+v"(a[k]), "+v"(b[k]), "+v"(c[k]), "+v"(d[k]), "+v"(t[k])
            : "v"(mx), "v"(my));
      }
    }
  }
  stamp_end(rec, recs);
  uint32_t acc = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) acc ^= a[k] ^ b[k] ^ c[k] ^ d[k] ^ t[k];
  if (acc == 0x12345678u) out[0] = acc;
}
static const int k_blake2s_G_plain_ipi = 4 * 4 * 22;


// dependence questions (round 3, second pass): ONE fully dependent chain of the plain op per lane, two chains, and a
// 1:1 mix of a half-rate and a full-rate op in independent chains
STREAM_KERNEL(k_add_dep1, asm volatile("v_add_u32 %0, %0, %1" : "+v"(v[0]) : "v"(b)), ILP* UNROLL)
STREAM_KERNEL(k_add_dep2, asm volatile("v_add_u32 %0, %0, %1" : "+v"(v[k & 1]) : "v"(b)), ILP* UNROLL)
STREAM_KERNEL(k_add_dep4, asm volatile("v_add_u32 %0, %0, %1" : "+v"(v[k & 3]) : "v"(b)), ILP* UNROLL)
STREAM_KERNEL(k_mix_add3_xor,
              if (k & 1) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(v[k]) : "v"(b), "v"(c));
              else asm volatile("v_xor_b32 %0, %0, %1" : "+v"(v[k]) : "v"(w[k])), ILP* UNROLL)

// which property of the "plain" quarter round keeps it at 0.25?  single ops it uses, its op sequence on independent chains,
// and the code-size question (the same independent add stream with a 4 KB loop body)
STREAM_KERNEL(k_or_b32, asm volatile("v_or_b32 %0, %0, %1" : "+v"(v[k]) : "v"(b)), ILP* UNROLL)
STREAM_KERNEL(k_lshl_c, asm volatile("v_lshlrev_b32 %0, 7, %0" : "+v"(v[k])), ILP* UNROLL)
STREAM_KERNEL(k_lshr_c, asm volatile("v_lshrrev_b32 %0, 12, %0" : "+v"(v[k])), ILP* UNROLL)
STREAM_KERNEL(k_lshr_2reg, asm volatile("v_lshrrev_b32 %1, 12, %0\n v_xor_b32 %0, %0, %1" : "+v"(v[k]), "+v"(w[k])), 2 * ILP* UNROLL)
STREAM_KERNEL(k_rot_seq, asm volatile("v_xor_b32 %0, %0, %2\n v_lshrrev_b32 %1, 12, %0\n v_lshlrev_b32 %0, 20, %0\n v_or_b32 %0, %0, %1\n v_add_u32 %0, %0, %2"
                                      : "+v"(v[k]), "+v"(w[k]) : "v"(b)), 5 * ILP* UNROLL)
STREAM_KERNEL_U(k_add_u32_big, asm volatile("v_add_u32 %0, %0, %1" : "+v"(v[k]) : "v"(b)), ILP * 128, 128)
STREAM_KERNEL_U(k_add_dep1_big, asm volatile("v_add_u32 %0, %0, %1" : "+v"(v[0]) : "v"(b)), ILP * 128, 128)

// ---- generated: the same quarter rounds with the independent states of a lane interleaved instruction by instruction

__global__ void __launch_bounds__(256) k_blake2s_G_ilv4(uint32_t* out, WaveRec* recs, int iters) {
  uint32_t a[4], b[4], c[4], d[4], t[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    a[k] = threadIdx.x + k;
    b[k] = threadIdx.x * 3u + k;
    c[k] = threadIdx.x * 5u + k;
    d[k] = threadIdx.x * 7u + k;
    t[k] = 0;
  }
  const uint32_t mx = threadIdx.x | 1u, my = blockIdx.x + 7u;
  WaveRec rec;
  __syncthreads();
  stamp_begin(rec);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      asm volatile(
            "v_add3_u32 %0, %0, %4, %16\n v_add3_u32 %1, %1, %5, %16\n v_add3_u32 %2, %2, %6, %16\n v_add3_u32 %3, %3, %7, %16\n"
            "v_xor_b32 %12, %12, %0\n v_xor_b32 %13, %13, %1\n v_xor_b32 %14, %14, %2\n v_xor_b32 %15, %15, %3\n"
            "v_alignbit_b32 %12, %12, %12, 16\n v_alignbit_b32 %13, %13, %13, 16\n v_alignbit_b32 %14, %14, %14, 16\n v_alignbit_b32 %15, %15, %15, 16\n"
            "v_add_u32 %8, %8, %12\n v_add_u32 %9, %9, %13\n v_add_u32 %10, %10, %14\n v_add_u32 %11, %11, %15\n"
            "v_xor_b32 %4, %4, %8\n v_xor_b32 %5, %5, %9\n v_xor_b32 %6, %6, %10\n v_xor_b32 %7, %7, %11\n"
            "v_alignbit_b32 %4, %4, %4, 12\n v_alignbit_b32 %5, %5, %5, 12\n v_alignbit_b32 %6, %6, %6, 12\n v_alignbit_b32 %7, %7, %7, 12\n"
            "v_add3_u32 %0, %0, %4, %17\n v_add3_u32 %1, %1, %5, %17\n v_add3_u32 %2, %2, %6, %17\n v_add3_u32 %3, %3, %7, %17\n"
            "v_xor_b32 %12, %12, %0\n v_xor_b32 %13, %13, %1\n v_xor_b32 %14, %14, %2\n v_xor_b32 %15, %15, %3\n"
            "v_alignbit_b32 %12, %12, %12, 8\n v_alignbit_b32 %13, %13, %13, 8\n v_alignbit_b32 %14, %14, %14, 8\n v_alignbit_b32 %15, %15, %15, 8\n"
            "v_add_u32 %8, %8, %12\n v_add_u32 %9, %9, %13\n v_add_u32 %10, %10, %14\n v_add_u32 %11, %11, %15\n"
            "v_xor_b32 %4, %4, %8\n v_xor_b32 %5, %5, %9\n v_xor_b32 %6, %6, %10\n v_xor_b32 %7, %7, %11\n"
            "v_alignbit_b32 %4, %4, %4, 7\n v_alignbit_b32 %5, %5, %5, 7\n v_alignbit_b32 %6, %6, %6, 7\n v_alignbit_b32 %7, %7, %7, 7\n"
          : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3])
          : "v"(mx), "v"(my));
    }
  }
  stamp_end(rec, recs);
  uint32_t acc = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) acc ^= a[k] ^ b[k] ^ c[k] ^ d[k] ^ t[k];
  if (acc == 0x12345678u) out[0] = acc;
}
static const int k_blake2s_G_ilv4_ipi = 16 * 12;

__global__ void __launch_bounds__(256) k_blake2s_G_plain_ilv4(uint32_t* out, WaveRec* recs, int iters) {
  uint32_t a[4], b[4], c[4], d[4], t[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    a[k] = threadIdx.x + k;
    b[k] = threadIdx.x * 3u + k;
    c[k] = threadIdx.x * 5u + k;
    d[k] = threadIdx.x * 7u + k;
    t[k] = 0;
  }
  const uint32_t mx = threadIdx.x | 1u, my = blockIdx.x + 7u;
  WaveRec rec;
  __syncthreads();
  stamp_begin(rec);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      asm volatile(
            "v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %5\n v_add_u32 %2, %2, %6\n v_add_u32 %3, %3, %7\n"
            "v_add_u32 %0, %0, %20\n v_add_u32 %1, %1, %20\n v_add_u32 %2, %2, %20\n v_add_u32 %3, %3, %20\n"
            "v_xor_b32 %12, %12, %0\n v_xor_b32 %13, %13, %1\n v_xor_b32 %14, %14, %2\n v_xor_b32 %15, %15, %3\n"
            "v_lshrrev_b32 %16, 16, %12\n v_lshrrev_b32 %17, 16, %13\n v_lshrrev_b32 %18, 16, %14\n v_lshrrev_b32 %19, 16, %15\n"
            "v_lshlrev_b32 %12, 16, %12\n v_lshlrev_b32 %13, 16, %13\n v_lshlrev_b32 %14, 16, %14\n v_lshlrev_b32 %15, 16, %15\n"
            "v_or_b32 %12, %12, %16\n v_or_b32 %13, %13, %17\n v_or_b32 %14, %14, %18\n v_or_b32 %15, %15, %19\n"
            "v_add_u32 %8, %8, %12\n v_add_u32 %9, %9, %13\n v_add_u32 %10, %10, %14\n v_add_u32 %11, %11, %15\n"
            "v_xor_b32 %4, %4, %8\n v_xor_b32 %5, %5, %9\n v_xor_b32 %6, %6, %10\n v_xor_b32 %7, %7, %11\n"
            "v_lshrrev_b32 %16, 12, %4\n v_lshrrev_b32 %17, 12, %5\n v_lshrrev_b32 %18, 12, %6\n v_lshrrev_b32 %19, 12, %7\n"
            "v_lshlrev_b32 %4, 20, %4\n v_lshlrev_b32 %5, 20, %5\n v_lshlrev_b32 %6, 20, %6\n v_lshlrev_b32 %7, 20, %7\n"
            "v_or_b32 %4, %4, %16\n v_or_b32 %5, %5, %17\n v_or_b32 %6, %6, %18\n v_or_b32 %7, %7, %19\n"
            "v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %5\n v_add_u32 %2, %2, %6\n v_add_u32 %3, %3, %7\n"
            "v_add_u32 %0, %0, %21\n v_add_u32 %1, %1, %21\n v_add_u32 %2, %2, %21\n v_add_u32 %3, %3, %21\n"
            "v_xor_b32 %12, %12, %0\n v_xor_b32 %13, %13, %1\n v_xor_b32 %14, %14, %2\n v_xor_b32 %15, %15, %3\n"
            "v_lshrrev_b32 %16, 8, %12\n v_lshrrev_b32 %17, 8, %13\n v_lshrrev_b32 %18, 8, %14\n v_lshrrev_b32 %19, 8, %15\n"
            "v_lshlrev_b32 %12, 24, %12\n v_lshlrev_b32 %13, 24, %13\n v_lshlrev_b32 %14, 24, %14\n v_lshlrev_b32 %15, 24, %15\n"
            "v_or_b32 %12, %12, %16\n v_or_b32 %13, %13, %17\n v_or_b32 %14, %14, %18\n v_or_b32 %15, %15, %19\n"
            "v_add_u32 %8, %8, %12\n v_add_u32 %9, %9, %13\n v_add_u32 %10, %10, %14\n v_add_u32 %11, %11, %15\n"
            "v_xor_b32 %4, %4, %8\n v_xor_b32 %5, %5, %9\n v_xor_b32 %6, %6, %10\n v_xor_b32 %7, %7, %11\n"
            "v_lshrrev_b32 %16, 7, %4\n v_lshrrev_b32 %17, 7, %5\n v_lshrrev_b32 %18, 7, %6\n v_lshrrev_b32 %19, 7, %7\n"
            "v_lshlrev_b32 %4, 25, %4\n v_lshlrev_b32 %5, 25, %5\n v_lshlrev_b32 %6, 25, %6\n v_lshlrev_b32 %7, 25, %7\n"
            "v_or_b32 %4, %4, %16\n v_or_b32 %5, %5, %17\n v_or_b32 %6, %6, %18\n v_or_b32 %7, %7, %19\n"
          : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(t[0]), "+v"(t[1]), "+v"(t[2]), "+v"(t[3])
          : "v"(mx), "v"(my));
    }
  }
  stamp_end(rec, recs);
  uint32_t acc = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) acc ^= a[k] ^ b[k] ^ c[k] ^ d[k] ^ t[k];
  if (acc == 0x12345678u) out[0] = acc;
}
static const int k_blake2s_G_plain_ilv4_ipi = 16 * 22;

__global__ void __launch_bounds__(256) k_blake2s_G_ilv8(uint32_t* out, WaveRec* recs, int iters) {
  uint32_t a[8], b[8], c[8], d[8], t[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    a[k] = threadIdx.x + k;
    b[k] = threadIdx.x * 3u + k;
    c[k] = threadIdx.x * 5u + k;
    d[k] = threadIdx.x * 7u + k;
    t[k] = 0;
  }
  const uint32_t mx = threadIdx.x | 1u, my = blockIdx.x + 7u;
  WaveRec rec;
  __syncthreads();
  stamp_begin(rec);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      asm volatile(
            "v_add3_u32 %0, %0, %8, %32\n v_add3_u32 %1, %1, %9, %32\n v_add3_u32 %2, %2, %10, %32\n v_add3_u32 %3, %3, %11, %32\n v_add3_u32 %4, %4, %12, %32\n v_add3_u32 %5, %5, %13, %32\n v_add3_u32 %6, %6, %14, %32\n v_add3_u32 %7, %7, %15, %32\n"
            "v_xor_b32 %24, %24, %0\n v_xor_b32 %25, %25, %1\n v_xor_b32 %26, %26, %2\n v_xor_b32 %27, %27, %3\n v_xor_b32 %28, %28, %4\n v_xor_b32 %29, %29, %5\n v_xor_b32 %30, %30, %6\n v_xor_b32 %31, %31, %7\n"
            "v_alignbit_b32 %24, %24, %24, 16\n v_alignbit_b32 %25, %25, %25, 16\n v_alignbit_b32 %26, %26, %26, 16\n v_alignbit_b32 %27, %27, %27, 16\n v_alignbit_b32 %28, %28, %28, 16\n v_alignbit_b32 %29, %29, %29, 16\n v_alignbit_b32 %30, %30, %30, 16\n v_alignbit_b32 %31, %31, %31, 16\n"
            "v_add_u32 %16, %16, %24\n v_add_u32 %17, %17, %25\n v_add_u32 %18, %18, %26\n v_add_u32 %19, %19, %27\n v_add_u32 %20, %20, %28\n v_add_u32 %21, %21, %29\n v_add_u32 %22, %22, %30\n v_add_u32 %23, %23, %31\n"
            "v_xor_b32 %8, %8, %16\n v_xor_b32 %9, %9, %17\n v_xor_b32 %10, %10, %18\n v_xor_b32 %11, %11, %19\n v_xor_b32 %12, %12, %20\n v_xor_b32 %13, %13, %21\n v_xor_b32 %14, %14, %22\n v_xor_b32 %15, %15, %23\n"
            "v_alignbit_b32 %8, %8, %8, 12\n v_alignbit_b32 %9, %9, %9, 12\n v_alignbit_b32 %10, %10, %10, 12\n v_alignbit_b32 %11, %11, %11, 12\n v_alignbit_b32 %12, %12, %12, 12\n v_alignbit_b32 %13, %13, %13, 12\n v_alignbit_b32 %14, %14, %14, 12\n v_alignbit_b32 %15, %15, %15, 12\n"
            "v_add3_u32 %0, %0, %8, %33\n v_add3_u32 %1, %1, %9, %33\n v_add3_u32 %2, %2, %10, %33\n v_add3_u32 %3, %3, %11, %33\n v_add3_u32 %4, %4, %12, %33\n v_add3_u32 %5, %5, %13, %33\n v_add3_u32 %6, %6, %14, %33\n v_add3_u32 %7, %7, %15, %33\n"
            "v_xor_b32 %24, %24, %0\n v_xor_b32 %25, %25, %1\n v_xor_b32 %26, %26, %2\n v_xor_b32 %27, %27, %3\n v_xor_b32 %28, %28, %4\n v_xor_b32 %29, %29, %5\n v_xor_b32 %30, %30, %6\n v_xor_b32 %31, %31, %7\n"
            "v_alignbit_b32 %24, %24, %24, 8\n v_alignbit_b32 %25, %25, %25, 8\n v_alignbit_b32 %26, %26, %26, 8\n v_alignbit_b32 %27, %27, %27, 8\n v_alignbit_b32 %28, %28, %28, 8\n v_alignbit_b32 %29, %29, %29, 8\n v_alignbit_b32 %30, %30, %30, 8\n v_alignbit_b32 %31, %31, %31, 8\n"
            "v_add_u32 %16, %16, %24\n v_add_u32 %17, %17, %25\n v_add_u32 %18, %18, %26\n v_add_u32 %19, %19, %27\n v_add_u32 %20, %20, %28\n v_add_u32 %21, %21, %29\n v_add_u32 %22, %22, %30\n v_add_u32 %23, %23, %31\n"
            "v_xor_b32 %8, %8, %16\n v_xor_b32 %9, %9, %17\n v_xor_b32 %10, %10, %18\n v_xor_b32 %11, %11, %19\n v_xor_b32 %12, %12, %20\n v_xor_b32 %13, %13, %21\n v_xor_b32 %14, %14, %22\n v_xor_b32 %15, %15, %23\n"
            "v_alignbit_b32 %8, %8, %8, 7\n v_alignbit_b32 %9, %9, %9, 7\n v_alignbit_b32 %10, %10, %10, 7\n v_alignbit_b32 %11, %11, %11, 7\n v_alignbit_b32 %12, %12, %12, 7\n v_alignbit_b32 %13, %13, %13, 7\n v_alignbit_b32 %14, %14, %14, 7\n v_alignbit_b32 %15, %15, %15, 7\n"
          : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(b[4]), "+v"(b[5]), "+v"(b[6]), "+v"(b[7]), "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(c[4]), "+v"(c[5]), "+v"(c[6]), "+v"(c[7]), "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]), "+v"(d[6]), "+v"(d[7])
          : "v"(mx), "v"(my));
    }
  }
  stamp_end(rec, recs);
  uint32_t acc = 0;
#pragma unroll
  for (int k = 0; k < 8; ++k) acc ^= a[k] ^ b[k] ^ c[k] ^ d[k] ^ t[k];
  if (acc == 0x12345678u) out[0] = acc;
}
static const int k_blake2s_G_ilv8_ipi = 16 * 12;


// runs of RUN fast-class instructions (v_xor_b32) alternating with runs of RUN slow-class ones (v_add3_u32), ILP independent
// chains; PRIO 0: no priority changes, 1: slow runs at s_setprio 3 / fast runs at 0, 2: the opposite
template <int PRIO, int RUN>
__global__ void __launch_bounds__(256) k_phase(uint32_t* out, WaveRec* recs, int iters) {
  uint32_t v[ILP], w[ILP];
#pragma unroll
  for (int k = 0; k < ILP; ++k) {
    v[k] = threadIdx.x * 2654435761u + k;
    w[k] = threadIdx.x * 40503u + 977u * k + blockIdx.x;
  }
  const uint32_t b = threadIdx.x | 1u, c = blockIdx.x + 7u;
  WaveRec rec;
  __syncthreads();
  stamp_begin(rec);
  for (int it = 0; it < iters; ++it) {
    if (PRIO == 1) asm volatile("s_setprio 0");
    if (PRIO == 2) asm volatile("s_setprio 3");
#pragma unroll
    for (int u = 0; u < RUN / ILP; ++u) {
#pragma unroll
      for (int k = 0; k < ILP; ++k) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(v[k]) : "v"(w[k]));
    }
    if (PRIO == 1) asm volatile("s_setprio 3");
    if (PRIO == 2) asm volatile("s_setprio 0");
#pragma unroll
    for (int u = 0; u < RUN / ILP; ++u) {
#pragma unroll
      for (int k = 0; k < ILP; ++k) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(v[k]) : "v"(b), "v"(c));
    }
  }
  stamp_end(rec, recs);
  uint32_t acc = 0;
#pragma unroll
  for (int k = 0; k < ILP; ++k) acc ^= v[k] ^ w[k];
  if (acc == 0x12345678u) out[0] = acc;
}
// ---- generated: interleaved quarter rounds with s_setprio around the runs of slow-class / fast-class instructions

__global__ void __launch_bounds__(256) k_blake2s_G_ilv4_prio1(uint32_t* out, WaveRec* recs, int iters) {
  uint32_t a[4], b[4], c[4], d[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    a[k] = threadIdx.x + k;
    b[k] = threadIdx.x * 3u + k;
    c[k] = threadIdx.x * 5u + k;
    d[k] = threadIdx.x * 7u + k;
  }
  const uint32_t mx = threadIdx.x | 1u, my = blockIdx.x + 7u;
  WaveRec rec;
  __syncthreads();
  stamp_begin(rec);
  asm volatile("s_setprio 3");
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      asm volatile(
            "v_add3_u32 %0, %0, %4, %16\n v_add3_u32 %1, %1, %5, %16\n v_add3_u32 %2, %2, %6, %16\n v_add3_u32 %3, %3, %7, %16\n"
            "s_setprio 0\n v_xor_b32 %12, %12, %0\n v_xor_b32 %13, %13, %1\n v_xor_b32 %14, %14, %2\n v_xor_b32 %15, %15, %3\n"
            "s_setprio 3\n v_alignbit_b32 %12, %12, %12, 16\n v_alignbit_b32 %13, %13, %13, 16\n v_alignbit_b32 %14, %14, %14, 16\n v_alignbit_b32 %15, %15, %15, 16\n"
            "s_setprio 0\n v_add_u32 %8, %8, %12\n v_add_u32 %9, %9, %13\n v_add_u32 %10, %10, %14\n v_add_u32 %11, %11, %15\n"
            "v_xor_b32 %4, %4, %8\n v_xor_b32 %5, %5, %9\n v_xor_b32 %6, %6, %10\n v_xor_b32 %7, %7, %11\n"
            "s_setprio 3\n v_alignbit_b32 %4, %4, %4, 12\n v_alignbit_b32 %5, %5, %5, 12\n v_alignbit_b32 %6, %6, %6, 12\n v_alignbit_b32 %7, %7, %7, 12\n"
            "v_add3_u32 %0, %0, %4, %17\n v_add3_u32 %1, %1, %5, %17\n v_add3_u32 %2, %2, %6, %17\n v_add3_u32 %3, %3, %7, %17\n"
            "s_setprio 0\n v_xor_b32 %12, %12, %0\n v_xor_b32 %13, %13, %1\n v_xor_b32 %14, %14, %2\n v_xor_b32 %15, %15, %3\n"
            "s_setprio 3\n v_alignbit_b32 %12, %12, %12, 8\n v_alignbit_b32 %13, %13, %13, 8\n v_alignbit_b32 %14, %14, %14, 8\n v_alignbit_b32 %15, %15, %15, 8\n"
            "s_setprio 0\n v_add_u32 %8, %8, %12\n v_add_u32 %9, %9, %13\n v_add_u32 %10, %10, %14\n v_add_u32 %11, %11, %15\n"
            "v_xor_b32 %4, %4, %8\n v_xor_b32 %5, %5, %9\n v_xor_b32 %6, %6, %10\n v_xor_b32 %7, %7, %11\n"
            "s_setprio 3\n v_alignbit_b32 %4, %4, %4, 7\n v_alignbit_b32 %5, %5, %5, 7\n v_alignbit_b32 %6, %6, %6, 7\n v_alignbit_b32 %7, %7, %7, 7\n"
          : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3])
          : "v"(mx), "v"(my));
    }
  }
  stamp_end(rec, recs);
  uint32_t acc = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) acc ^= a[k] ^ b[k] ^ c[k] ^ d[k];
  if (acc == 0x12345678u) out[0] = acc;
}
static const int k_blake2s_G_ilv4_prio1_ipi = 16 * 12;

__global__ void __launch_bounds__(256) k_blake2s_G_ilv4_prio2(uint32_t* out, WaveRec* recs, int iters) {
  uint32_t a[4], b[4], c[4], d[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    a[k] = threadIdx.x + k;
    b[k] = threadIdx.x * 3u + k;
    c[k] = threadIdx.x * 5u + k;
    d[k] = threadIdx.x * 7u + k;
  }
  const uint32_t mx = threadIdx.x | 1u, my = blockIdx.x + 7u;
  WaveRec rec;
  __syncthreads();
  stamp_begin(rec);
  asm volatile("s_setprio 0");
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      asm volatile(
            "v_add3_u32 %0, %0, %4, %16\n v_add3_u32 %1, %1, %5, %16\n v_add3_u32 %2, %2, %6, %16\n v_add3_u32 %3, %3, %7, %16\n"
            "s_setprio 3\n v_xor_b32 %12, %12, %0\n v_xor_b32 %13, %13, %1\n v_xor_b32 %14, %14, %2\n v_xor_b32 %15, %15, %3\n"
            "s_setprio 0\n v_alignbit_b32 %12, %12, %12, 16\n v_alignbit_b32 %13, %13, %13, 16\n v_alignbit_b32 %14, %14, %14, 16\n v_alignbit_b32 %15, %15, %15, 16\n"
            "s_setprio 3\n v_add_u32 %8, %8, %12\n v_add_u32 %9, %9, %13\n v_add_u32 %10, %10, %14\n v_add_u32 %11, %11, %15\n"
            "v_xor_b32 %4, %4, %8\n v_xor_b32 %5, %5, %9\n v_xor_b32 %6, %6, %10\n v_xor_b32 %7, %7, %11\n"
            "s_setprio 0\n v_alignbit_b32 %4, %4, %4, 12\n v_alignbit_b32 %5, %5, %5, 12\n v_alignbit_b32 %6, %6, %6, 12\n v_alignbit_b32 %7, %7, %7, 12\n"
            "v_add3_u32 %0, %0, %4, %17\n v_add3_u32 %1, %1, %5, %17\n v_add3_u32 %2, %2, %6, %17\n v_add3_u32 %3, %3, %7, %17\n"
            "s_setprio 3\n v_xor_b32 %12, %12, %0\n v_xor_b32 %13, %13, %1\n v_xor_b32 %14, %14, %2\n v_xor_b32 %15, %15, %3\n"
            "s_setprio 0\n v_alignbit_b32 %12, %12, %12, 8\n v_alignbit_b32 %13, %13, %13, 8\n v_alignbit_b32 %14, %14, %14, 8\n v_alignbit_b32 %15, %15, %15, 8\n"
            "s_setprio 3\n v_add_u32 %8, %8, %12\n v_add_u32 %9, %9, %13\n v_add_u32 %10, %10, %14\n v_add_u32 %11, %11, %15\n"
            "v_xor_b32 %4, %4, %8\n v_xor_b32 %5, %5, %9\n v_xor_b32 %6, %6, %10\n v_xor_b32 %7, %7, %11\n"
            "s_setprio 0\n v_alignbit_b32 %4, %4, %4, 7\n v_alignbit_b32 %5, %5, %5, 7\n v_alignbit_b32 %6, %6, %6, 7\n v_alignbit_b32 %7, %7, %7, 7\n"
          : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3])
          : "v"(mx), "v"(my));
    }
  }
  stamp_end(rec, recs);
  uint32_t acc = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) acc ^= a[k] ^ b[k] ^ c[k] ^ d[k];
  if (acc == 0x12345678u) out[0] = acc;
}
static const int k_blake2s_G_ilv4_prio2_ipi = 16 * 12;

__global__ void __launch_bounds__(256) k_blake2s_G_ilv8_prio1(uint32_t* out, WaveRec* recs, int iters) {
  uint32_t a[8], b[8], c[8], d[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    a[k] = threadIdx.x + k;
    b[k] = threadIdx.x * 3u + k;
    c[k] = threadIdx.x * 5u + k;
    d[k] = threadIdx.x * 7u + k;
  }
  const uint32_t mx = threadIdx.x | 1u, my = blockIdx.x + 7u;
  WaveRec rec;
  __syncthreads();
  stamp_begin(rec);
  asm volatile("s_setprio 3");
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      asm volatile(
            "v_add3_u32 %0, %0, %8, %32\n v_add3_u32 %1, %1, %9, %32\n v_add3_u32 %2, %2, %10, %32\n v_add3_u32 %3, %3, %11, %32\n v_add3_u32 %4, %4, %12, %32\n v_add3_u32 %5, %5, %13, %32\n v_add3_u32 %6, %6, %14, %32\n v_add3_u32 %7, %7, %15, %32\n"
            "s_setprio 0\n v_xor_b32 %24, %24, %0\n v_xor_b32 %25, %25, %1\n v_xor_b32 %26, %26, %2\n v_xor_b32 %27, %27, %3\n v_xor_b32 %28, %28, %4\n v_xor_b32 %29, %29, %5\n v_xor_b32 %30, %30, %6\n v_xor_b32 %31, %31, %7\n"
            "s_setprio 3\n v_alignbit_b32 %24, %24, %24, 16\n v_alignbit_b32 %25, %25, %25, 16\n v_alignbit_b32 %26, %26, %26, 16\n v_alignbit_b32 %27, %27, %27, 16\n v_alignbit_b32 %28, %28, %28, 16\n v_alignbit_b32 %29, %29, %29, 16\n v_alignbit_b32 %30, %30, %30, 16\n v_alignbit_b32 %31, %31, %31, 16\n"
            "s_setprio 0\n v_add_u32 %16, %16, %24\n v_add_u32 %17, %17, %25\n v_add_u32 %18, %18, %26\n v_add_u32 %19, %19, %27\n v_add_u32 %20, %20, %28\n v_add_u32 %21, %21, %29\n v_add_u32 %22, %22, %30\n v_add_u32 %23, %23, %31\n"
            "v_xor_b32 %8, %8, %16\n v_xor_b32 %9, %9, %17\n v_xor_b32 %10, %10, %18\n v_xor_b32 %11, %11, %19\n v_xor_b32 %12, %12, %20\n v_xor_b32 %13, %13, %21\n v_xor_b32 %14, %14, %22\n v_xor_b32 %15, %15, %23\n"
            "s_setprio 3\n v_alignbit_b32 %8, %8, %8, 12\n v_alignbit_b32 %9, %9, %9, 12\n v_alignbit_b32 %10, %10, %10, 12\n v_alignbit_b32 %11, %11, %11, 12\n v_alignbit_b32 %12, %12, %12, 12\n v_alignbit_b32 %13, %13, %13, 12\n v_alignbit_b32 %14, %14, %14, 12\n v_alignbit_b32 %15, %15, %15, 12\n"
            "v_add3_u32 %0, %0, %8, %33\n v_add3_u32 %1, %1, %9, %33\n v_add3_u32 %2, %2, %10, %33\n v_add3_u32 %3, %3, %11, %33\n v_add3_u32 %4, %4, %12, %33\n v_add3_u32 %5, %5, %13, %33\n v_add3_u32 %6, %6, %14, %33\n v_add3_u32 %7, %7, %15, %33\n"
            "s_setprio 0\n v_xor_b32 %24, %24, %0\n v_xor_b32 %25, %25, %1\n v_xor_b32 %26, %26, %2\n v_xor_b32 %27, %27, %3\n v_xor_b32 %28, %28, %4\n v_xor_b32 %29, %29, %5\n v_xor_b32 %30, %30, %6\n v_xor_b32 %31, %31, %7\n"
            "s_setprio 3\n v_alignbit_b32 %24, %24, %24, 8\n v_alignbit_b32 %25, %25, %25, 8\n v_alignbit_b32 %26, %26, %26, 8\n v_alignbit_b32 %27, %27, %27, 8\n v_alignbit_b32 %28, %28, %28, 8\n v_alignbit_b32 %29, %29, %29, 8\n v_alignbit_b32 %30, %30, %30, 8\n v_alignbit_b32 %31, %31, %31, 8\n"
            "s_setprio 0\n v_add_u32 %16, %16, %24\n v_add_u32 %17, %17, %25\n v_add_u32 %18, %18, %26\n v_add_u32 %19, %19, %27\n v_add_u32 %20, %20, %28\n v_add_u32 %21, %21, %29\n v_add_u32 %22, %22, %30\n v_add_u32 %23, %23, %31\n"
            "v_xor_b32 %8, %8, %16\n v_xor_b32 %9, %9, %17\n v_xor_b32 %10, %10, %18\n v_xor_b32 %11, %11, %19\n v_xor_b32 %12, %12, %20\n v_xor_b32 %13, %13, %21\n v_xor_b32 %14, %14, %22\n v_xor_b32 %15, %15, %23\n"
            "s_setprio 3\n v_alignbit_b32 %8, %8, %8, 7\n v_alignbit_b32 %9, %9, %9, 7\n v_alignbit_b32 %10, %10, %10, 7\n v_alignbit_b32 %11, %11, %11, 7\n v_alignbit_b32 %12, %12, %12, 7\n v_alignbit_b32 %13, %13, %13, 7\n v_alignbit_b32 %14, %14, %14, 7\n v_alignbit_b32 %15, %15, %15, 7\n"
          : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(b[4]), "+v"(b[5]), "+v"(b[6]), "+v"(b[7]), "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(c[4]), "+v"(c[5]), "+v"(c[6]), "+v"(c[7]), "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]), "+v"(d[6]), "+v"(d[7])
          : "v"(mx), "v"(my));
    }
  }
  stamp_end(rec, recs);
  uint32_t acc = 0;
#pragma unroll
  for (int k = 0; k < 8; ++k) acc ^= a[k] ^ b[k] ^ c[k] ^ d[k];
  if (acc == 0x12345678u) out[0] = acc;
}
static const int k_blake2s_G_ilv8_prio1_ipi = 16 * 12;

__global__ void __launch_bounds__(256) k_blake2s_G_ilv8_prio2(uint32_t* out, WaveRec* recs, int iters) {
  uint32_t a[8], b[8], c[8], d[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    a[k] = threadIdx.x + k;
    b[k] = threadIdx.x * 3u + k;
    c[k] = threadIdx.x * 5u + k;
    d[k] = threadIdx.x * 7u + k;
  }
  const uint32_t mx = threadIdx.x | 1u, my = blockIdx.x + 7u;
  WaveRec rec;
  __syncthreads();
  stamp_begin(rec);
  asm volatile("s_setprio 0");
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      asm volatile(
            "v_add3_u32 %0, %0, %8, %32\n v_add3_u32 %1, %1, %9, %32\n v_add3_u32 %2, %2, %10, %32\n v_add3_u32 %3, %3, %11, %32\n v_add3_u32 %4, %4, %12, %32\n v_add3_u32 %5, %5, %13, %32\n v_add3_u32 %6, %6, %14, %32\n v_add3_u32 %7, %7, %15, %32\n"
            "s_setprio 3\n v_xor_b32 %24, %24, %0\n v_xor_b32 %25, %25, %1\n v_xor_b32 %26, %26, %2\n v_xor_b32 %27, %27, %3\n v_xor_b32 %28, %28, %4\n v_xor_b32 %29, %29, %5\n v_xor_b32 %30, %30, %6\n v_xor_b32 %31, %31, %7\n"
            "s_setprio 0\n v_alignbit_b32 %24, %24, %24, 16\n v_alignbit_b32 %25, %25, %25, 16\n v_alignbit_b32 %26, %26, %26, 16\n v_alignbit_b32 %27, %27, %27, 16\n v_alignbit_b32 %28, %28, %28, 16\n v_alignbit_b32 %29, %29, %29, 16\n v_alignbit_b32 %30, %30, %30, 16\n v_alignbit_b32 %31, %31, %31, 16\n"
            "s_setprio 3\n v_add_u32 %16, %16, %24\n v_add_u32 %17, %17, %25\n v_add_u32 %18, %18, %26\n v_add_u32 %19, %19, %27\n v_add_u32 %20, %20, %28\n v_add_u32 %21, %21, %29\n v_add_u32 %22, %22, %30\n v_add_u32 %23, %23, %31\n"
            "v_xor_b32 %8, %8, %16\n v_xor_b32 %9, %9, %17\n v_xor_b32 %10, %10, %18\n v_xor_b32 %11, %11, %19\n v_xor_b32 %12, %12, %20\n v_xor_b32 %13, %13, %21\n v_xor_b32 %14, %14, %22\n v_xor_b32 %15, %15, %23\n"
            "s_setprio 0\n v_alignbit_b32 %8, %8, %8, 12\n v_alignbit_b32 %9, %9, %9, 12\n v_alignbit_b32 %10, %10, %10, 12\n v_alignbit_b32 %11, %11, %11, 12\n v_alignbit_b32 %12, %12, %12, 12\n v_alignbit_b32 %13, %13, %13, 12\n v_alignbit_b32 %14, %14, %14, 12\n v_alignbit_b32 %15, %15, %15, 12\n"
            "v_add3_u32 %0, %0, %8, %33\n v_add3_u32 %1, %1, %9, %33\n v_add3_u32 %2, %2, %10, %33\n v_add3_u32 %3, %3, %11, %33\n v_add3_u32 %4, %4, %12, %33\n v_add3_u32 %5, %5, %13, %33\n v_add3_u32 %6, %6, %14, %33\n v_add3_u32 %7, %7, %15, %33\n"
            "s_setprio 3\n v_xor_b32 %24, %24, %0\n v_xor_b32 %25, %25, %1\n v_xor_b32 %26, %26, %2\n v_xor_b32 %27, %27, %3\n v_xor_b32 %28, %28, %4\n v_xor_b32 %29, %29, %5\n v_xor_b32 %30, %30, %6\n v_xor_b32 %31, %31, %7\n"
            "s_setprio 0\n v_alignbit_b32 %24, %24, %24, 8\n v_alignbit_b32 %25, %25, %25, 8\n v_alignbit_b32 %26, %26, %26, 8\n v_alignbit_b32 %27, %27, %27, 8\n v_alignbit_b32 %28, %28, %28, 8\n v_alignbit_b32 %29, %29, %29, 8\n v_alignbit_b32 %30, %30, %30, 8\n v_alignbit_b32 %31, %31, %31, 8\n"
            "s_setprio 3\n v_add_u32 %16, %16, %24\n v_add_u32 %17, %17, %25\n v_add_u32 %18, %18, %26\n v_add_u32 %19, %19, %27\n v_add_u32 %20, %20, %28\n v_add_u32 %21, %21, %29\n v_add_u32 %22, %22, %30\n v_add_u32 %23, %23, %31\n"
            "v_xor_b32 %8, %8, %16\n v_xor_b32 %9, %9, %17\n v_xor_b32 %10, %10, %18\n v_xor_b32 %11, %11, %19\n v_xor_b32 %12, %12, %20\n v_xor_b32 %13, %13, %21\n v_xor_b32 %14, %14, %22\n v_xor_b32 %15, %15, %23\n"
            "s_setprio 0\n v_alignbit_b32 %8, %8, %8, 7\n v_alignbit_b32 %9, %9, %9, 7\n v_alignbit_b32 %10, %10, %10, 7\n v_alignbit_b32 %11, %11, %11, 7\n v_alignbit_b32 %12, %12, %12, 7\n v_alignbit_b32 %13, %13, %13, 7\n v_alignbit_b32 %14, %14, %14, 7\n v_alignbit_b32 %15, %15, %15, 7\n"
          : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(b[4]), "+v"(b[5]), "+v"(b[6]), "+v"(b[7]), "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(c[4]), "+v"(c[5]), "+v"(c[6]), "+v"(c[7]), "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]), "+v"(d[6]), "+v"(d[7])
          : "v"(mx), "v"(my));
    }
  }
  stamp_end(rec, recs);
  uint32_t acc = 0;
#pragma unroll
  for (int k = 0; k < 8; ++k) acc ^= a[k] ^ b[k] ^ c[k] ^ d[k];
  if (acc == 0x12345678u) out[0] = acc;
}
static const int k_blake2s_G_ilv8_prio2_ipi = 16 * 12;


// the M31 butterfly of the FFT kernels (one multiplication by a twiddle, one add, one sub), 8 independent ones per lane and
// iteration.  MODE 0: as the compiler schedules field.h's formulas; MODE 1: the same arithmetic issued class by class
// (sched_barrier keeps the groups apart) with the port-0-only groups at raised wave priority.
template <int MODE>
__global__ void __launch_bounds__(256) k_bfly(uint32_t* out, WaveRec* recs, int iters) {
  constexpr uint32_t P = 0x7fffffffu;
  uint32_t a[8], b[8], tw[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    a[k] = (threadIdx.x * 2654435761u + k) & P;
    b[k] = (threadIdx.x * 40503u + 977u * k + blockIdx.x) & P;
    tw[k] = (threadIdx.x * 7919u + 31u * k + 5u) & P;
  }
  WaveRec rec;
  __syncthreads();
  stamp_begin(rec);
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        uint64_t p = (uint64_t)b[k] * tw[k];
        uint32_t s = (uint32_t)(p & P) + (uint32_t)(p >> 31);
        uint32_t m = s < s - P ? s : s - P;
        uint32_t u = a[k] + m, d = a[k] - m;
        a[k] = u < u - P ? u : u - P;
        b[k] = d < d + P ? d : d + P;
      }
    } else {
      uint64_t p[8];
      uint32_t hi[8], s[8], s2[8], m[8], u[8], u2[8], d[8], d2[8];
      __builtin_amdgcn_s_setprio(3);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int k = 0; k < 8; ++k) p[k] = (uint64_t)b[k] * tw[k];
#pragma unroll
      for (int k = 0; k < 8; ++k) hi[k] = (uint32_t)(p[k] >> 31);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        s[k] = ((uint32_t)p[k] & P) + hi[k];
        s2[k] = s[k] - P;
      }
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_setprio(3);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int k = 0; k < 8; ++k) m[k] = s[k] < s2[k] ? s[k] : s2[k];
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        u[k] = a[k] + m[k];
        u2[k] = u[k] - P;
        d[k] = a[k] - m[k];
        d2[k] = d[k] + P;
      }
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_setprio(3);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        a[k] = u[k] < u2[k] ? u[k] : u2[k];
        b[k] = d[k] < d2[k] ? d[k] : d2[k];
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    // next "layer": partners change (keeps every value live and dependent)
    uint32_t t = b[0];
#pragma unroll
    for (int k = 0; k < 7; ++k) b[k] = b[k + 1];
    b[7] = t;
  }
  stamp_end(rec, recs);
  uint32_t acc = 0;
#pragma unroll
  for (int k = 0; k < 8; ++k) acc ^= a[k] ^ b[k];
  if (acc == 0x12345678u) out[0] = acc;
}

#define CHECK(x)                                                                  \
  do {                                                                            \
    hipError_t e_ = (x);                                                          \
    if (e_ != hipSuccess) {                                                       \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                     \
      exit(1);                                                                    \
    }                                                                             \
  } while (0)

typedef void (*kern_t)(uint32_t*, WaveRec*, int);

static void run(const char* label, kern_t k, int ipi, int W, int n_cu, int iters, uint32_t* d_out, WaveRec* d_recs,
                bool verbose) {
  const int blocks = n_cu * W;  // 256-thread workgroups: one wave per SIMD each, W workgroups per CU
  const int waves = blocks * 4;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d_out, d_recs, 16);  // warm-up (code object, clocks)
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0, 0));
  hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d_out, d_recs, iters);
  CHECK(hipEventRecord(e1, 0));
  CHECK(hipDeviceSynchronize());
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<WaveRec> r(waves);
  CHECK(hipMemcpy(r.data(), d_recs, sizeof(WaveRec) * waves, hipMemcpyDeviceToHost));
  const double instr_per_wave = (double)ipi * iters;
  struct Simd {
    int n = 0;
    uint64_t first = ~0ull, last = 0, tmin = ~0ull, tmax = 0;
    double sum_elapsed = 0;
  };
  std::map<uint64_t, Simd> simds;
  std::map<uint64_t, int> cus, xccs;
  double clk_ticks = 0, rt_ticks = 0;
  for (auto& w : r) {
    const uint64_t simd = (w.hw_id >> 4) & 3u, loc = (w.hw_id >> 8) & 0xFFu;  // cu_id[11:8] sh_id[12] se_id[15:13]
    const uint64_t cu_key = ((uint64_t)(w.xcc_id & 0xF) << 8) | loc;
    Simd& s = simds[(cu_key << 2) | simd];
    s.n++;
    s.first = std::min(s.first, w.t0);
    s.last = std::max(s.last, w.t1);
    s.tmin = std::min(s.tmin, w.t1 - w.t0);
    s.tmax = std::max(s.tmax, w.t1 - w.t0);
    s.sum_elapsed += (double)(w.t1 - w.t0);
    cus[cu_key]++;
    xccs[w.xcc_id & 0xF]++;
    clk_ticks += (double)(w.t1 - w.t0);
    rt_ticks += (double)(w.r1 - w.r0);
  }
  const double mhz = clk_ticks / rt_ticks * 100.0;  // s_memrealtime ticks at 100 MHz
  double per_wave = 0, makespan = 0, spread = 0;
  int nmin = 1 << 30, nmax = 0;
  for (auto& kv : simds) {
    const Simd& s = kv.second;
    per_wave += s.n * instr_per_wave / (s.sum_elapsed / s.n);
    makespan += s.n * instr_per_wave / (double)(s.last - s.first);
    spread += (double)s.tmin / (double)s.tmax;
    nmin = std::min(nmin, s.n);
    nmax = std::max(nmax, s.n);
  }
  per_wave /= simds.size();
  makespan /= simds.size();
  spread /= simds.size();
  const double wall = (double)waves * instr_per_wave / (ms * 1e-3 * mhz * 1e6) / (double)simds.size();
  printf("%-18s w=%d  per-wave %.3f  makespan %.3f  wall %.3f   [clk %4.0f MHz, %.3f ms; %zu XCCs, %zu CUs, %zu SIMDs, "
         "waves/SIMD %d..%d, shortest/longest wave on a SIMD %.2f]\n",
         label, W, per_wave, makespan, wall, mhz, ms, xccs.size(), cus.size(), simds.size(), nmin, nmax, spread);
  if (verbose) {
    // one SIMD in detail: every wave's start and end relative to the SIMD's first start
    const auto& kv = *simds.begin();
    printf("    SIMD %llx:", (unsigned long long)kv.first);
    std::vector<std::pair<uint64_t, uint64_t>> ws;
    for (auto& w : r) {
      const uint64_t simd = (w.hw_id >> 4) & 3u, loc = (w.hw_id >> 8) & 0xFFu;
      if (((((uint64_t)(w.xcc_id & 0xF) << 8) | loc) << 2 | simd) == kv.first) ws.push_back({w.t0, w.t1});
    }
    std::sort(ws.begin(), ws.end());
    for (auto& p : ws) printf("  [%llu..%llu]", (unsigned long long)(p.first - kv.second.first), (unsigned long long)(p.second - kv.second.first));
    printf("  cycles\n");
  }
  CHECK(hipEventDestroy(e0));
  CHECK(hipEventDestroy(e1));
}

int main(int argc, char** argv) {
  const bool pmc = argc > 1 && !strcmp(argv[1], "pmc");
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int n_cu = prop.multiProcessorCount;
  uint32_t* d_out;
  WaveRec* d_recs;
  CHECK(hipMalloc(&d_out, 64));
  CHECK(hipMalloc(&d_recs, sizeof(WaveRec) * (size_t)n_cu * 8 * 4));
  printf("device %s  CUs %d  nominal clock %d MHz\n", prop.name, n_cu, prop.clockRate / 1000);
  printf("wave-instructions / shader clock / SIMD.  per-wave: from the MEAN elapsed s_memtime of a SIMD's waves; makespan: from "
         "first start to last end on the SIMD; wall: HIP events x measured clock\n");
  struct S {
    const char* label;
    kern_t k;
    int ipi;
  } streams[] = {
      {"add_u32", k_add_u32, k_add_u32_ipi},
      {"fma_f32", k_fma_f32, k_fma_f32_ipi},
      {"xor distinct-src", k_xor_distinct, k_xor_distinct_ipi},
      {"add3_u32", k_add3_u32, k_add3_u32_ipi},
      {"add3 distinct-src", k_add3_distinct, k_add3_distinct_ipi},
      {"alignbit_b32", k_alignbit_b32, k_alignbit_b32_ipi},
      {"blake2s_G", k_blake2s_G, k_blake2s_G_ipi},
      {"blake2s_G plain", k_blake2s_G_plain, k_blake2s_G_plain_ipi},
  };
  if (argc > 1 && !strcmp(argv[1], "ilv")) {
    S more[] = {
        {"add dep x1", k_add_dep1, k_add_dep1_ipi},
        {"add dep x2", k_add_dep2, k_add_dep2_ipi},
        {"add dep x4", k_add_dep4, k_add_dep4_ipi},
        {"add_u32 (x8)", k_add_u32, k_add_u32_ipi},
        {"mix add3:xor", k_mix_add3_xor, k_mix_add3_xor_ipi},
        {"blake2s_G", k_blake2s_G, k_blake2s_G_ipi},
        {"blake2s_G ilv4", k_blake2s_G_ilv4, k_blake2s_G_ilv4_ipi},
        {"blake2s_G ilv8", k_blake2s_G_ilv8, k_blake2s_G_ilv8_ipi},
        {"blake2s_G plain", k_blake2s_G_plain, k_blake2s_G_plain_ipi},
        {"blake2s_G pl ilv4", k_blake2s_G_plain_ilv4, k_blake2s_G_plain_ilv4_ipi},
    };
    for (auto& s : more) {
      const int iters = (int)(4096ll * 64 / s.ipi);
      for (int W : {1, 2, 4, 8}) run(s.label, s.k, s.ipi, W, n_cu, iters, d_out, d_recs, false);
    }
    return 0;
  }
  if (argc > 1 && !strcmp(argv[1], "why")) {
    S more[] = {
        {"or_b32", k_or_b32, k_or_b32_ipi},
        {"lshlrev const", k_lshl_c, k_lshl_c_ipi},
        {"lshrrev const", k_lshr_c, k_lshr_c_ipi},
        {"lshr->tmp, xor", k_lshr_2reg, k_lshr_2reg_ipi},
        {"rot sequence", k_rot_seq, k_rot_seq_ipi},
        {"add_u32 4KB body", k_add_u32_big, k_add_u32_big_ipi},
        {"add dep 4KB body", k_add_dep1_big, k_add_dep1_big_ipi},
    };
    for (auto& s : more) {
      const int iters = (int)(4096ll * 64 / s.ipi);
      for (int W : {1, 2, 4, 8}) run(s.label, s.k, s.ipi, W, n_cu, iters, d_out, d_recs, false);
    }
    return 0;
  }
  if (argc > 1 && !strcmp(argv[1], "prio")) {
    S more[] = {
        {"phase 8 prio-", k_phase<0, 8>, 16},      {"phase 8 slow-hi", k_phase<1, 8>, 16},      {"phase 8 fast-hi", k_phase<2, 8>, 16},
        {"phase 32 prio-", k_phase<0, 32>, 64},    {"phase 32 slow-hi", k_phase<1, 32>, 64},    {"phase 32 fast-hi", k_phase<2, 32>, 64},
        {"phase 128 prio-", k_phase<0, 128>, 256}, {"phase 128 slow-hi", k_phase<1, 128>, 256}, {"phase 128 fast-hi", k_phase<2, 128>, 256},
        {"G ilv4", k_blake2s_G_ilv4, k_blake2s_G_ilv4_ipi},
        {"G ilv4 slow-hi", k_blake2s_G_ilv4_prio1, k_blake2s_G_ilv4_prio1_ipi},
        {"G ilv4 fast-hi", k_blake2s_G_ilv4_prio2, k_blake2s_G_ilv4_prio2_ipi},
        {"G ilv8", k_blake2s_G_ilv8, k_blake2s_G_ilv8_ipi},
        {"G ilv8 slow-hi", k_blake2s_G_ilv8_prio1, k_blake2s_G_ilv8_prio1_ipi},
        {"G ilv8 fast-hi", k_blake2s_G_ilv8_prio2, k_blake2s_G_ilv8_prio2_ipi},
    };
    for (auto& s : more) {
      const int iters = (int)(4096ll * 64 / s.ipi);
      for (int W : {2, 4, 8}) run(s.label, s.k, s.ipi, W, n_cu, iters, d_out, d_recs, false);
    }
    return 0;
  }
  if (argc > 1 && !strcmp(argv[1], "bfly")) {
    // "instructions" per iteration = 8 butterflies: the printed rates are BUTTERFLIES per clock per SIMD (wave level)
    S more[] = {{"bfly compiler", k_bfly<0>, 8}, {"bfly phased", k_bfly<1>, 8}};
    for (auto& s : more)
      for (int W : {1, 2, 4, 8}) run(s.label, s.k, s.ipi, W, n_cu, 2048, d_out, d_recs, false);
    return 0;
  }
  if (argc > 1 && !strcmp(argv[1], "pmc2")) {  // the co-issue streams under rocprofv3 --pmc, 8 waves per SIMD
    S more[] = {
        {"G ilv4", k_blake2s_G_ilv4, k_blake2s_G_ilv4_ipi},
        {"G ilv4 slow-hi", k_blake2s_G_ilv4_prio1, k_blake2s_G_ilv4_prio1_ipi},
        {"phase 128 prio-", k_phase<0, 128>, 256},
        {"phase 128 slow-hi", k_phase<1, 128>, 256},
    };
    for (auto& s : more) run(s.label, s.k, s.ipi, 8, n_cu, (int)(4096ll * 64 / s.ipi), d_out, d_recs, false);
    return 0;
  }
  for (auto& s : streams) {
    const int iters = (int)(4096ll * 64 / s.ipi);
    if (pmc) {
      run(s.label, s.k, s.ipi, 8, n_cu, iters, d_out, d_recs, false);
      continue;
    }
    for (int W : {1, 2, 4, 8}) run(s.label, s.k, s.ipi, W, n_cu, iters, d_out, d_recs, W == 8 && (s.k == k_add3_u32 || s.k == k_blake2s_G));
  }
  return 0;
}

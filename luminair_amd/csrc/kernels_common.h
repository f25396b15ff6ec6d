// Shared preamble of the kernel translation units (kernels_*.hip): block size, the serial-path wave priority, the QM31
// issue-phase experiment switch.
#pragma once
#include <cmath>

#include <algorithm>

#include "kernels.h"
#include "issue_phases.h"
#include "fft_fixed.h"
#include "launch_util.h"

namespace lmn {

constexpr int TPB = 256;

// Wave priority of the short kernels that sit on a proof's serial path (tree tops, FRI tail, scans, small reductions):
// with several proofs in flight their few waves otherwise wait behind every older wave of the big kernels.
#ifndef LMN_SERIAL_PRIO
#define LMN_SERIAL_PRIO 0
#endif
#if defined(__HIP_DEVICE_COMPILE__) && !defined(LMN_EMU)
#define LMN_SERIAL_KERNEL()                                              \
  do {                                                                   \
    if (LMN_SERIAL_PRIO) __builtin_amdgcn_s_setprio(LMN_SERIAL_PRIO);    \
  } while (0)
#else
#define LMN_SERIAL_KERNEL() do { } while (0)
#endif

static inline unsigned cdiv(uint64_t a, uint64_t b) { return (unsigned)((a + b - 1) / b); }

// Experiment switch (tools/build_variants.sh qphase "-DLMN_QM31_PHASES"): the 16 multiply-accumulates of a QM31 product and
// the multiply-accumulate runs of the lazy dot products issued as first-port phases (issue_phases.h), as the Blake2s and
// butterfly code does.
#if defined(LMN_QM31_PHASES) && defined(__HIP_DEVICE_COMPILE__) && !defined(LMN_EMU)
LMN_D QM31 q_mul_phased(QM31 x, QM31 y) {
  const uint32_t nb = P31 - x.b, nd = P31 - x.d;
  const uint32_t c2 = m_dbl(x.c), d2 = m_dbl(x.d);
  const uint32_t e1 = m_sub(c2, x.d);
  const uint32_t e2 = P31 - m_add(d2, x.c);
  const uint32_t f1 = m_add(x.c, d2);
  LMN_PHASE_PORT0();
  const uint64_t lre = (uint64_t)x.a * y.a + (uint64_t)nb * y.b + (uint64_t)e1 * y.c + (uint64_t)e2 * y.d;
  const uint64_t lim = (uint64_t)x.a * y.b + (uint64_t)x.b * y.a + (uint64_t)f1 * y.c + (uint64_t)e1 * y.d;
  const uint64_t hre = (uint64_t)x.a * y.c + (uint64_t)nb * y.d + (uint64_t)x.c * y.a + (uint64_t)nd * y.b;
  const uint64_t him = (uint64_t)x.a * y.d + (uint64_t)x.b * y.c + (uint64_t)x.c * y.b + (uint64_t)x.d * y.a;
  LMN_PHASE_ANY();
  return {m_red64(lre), m_red64(lim), m_red64(hre), m_red64(him)};
}
#define q_mul q_mul_phased
#define LMN_QPHASE_PORT0() LMN_PHASE_PORT0()
#define LMN_QPHASE_ANY() LMN_PHASE_ANY()
#else
#define LMN_QPHASE_PORT0() do { } while (0)
#define LMN_QPHASE_ANY() do { } while (0)
#endif

// Word `lane_word` of a column whose base address is wave-uniform (a kernel argument + a uniform column offset, or a
// pointer read through the scalar cache): a raw buffer access - the base travels in an SGPR resource built by scalar
// instructions, the lane's part of the address is one 32-bit byte offset (rows < 2^27: lane_word * 4 < 2^30) - so no
// 64-bit vector address is formed per access (a per-lane pointer + a uniform offset cost one v_lshl_add_u64 per
// load; the SGPR-base form of global_load is lost as soon as a zero-extension is hoisted out of a loop, since
// instruction selection works per basic block).  As fft_fixed.hip's GTile.
LMN_D uint32_t ld_ub(const uint32_t* __restrict__ uniform_base, uint32_t lane_word) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(LMN_EMU)
  const __amdgpu_buffer_rsrc_t r =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(uniform_base), (short)0, (int)0xffffffff, 0x00020000);
  return __builtin_amdgcn_raw_buffer_load_b32(r, (int)(lane_word << 2), 0, 0);
#else
  return uniform_base[lane_word];
#endif
}
// The same with a uniform word offset next to the base (column k of a group at base + k * stride): ONE resource for the
// group, the column's offset in an SGPR (soffset) - a resource per column costs four SGPRs each and spills them once a
// dozen are live.  uniform_words * 4 + lane_word * 4 must stay below 2^32 - 4: callers rebase beyond that.
LMN_D uint32_t ld_ubs(const uint32_t* __restrict__ uniform_base, uint32_t uniform_words, uint32_t lane_word) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(LMN_EMU)
  const __amdgpu_buffer_rsrc_t r =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(uniform_base), (short)0, (int)0xffffffff, 0x00020000);
  return __builtin_amdgcn_raw_buffer_load_b32(r, (int)(lane_word << 2), (int)(uniform_words << 2), 0);
#else
  return uniform_base[(uint64_t)uniform_words + lane_word];
#endif
}
// column k of a group of columns `stride` words apart, row `lane_word` (< stride): one resource per 8 columns, so that
// soffset + voffset <= 8 * stride * 4 - 4 stays inside the resource's 2^32 - 1 bytes for every stride a multi-column
// group can have (<= 2^26: the evaluation domain of the largest admissible table; the 2^27-row trees have 4 columns)
LMN_D uint32_t ld_col(const uint32_t* __restrict__ group_base, int k, uint64_t stride, uint32_t lane_word) {
  return ld_ubs(group_base + (uint64_t)(k & ~7) * stride, (uint32_t)(k & 7) * (uint32_t)stride, lane_word);
}
LMN_D void st_ub(uint32_t* __restrict__ uniform_base, uint32_t lane_word, uint32_t v) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(LMN_EMU)
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(uniform_base, (short)0, (int)0xffffffff, 0x00020000);
  __builtin_amdgcn_raw_buffer_store_b32(v, r, (int)(lane_word << 2), 0, 0);
#else
  uniform_base[lane_word] = v;
#endif
}
// words 2 * lane_pair and 2 * lane_pair + 1 (one 8-byte access: a FRI fold's pair of neighbours)
LMN_D void ld_ub_pair(const uint32_t* __restrict__ uniform_base, uint32_t lane_pair, uint32_t& even, uint32_t& odd) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(LMN_EMU)
  typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
  const __amdgpu_buffer_rsrc_t r =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(uniform_base), (short)0, (int)0xffffffff, 0x00020000);
  const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(r, (int)(lane_pair << 3), 0, 0);
  even = v.x;
  odd = v.y;
#else
  even = uniform_base[2ull * lane_pair];
  odd = uniform_base[2ull * lane_pair + 1];
#endif
}
LMN_D QM31 load_secure_ub(const uint32_t* __restrict__ uniform_base, uint64_t stride, uint32_t lane_word) {
  return QM31{ld_ub(uniform_base, lane_word), ld_ub(uniform_base + stride, lane_word), ld_ub(uniform_base + 2 * stride, lane_word),
              ld_ub(uniform_base + 3 * stride, lane_word)};
}

LMN_D QM31 load_secure_col(const uint32_t* __restrict__ base, uint64_t stride, uint64_t i) {
  return QM31{base[i], base[stride + i], base[2 * stride + i], base[3 * stride + i]};
}


}  // namespace lmn

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

LMN_D QM31 load_secure_col(const uint32_t* __restrict__ base, uint64_t stride, uint64_t i) {
  return QM31{base[i], base[stride + i], base[2 * stride + i], base[3 * stride + i]};
}


}  // namespace lmn

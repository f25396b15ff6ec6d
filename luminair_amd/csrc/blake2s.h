// Blake2s-256 (RFC 7693, unkeyed) compression shared by host transcript code and gfx950 kernels.
// Replaces the `blake2` 0.10.6 crate + stwo `core/vcs/blake2_hash.rs` used behind
// /root/reference/crates/prover/src/prover.rs:44-46 (Blake2sChannel, Blake2sMerkleChannel).
#pragma once
#include "platform.h"

namespace lmn {

struct Hash32 {
  uint32_t w[8];
};

LMN_HD uint32_t b2_rotr(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }

#define LMN_B2_G(a, b, c, d, x, y) \
  a = a + b + (x);                 \
  d = b2_rotr(d ^ a, 16);          \
  c = c + d;                       \
  b = b2_rotr(b ^ c, 12);          \
  a = a + b + (y);                 \
  d = b2_rotr(d ^ a, 8);           \
  c = c + d;                       \
  b = b2_rotr(b ^ c, 7);

#define LMN_B2_ROUND(s0, s1, s2, s3, s4, s5, s6, s7, s8, s9, s10, s11, s12, s13, s14, s15) \
  LMN_B2_G(v0, v4, v8, v12, m[s0], m[s1])                                                  \
  LMN_B2_G(v1, v5, v9, v13, m[s2], m[s3])                                                  \
  LMN_B2_G(v2, v6, v10, v14, m[s4], m[s5])                                                 \
  LMN_B2_G(v3, v7, v11, v15, m[s6], m[s7])                                                 \
  LMN_B2_G(v0, v5, v10, v15, m[s8], m[s9])                                                 \
  LMN_B2_G(v1, v6, v11, v12, m[s10], m[s11])                                               \
  LMN_B2_G(v2, v7, v8, v13, m[s12], m[s13])                                                \
  LMN_B2_G(v3, v4, v9, v14, m[s14], m[s15])

#if defined(__HIP_DEVICE_COMPILE__) && !defined(LMN_EMU) && !defined(LMN_B2_NO_PRIO)
// gfx950 issue order for one half round (four independent quarter rounds), written out as instructions.
// Measured (profiles/ceilings/valu_coissue_two_ports.txt, tools/microbench_reconcile.hip `prio`): a SIMD issues up to two VALU
// instructions per 4-cycle slot from two different waves, but v_add3_u32 / v_alignbit_b32 (and every other three-operand
// or multiplier op) can only take the first of the two ports, and the arbiter hands that port to the OLDEST wave whatever
// it is about to issue - so co-resident waves that execute a quarter round as the compiler schedules it run at 0.25
// instructions/clk/SIMD.  Grouping the four quarter rounds' instructions by class and raising the wave's priority
// (s_setprio) while it issues the port-0-only class lets another wave's plain ops take the second port: 0.42/clk/SIMD.
template <int LO, int HI>
__device__ __forceinline__ void b2_half(uint32_t& a0, uint32_t& a1, uint32_t& a2, uint32_t& a3, uint32_t& b0, uint32_t& b1,
                                        uint32_t& b2, uint32_t& b3, uint32_t& c0, uint32_t& c1, uint32_t& c2, uint32_t& c3,
                                        uint32_t& d0, uint32_t& d1, uint32_t& d2, uint32_t& d3, uint32_t x0, uint32_t y0,
                                        uint32_t x1, uint32_t y1, uint32_t x2, uint32_t y2, uint32_t x3, uint32_t y3) {
  asm volatile(
      "v_add3_u32 %0, %0, %4, %16\n v_add3_u32 %1, %1, %5, %18\n v_add3_u32 %2, %2, %6, %20\n v_add3_u32 %3, %3, %7, %22\n"
      "s_setprio %24\n"
      "v_xor_b32 %12, %12, %0\n v_xor_b32 %13, %13, %1\n v_xor_b32 %14, %14, %2\n v_xor_b32 %15, %15, %3\n"
      "s_setprio %25\n"
      "v_alignbit_b32 %12, %12, %12, 16\n v_alignbit_b32 %13, %13, %13, 16\n v_alignbit_b32 %14, %14, %14, 16\n"
      "v_alignbit_b32 %15, %15, %15, 16\n"
      "s_setprio %24\n"
      "v_add_u32 %8, %8, %12\n v_add_u32 %9, %9, %13\n v_add_u32 %10, %10, %14\n v_add_u32 %11, %11, %15\n"
      "v_xor_b32 %4, %4, %8\n v_xor_b32 %5, %5, %9\n v_xor_b32 %6, %6, %10\n v_xor_b32 %7, %7, %11\n"
      "s_setprio %25\n"
      "v_alignbit_b32 %4, %4, %4, 12\n v_alignbit_b32 %5, %5, %5, 12\n v_alignbit_b32 %6, %6, %6, 12\n"
      "v_alignbit_b32 %7, %7, %7, 12\n"
      "v_add3_u32 %0, %0, %4, %17\n v_add3_u32 %1, %1, %5, %19\n v_add3_u32 %2, %2, %6, %21\n v_add3_u32 %3, %3, %7, %23\n"
      "s_setprio %24\n"
      "v_xor_b32 %12, %12, %0\n v_xor_b32 %13, %13, %1\n v_xor_b32 %14, %14, %2\n v_xor_b32 %15, %15, %3\n"
      "s_setprio %25\n"
      "v_alignbit_b32 %12, %12, %12, 8\n v_alignbit_b32 %13, %13, %13, 8\n v_alignbit_b32 %14, %14, %14, 8\n"
      "v_alignbit_b32 %15, %15, %15, 8\n"
      "s_setprio %24\n"
      "v_add_u32 %8, %8, %12\n v_add_u32 %9, %9, %13\n v_add_u32 %10, %10, %14\n v_add_u32 %11, %11, %15\n"
      "v_xor_b32 %4, %4, %8\n v_xor_b32 %5, %5, %9\n v_xor_b32 %6, %6, %10\n v_xor_b32 %7, %7, %11\n"
      "s_setprio %25\n"
      "v_alignbit_b32 %4, %4, %4, 7\n v_alignbit_b32 %5, %5, %5, 7\n v_alignbit_b32 %6, %6, %6, 7\n"
      "v_alignbit_b32 %7, %7, %7, 7\n"
      // early clobber: the message words are still read after the first state registers have been written
      : "+&v"(a0), "+&v"(a1), "+&v"(a2), "+&v"(a3), "+&v"(b0), "+&v"(b1), "+&v"(b2), "+&v"(b3), "+&v"(c0), "+&v"(c1),
        "+&v"(c2), "+&v"(c3), "+&v"(d0), "+&v"(d1), "+&v"(d2), "+&v"(d3)
      : "v"(x0), "v"(y0), "v"(x1), "v"(y1), "v"(x2), "v"(y2), "v"(x3), "v"(y3), "n"(LO), "n"(HI));
}


#undef LMN_B2_ROUND
#define LMN_B2_ROUND(s0, s1, s2, s3, s4, s5, s6, s7, s8, s9, s10, s11, s12, s13, s14, s15)                               \
  b2_half<LO, HI>(v0, v1, v2, v3, v4, v5, v6, v7, v8, v9, v10, v11, v12, v13, v14, v15, m[s0], m[s1], m[s2], m[s3],     \
                  m[s4], m[s5], m[s6], m[s7]);                                                                           \
  b2_half<LO, HI>(v0, v1, v2, v3, v5, v6, v7, v4, v10, v11, v8, v9, v15, v12, v13, v14, m[s8], m[s9], m[s10], m[s11],    \
                  m[s12], m[s13], m[s14], m[s15]);
#define LMN_B2_ENTER asm volatile("s_setprio %0" ::"n"(HI));
#define LMN_B2_LEAVE asm volatile("s_setprio %0" ::"n"(LO));
#else
#define LMN_B2_ENTER
#define LMN_B2_LEAVE
#endif

// h <- F(h, m, t, f0).  The sigma schedule is unrolled so message words stay in registers.  LO / HI: wave priority
// while issuing the any-port / first-port-only instruction runs on the device (LO == HI: constant priority).
#ifndef LMN_B2_PRIO_HI
#define LMN_B2_PRIO_HI 3
#endif
template <int LO = 0, int HI = LMN_B2_PRIO_HI>
LMN_HD void b2_compress(uint32_t h[8], const uint32_t m[16], uint32_t t0, uint32_t f0) {
  uint32_t v0 = h[0], v1 = h[1], v2 = h[2], v3 = h[3], v4 = h[4], v5 = h[5], v6 = h[6], v7 = h[7];
  uint32_t v8 = 0x6A09E667u, v9 = 0xBB67AE85u, v10 = 0x3C6EF372u, v11 = 0xA54FF53Au;
  uint32_t v12 = 0x510E527Fu ^ t0, v13 = 0x9B05688Cu, v14 = 0x1F83D9ABu ^ f0, v15 = 0x5BE0CD19u;
  LMN_B2_ENTER
  LMN_B2_ROUND(0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15)
  LMN_B2_ROUND(14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3)
  LMN_B2_ROUND(11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4)
  LMN_B2_ROUND(7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8)
  LMN_B2_ROUND(9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13)
  LMN_B2_ROUND(2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9)
  LMN_B2_ROUND(12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11)
  LMN_B2_ROUND(13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10)
  LMN_B2_ROUND(6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5)
  LMN_B2_ROUND(10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0)
  LMN_B2_LEAVE
  h[0] ^= v0 ^ v8;
  h[1] ^= v1 ^ v9;
  h[2] ^= v2 ^ v10;
  h[3] ^= v3 ^ v11;
  h[4] ^= v4 ^ v12;
  h[5] ^= v5 ^ v13;
  h[6] ^= v6 ^ v14;
  h[7] ^= v7 ^ v15;
}

// out <- F(h0, m, t0, final) for a FRESH hash (h0 = the unkeyed 32-byte-digest initial state) whose single block is also
// its last: every Merkle node and every <= 16-column leaf.  On the device the first column half round takes the initial
// state as LITERAL operands - a = (h_a + h_b) + x, d = K_d ^ a, c = K_c + d, b = K_b ^ c - so the 16 state registers are
// first written by arithmetic instead of 16 constant moves in front of every compression (the asm operands of b2_half
// are read-write); instruction classes and phases as in b2_half.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(LMN_EMU) && !defined(LMN_B2_NO_PRIO) && !defined(LMN_B2_NO_FRESH)
template <int LO, int HI>
__device__ __forceinline__ void b2_half_first(uint32_t& a0, uint32_t& a1, uint32_t& a2, uint32_t& a3, uint32_t& b0, uint32_t& b1,
                                              uint32_t& b2, uint32_t& b3, uint32_t& c0, uint32_t& c1, uint32_t& c2, uint32_t& c3,
                                              uint32_t& d0, uint32_t& d1, uint32_t& d2, uint32_t& d3, uint32_t x0, uint32_t y0,
                                              uint32_t x1, uint32_t y1, uint32_t x2, uint32_t y2, uint32_t x3, uint32_t y3,
                                              uint32_t kd0 /* IV4 ^ t0, wave-uniform */) {
  constexpr uint32_t H0 = 0x6A09E667u ^ 0x01010020u, H1 = 0xBB67AE85u, H2 = 0x3C6EF372u, H3 = 0xA54FF53Au;
  constexpr uint32_t H4 = 0x510E527Fu, H5 = 0x9B05688Cu, H6 = 0x1F83D9ABu, H7 = 0x5BE0CD19u;
  asm volatile(
      "s_setprio %33\n"
      "v_add_u32 %0, %25, %16\n v_add_u32 %1, %26, %18\n v_add_u32 %2, %27, %20\n v_add_u32 %3, %28, %22\n"
      "v_xor_b32 %12, %24, %0\n v_xor_b32 %13, %29, %1\n v_xor_b32 %14, %30, %2\n v_xor_b32 %15, %31, %3\n"
      "s_setprio %34\n"
      "v_alignbit_b32 %12, %12, %12, 16\n v_alignbit_b32 %13, %13, %13, 16\n v_alignbit_b32 %14, %14, %14, 16\n"
      "v_alignbit_b32 %15, %15, %15, 16\n"
      "s_setprio %33\n"
      "v_add_u32 %8, 0x6A09E667, %12\n v_add_u32 %9, 0xBB67AE85, %13\n v_add_u32 %10, 0x3C6EF372, %14\n"
      "v_add_u32 %11, 0xA54FF53A, %15\n"
      "v_xor_b32 %4, 0x510E527F, %8\n v_xor_b32 %5, 0x9B05688C, %9\n v_xor_b32 %6, 0x1F83D9AB, %10\n"
      "v_xor_b32 %7, 0x5BE0CD19, %11\n"
      "s_setprio %34\n"
      "v_alignbit_b32 %4, %4, %4, 12\n v_alignbit_b32 %5, %5, %5, 12\n v_alignbit_b32 %6, %6, %6, 12\n"
      "v_alignbit_b32 %7, %7, %7, 12\n"
      "v_add3_u32 %0, %0, %4, %17\n v_add3_u32 %1, %1, %5, %19\n v_add3_u32 %2, %2, %6, %21\n v_add3_u32 %3, %3, %7, %23\n"
      "s_setprio %33\n"
      "v_xor_b32 %12, %12, %0\n v_xor_b32 %13, %13, %1\n v_xor_b32 %14, %14, %2\n v_xor_b32 %15, %15, %3\n"
      "s_setprio %34\n"
      "v_alignbit_b32 %12, %12, %12, 8\n v_alignbit_b32 %13, %13, %13, 8\n v_alignbit_b32 %14, %14, %14, 8\n"
      "v_alignbit_b32 %15, %15, %15, 8\n"
      "s_setprio %33\n"
      "v_add_u32 %8, %8, %12\n v_add_u32 %9, %9, %13\n v_add_u32 %10, %10, %14\n v_add_u32 %11, %11, %15\n"
      "v_xor_b32 %4, %4, %8\n v_xor_b32 %5, %5, %9\n v_xor_b32 %6, %6, %10\n v_xor_b32 %7, %7, %11\n"
      "s_setprio %34\n"
      "v_alignbit_b32 %4, %4, %4, 7\n v_alignbit_b32 %5, %5, %5, 7\n v_alignbit_b32 %6, %6, %6, 7\n"
      "v_alignbit_b32 %7, %7, %7, 7\n"
      : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3), "=&v"(b0), "=&v"(b1), "=&v"(b2), "=&v"(b3), "=&v"(c0), "=&v"(c1),
        "=&v"(c2), "=&v"(c3), "=&v"(d0), "=&v"(d1), "=&v"(d2), "=&v"(d3)
      : "v"(x0), "v"(y0), "v"(x1), "v"(y1), "v"(x2), "v"(y2), "v"(x3), "v"(y3), "s"(kd0), "n"(H0 + H4), "n"(H1 + H5),
        "n"(H2 + H6), "n"(H3 + H7), "n"(H5), "n"(H6 ^ 0xffffffffu), "n"(H7), "n"(0), "n"(LO), "n"(HI));
}
#define LMN_B2_HAVE_FRESH 1
#endif

LMN_HD void b2_init(uint32_t h[8]);
template <int LO = 0, int HI = LMN_B2_PRIO_HI>
LMN_HD void b2_compress_fresh(uint32_t out[8], const uint32_t m[16], uint32_t t0) {
#ifdef LMN_B2_HAVE_FRESH
  uint32_t v0, v1, v2, v3, v4, v5, v6, v7, v8, v9, v10, v11, v12, v13, v14, v15;
  b2_half_first<LO, HI>(v0, v1, v2, v3, v4, v5, v6, v7, v8, v9, v10, v11, v12, v13, v14, v15, m[0], m[1], m[2], m[3], m[4],
                        m[5], m[6], m[7], 0x510E527Fu ^ t0);
  b2_half<LO, HI>(v0, v1, v2, v3, v5, v6, v7, v4, v10, v11, v8, v9, v15, v12, v13, v14, m[8], m[9], m[10], m[11], m[12], m[13],
                  m[14], m[15]);
  LMN_B2_ROUND(14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3)
  LMN_B2_ROUND(11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4)
  LMN_B2_ROUND(7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8)
  LMN_B2_ROUND(9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13)
  LMN_B2_ROUND(2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9)
  LMN_B2_ROUND(12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11)
  LMN_B2_ROUND(13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10)
  LMN_B2_ROUND(6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5)
  LMN_B2_ROUND(10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0)
  LMN_B2_LEAVE
  out[0] = (0x6A09E667u ^ 0x01010020u) ^ v0 ^ v8;
  out[1] = 0xBB67AE85u ^ v1 ^ v9;
  out[2] = 0x3C6EF372u ^ v2 ^ v10;
  out[3] = 0xA54FF53Au ^ v3 ^ v11;
  out[4] = 0x510E527Fu ^ v4 ^ v12;
  out[5] = 0x9B05688Cu ^ v5 ^ v13;
  out[6] = 0x1F83D9ABu ^ v6 ^ v14;
  out[7] = 0x5BE0CD19u ^ v7 ^ v15;
#else
  b2_init(out);
  b2_compress<LO, HI>(out, m, t0, 0xffffffffu);
#endif
}

// out <- F(h0, m, t0, final) as b2_compress_fresh for a block whose message words m[NZ..15] are ZERO by construction
// (a Merkle leaf of NZ < 16 columns; a FRI layer's leaf = 4 coordinate words): m[k >= NZ] is never read.
// Which message word a quarter round adds is fixed by the sigma schedule, so for every one of the 160 message additions
// it is known at compile time whether it adds zero: those become a two-operand v_add_u32 - an instruction of the class
// that either issue port takes, issued in the wave's low-priority phase - instead of a v_add3_u32 on the first port, and
// the zero words hold no registers.  4-column leaves (the composition tree and every FRI layer: 12.6 M of the 36 M
// compressions of a 2^20-row proof): 120 of 160; 12 columns: 40; 15 columns: 10.  For NZ <= 4 the first half round's
// columns 2 and 3 are compile-time constants (folded by the compiler: that half round is plain C++).
#ifdef LMN_B2_HAVE_FRESH
// One half of a half round (two of the four steps of four independent quarter rounds) as ONE asm statement whose text
// depends on which of its four message words are zero: the message additions of the non-zero words first (v_add3_u32,
// first issue port, priority HI), then - at priority LO - the two-operand additions of the zero words and the rest of
// the phase structure of b2_half.  Operands: %0-3 a, %4-7 b, %8-11 c, %12-15 d, %16-19 message words (a zero word's
// operand is a dummy and not referenced), %20 LO, %21 HI.  R1 / R2: the two rotation amounts (16, 12 or 8, 7).
#define LMN_B2_P0_0(a, b, x) "v_add3_u32 " a ", " a ", " b ", " x "\n"
#define LMN_B2_P0_1(a, b, x) ""
#define LMN_B2_ANY_0(a, b) ""
#define LMN_B2_ANY_1(a, b) "v_add_u32 " a ", " a ", " b "\n"
#define LMN_B2_QSTEP_TEXT(Z0, Z1, Z2, Z3, R1, R2)                                                                      \
  LMN_B2_P0_##Z0("%0", "%4", "%16") LMN_B2_P0_##Z1("%1", "%5", "%17") LMN_B2_P0_##Z2("%2", "%6", "%18")                \
  LMN_B2_P0_##Z3("%3", "%7", "%19")                                                                                    \
  "s_setprio %20\n"                                                                                                    \
  LMN_B2_ANY_##Z0("%0", "%4") LMN_B2_ANY_##Z1("%1", "%5") LMN_B2_ANY_##Z2("%2", "%6") LMN_B2_ANY_##Z3("%3", "%7")      \
  "v_xor_b32 %12, %12, %0\n v_xor_b32 %13, %13, %1\n v_xor_b32 %14, %14, %2\n v_xor_b32 %15, %15, %3\n"                \
  "s_setprio %21\n"                                                                                                    \
  "v_alignbit_b32 %12, %12, %12, " R1 "\n v_alignbit_b32 %13, %13, %13, " R1 "\n v_alignbit_b32 %14, %14, %14, " R1    \
  "\n v_alignbit_b32 %15, %15, %15, " R1 "\n"                                                                          \
  "s_setprio %20\n"                                                                                                    \
  "v_add_u32 %8, %8, %12\n v_add_u32 %9, %9, %13\n v_add_u32 %10, %10, %14\n v_add_u32 %11, %11, %15\n"                \
  "v_xor_b32 %4, %4, %8\n v_xor_b32 %5, %5, %9\n v_xor_b32 %6, %6, %10\n v_xor_b32 %7, %7, %11\n"                      \
  "s_setprio %21\n"                                                                                                    \
  "v_alignbit_b32 %4, %4, %4, " R2 "\n v_alignbit_b32 %5, %5, %5, " R2 "\n v_alignbit_b32 %6, %6, %6, " R2             \
  "\n v_alignbit_b32 %7, %7, %7, " R2 "\n"
#define LMN_B2_QSTEP_CASE(Z0, Z1, Z2, Z3)                                                                               \
  if constexpr (ZP == (Z0 | Z1 << 1 | Z2 << 2 | Z3 << 3)) {                                                             \
    if constexpr (FIRST)                                                                                                \
      asm volatile(LMN_B2_QSTEP_TEXT(Z0, Z1, Z2, Z3, "16", "12") LMN_B2_QSTEP_OPERANDS);                                \
    else                                                                                                                \
      asm volatile(LMN_B2_QSTEP_TEXT(Z0, Z1, Z2, Z3, "8", "7") LMN_B2_QSTEP_OPERANDS);                                  \
  }
#define LMN_B2_QSTEP_OPERANDS                                                                                           \
  : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(c0), "+v"(c1), "+v"(c2),       \
    "+v"(c3), "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3)                                                                    \
  : "v"(w0), "v"(w1), "v"(w2), "v"(w3), "n"(LO), "n"(HI)
// ZP: bit i = message word w_i is zero (then w_i is not read: callers pass any register)
template <int LO, int HI, unsigned ZP, bool FIRST>
__device__ __forceinline__ void b2_qstep_z(uint32_t& a0, uint32_t& a1, uint32_t& a2, uint32_t& a3, uint32_t& b0, uint32_t& b1,
                                           uint32_t& b2, uint32_t& b3, uint32_t& c0, uint32_t& c1, uint32_t& c2, uint32_t& c3,
                                           uint32_t& d0, uint32_t& d1, uint32_t& d2, uint32_t& d3, uint32_t w0, uint32_t w1,
                                           uint32_t w2, uint32_t w3) {
  LMN_B2_QSTEP_CASE(0, 0, 0, 0) LMN_B2_QSTEP_CASE(1, 0, 0, 0) LMN_B2_QSTEP_CASE(0, 1, 0, 0) LMN_B2_QSTEP_CASE(1, 1, 0, 0)
  LMN_B2_QSTEP_CASE(0, 0, 1, 0) LMN_B2_QSTEP_CASE(1, 0, 1, 0) LMN_B2_QSTEP_CASE(0, 1, 1, 0) LMN_B2_QSTEP_CASE(1, 1, 1, 0)
  LMN_B2_QSTEP_CASE(0, 0, 0, 1) LMN_B2_QSTEP_CASE(1, 0, 0, 1) LMN_B2_QSTEP_CASE(0, 1, 0, 1) LMN_B2_QSTEP_CASE(1, 1, 0, 1)
  LMN_B2_QSTEP_CASE(0, 0, 1, 1) LMN_B2_QSTEP_CASE(1, 0, 1, 1) LMN_B2_QSTEP_CASE(0, 1, 1, 1) LMN_B2_QSTEP_CASE(1, 1, 1, 1)
}
// ZM: bit 2i = x_i is zero, bit 2i+1 = y_i is zero.  Entered and left at priority HI like b2_half.
template <int LO, int HI, unsigned ZM>
__device__ __forceinline__ void b2_half_z(uint32_t& a0, uint32_t& a1, uint32_t& a2, uint32_t& a3, uint32_t& b0, uint32_t& b1,
                                          uint32_t& b2, uint32_t& b3, uint32_t& c0, uint32_t& c1, uint32_t& c2, uint32_t& c3,
                                          uint32_t& d0, uint32_t& d1, uint32_t& d2, uint32_t& d3, uint32_t x0, uint32_t y0,
                                          uint32_t x1, uint32_t y1, uint32_t x2, uint32_t y2, uint32_t x3, uint32_t y3) {
  if constexpr (ZM == 0u) {
    b2_half<LO, HI>(a0, a1, a2, a3, b0, b1, b2, b3, c0, c1, c2, c3, d0, d1, d2, d3, x0, y0, x1, y1, x2, y2, x3, y3);
  } else {
    constexpr unsigned ZX = (ZM & 1u) | (ZM >> 1 & 2u) | (ZM >> 2 & 4u) | (ZM >> 3 & 8u);
    constexpr unsigned ZY = (ZM >> 1 & 1u) | (ZM >> 2 & 2u) | (ZM >> 3 & 4u) | (ZM >> 4 & 8u);
    // (a zero word's operand: the lane's own a_i - a register that is live anyway)
    b2_qstep_z<LO, HI, ZX, true>(a0, a1, a2, a3, b0, b1, b2, b3, c0, c1, c2, c3, d0, d1, d2, d3, (ZX & 1u) ? a0 : x0,
                                 (ZX & 2u) ? a1 : x1, (ZX & 4u) ? a2 : x2, (ZX & 8u) ? a3 : x3);
    b2_qstep_z<LO, HI, ZY, false>(a0, a1, a2, a3, b0, b1, b2, b3, c0, c1, c2, c3, d0, d1, d2, d3, (ZY & 1u) ? a0 : y0,
                                  (ZY & 2u) ? a1 : y1, (ZY & 4u) ? a2 : y2, (ZY & 8u) ? a3 : y3);
  }
}
#define LMN_B2_ZW(s) ((s) >= NZ ? 0u : m[(s) < NZ ? (s) : 0])
#define LMN_B2_ZM(s0, s1, s2, s3, s4, s5, s6, s7)                                                                      \
  (((s0) >= NZ ? 1u : 0u) | ((s1) >= NZ ? 2u : 0u) | ((s2) >= NZ ? 4u : 0u) | ((s3) >= NZ ? 8u : 0u) |                 \
   ((s4) >= NZ ? 16u : 0u) | ((s5) >= NZ ? 32u : 0u) | ((s6) >= NZ ? 64u : 0u) | ((s7) >= NZ ? 128u : 0u))
#define LMN_B2_HALF_Z(A0, A1, A2, A3, B0, B1, B2, B3, C0, C1, C2, C3, D0, D1, D2, D3, s0, s1, s2, s3, s4, s5, s6, s7)   \
  b2_half_z<LO, HI, LMN_B2_ZM(s0, s1, s2, s3, s4, s5, s6, s7)>(A0, A1, A2, A3, B0, B1, B2, B3, C0, C1, C2, C3, D0, D1, \
                                                               D2, D3, LMN_B2_ZW(s0), LMN_B2_ZW(s1), LMN_B2_ZW(s2),   \
                                                               LMN_B2_ZW(s3), LMN_B2_ZW(s4), LMN_B2_ZW(s5),           \
                                                               LMN_B2_ZW(s6), LMN_B2_ZW(s7));
#define LMN_B2_ROUND_Z(s0, s1, s2, s3, s4, s5, s6, s7, s8, s9, s10, s11, s12, s13, s14, s15)                            \
  LMN_B2_HALF_Z(v0, v1, v2, v3, v4, v5, v6, v7, v8, v9, v10, v11, v12, v13, v14, v15, s0, s1, s2, s3, s4, s5, s6, s7)   \
  LMN_B2_HALF_Z(v0, v1, v2, v3, v5, v6, v7, v4, v10, v11, v8, v9, v15, v12, v13, v14, s8, s9, s10, s11, s12, s13, s14, s15)
#endif

template <int NZ, int LO = 0, int HI = LMN_B2_PRIO_HI>
LMN_HD void b2_compress_fresh_nz(uint32_t out[8], const uint32_t m[16], uint32_t t0) {
  static_assert(NZ >= 1 && NZ <= 16, "number of leading message words that may be non-zero");
#ifdef LMN_B2_HAVE_FRESH
  if constexpr (NZ == 16) {
    b2_compress_fresh<LO, HI>(out, m, t0);
  } else {
    uint32_t v0, v1, v2, v3, v4, v5, v6, v7, v8, v9, v10, v11, v12, v13, v14, v15;
    if constexpr (NZ >= 8) {
      b2_half_first<LO, HI>(v0, v1, v2, v3, v4, v5, v6, v7, v8, v9, v10, v11, v12, v13, v14, v15, m[0], m[1], m[2], m[3],
                            m[4], m[5], m[6], m[7], 0x510E527Fu ^ t0);
    } else {
      // column step of round 0 from the constant initial state: the columns whose two words are zero fold to constants
      v0 = 0x6A09E667u ^ 0x01010020u; v1 = 0xBB67AE85u; v2 = 0x3C6EF372u; v3 = 0xA54FF53Au;
      v4 = 0x510E527Fu; v5 = 0x9B05688Cu; v6 = 0x1F83D9ABu; v7 = 0x5BE0CD19u;
      v8 = 0x6A09E667u; v9 = 0xBB67AE85u; v10 = 0x3C6EF372u; v11 = 0xA54FF53Au;
      v12 = 0x510E527Fu ^ t0; v13 = 0x9B05688Cu; v14 = 0x1F83D9ABu ^ 0xffffffffu; v15 = 0x5BE0CD19u;
      LMN_B2_G(v0, v4, v8, v12, LMN_B2_ZW(0), LMN_B2_ZW(1))
      LMN_B2_G(v1, v5, v9, v13, LMN_B2_ZW(2), LMN_B2_ZW(3))
      LMN_B2_G(v2, v6, v10, v14, LMN_B2_ZW(4), LMN_B2_ZW(5))
      LMN_B2_G(v3, v7, v11, v15, LMN_B2_ZW(6), LMN_B2_ZW(7))
      LMN_B2_ENTER
    }
    LMN_B2_HALF_Z(v0, v1, v2, v3, v5, v6, v7, v4, v10, v11, v8, v9, v15, v12, v13, v14, 8, 9, 10, 11, 12, 13, 14, 15)
    LMN_B2_ROUND_Z(14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3)
    LMN_B2_ROUND_Z(11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4)
    LMN_B2_ROUND_Z(7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8)
    LMN_B2_ROUND_Z(9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13)
    LMN_B2_ROUND_Z(2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9)
    LMN_B2_ROUND_Z(12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11)
    LMN_B2_ROUND_Z(13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10)
    LMN_B2_ROUND_Z(6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5)
    LMN_B2_ROUND_Z(10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0)
    LMN_B2_LEAVE
    out[0] = (0x6A09E667u ^ 0x01010020u) ^ v0 ^ v8;
    out[1] = 0xBB67AE85u ^ v1 ^ v9;
    out[2] = 0x3C6EF372u ^ v2 ^ v10;
    out[3] = 0xA54FF53Au ^ v3 ^ v11;
    out[4] = 0x510E527Fu ^ v4 ^ v12;
    out[5] = 0x9B05688Cu ^ v5 ^ v13;
    out[6] = 0x1F83D9ABu ^ v6 ^ v14;
    out[7] = 0x5BE0CD19u ^ v7 ^ v15;
  }
#else
  b2_compress_fresh<LO, HI>(out, m, t0);   // (host / emulation: the caller's zero words are read)
#endif
}

// Two independent compressions interleaved statement by statement: 8 independent dependency chains per
// half-round instead of 4, which keeps the VALU issuing when only ~2 waves share a SIMD.
#define LMN_B2_G2(a, b, c, d, x, y, A, B, C, D, X, Y) \
  a = a + b + (x);                                    \
  A = A + B + (X);                                    \
  d = b2_rotr(d ^ a, 16);                             \
  D = b2_rotr(D ^ A, 16);                             \
  c = c + d;                                          \
  C = C + D;                                          \
  b = b2_rotr(b ^ c, 12);                             \
  B = b2_rotr(B ^ C, 12);                             \
  a = a + b + (y);                                    \
  A = A + B + (Y);                                    \
  d = b2_rotr(d ^ a, 8);                              \
  D = b2_rotr(D ^ A, 8);                              \
  c = c + d;                                          \
  C = C + D;                                          \
  b = b2_rotr(b ^ c, 7);                              \
  B = b2_rotr(B ^ C, 7);

#define LMN_B2_ROUND2(s0, s1, s2, s3, s4, s5, s6, s7, s8, s9, s10, s11, s12, s13, s14, s15)      \
  LMN_B2_G2(v0, v4, v8, v12, m[s0], m[s1], w0, w4, w8, w12, n[s0], n[s1])                        \
  LMN_B2_G2(v1, v5, v9, v13, m[s2], m[s3], w1, w5, w9, w13, n[s2], n[s3])                        \
  LMN_B2_G2(v2, v6, v10, v14, m[s4], m[s5], w2, w6, w10, w14, n[s4], n[s5])                      \
  LMN_B2_G2(v3, v7, v11, v15, m[s6], m[s7], w3, w7, w11, w15, n[s6], n[s7])                      \
  LMN_B2_G2(v0, v5, v10, v15, m[s8], m[s9], w0, w5, w10, w15, n[s8], n[s9])                      \
  LMN_B2_G2(v1, v6, v11, v12, m[s10], m[s11], w1, w6, w11, w12, n[s10], n[s11])                  \
  LMN_B2_G2(v2, v7, v8, v13, m[s12], m[s13], w2, w7, w8, w13, n[s12], n[s13])                    \
  LMN_B2_G2(v3, v4, v9, v14, m[s14], m[s15], w3, w4, w9, w14, n[s14], n[s15])

// h <- F(h, m, t, f0) and g <- F(g, n, t, f0) (same counter / finalisation flag for both)
LMN_HD void b2_compress2(uint32_t h[8], const uint32_t m[16], uint32_t g[8], const uint32_t n[16], uint32_t t0,
                         uint32_t f0) {
  uint32_t v0 = h[0], v1 = h[1], v2 = h[2], v3 = h[3], v4 = h[4], v5 = h[5], v6 = h[6], v7 = h[7];
  uint32_t w0 = g[0], w1 = g[1], w2 = g[2], w3 = g[3], w4 = g[4], w5 = g[5], w6 = g[6], w7 = g[7];
  uint32_t v8 = 0x6A09E667u, v9 = 0xBB67AE85u, v10 = 0x3C6EF372u, v11 = 0xA54FF53Au;
  uint32_t w8 = 0x6A09E667u, w9 = 0xBB67AE85u, w10 = 0x3C6EF372u, w11 = 0xA54FF53Au;
  uint32_t v12 = 0x510E527Fu ^ t0, v13 = 0x9B05688Cu, v14 = 0x1F83D9ABu ^ f0, v15 = 0x5BE0CD19u;
  uint32_t w12 = 0x510E527Fu ^ t0, w13 = 0x9B05688Cu, w14 = 0x1F83D9ABu ^ f0, w15 = 0x5BE0CD19u;
  LMN_B2_ROUND2(0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15)
  LMN_B2_ROUND2(14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3)
  LMN_B2_ROUND2(11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4)
  LMN_B2_ROUND2(7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8)
  LMN_B2_ROUND2(9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13)
  LMN_B2_ROUND2(2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9)
  LMN_B2_ROUND2(12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11)
  LMN_B2_ROUND2(13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10)
  LMN_B2_ROUND2(6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5)
  LMN_B2_ROUND2(10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0)
  h[0] ^= v0 ^ v8;
  h[1] ^= v1 ^ v9;
  h[2] ^= v2 ^ v10;
  h[3] ^= v3 ^ v11;
  h[4] ^= v4 ^ v12;
  h[5] ^= v5 ^ v13;
  h[6] ^= v6 ^ v14;
  h[7] ^= v7 ^ v15;
  g[0] ^= w0 ^ w8;
  g[1] ^= w1 ^ w9;
  g[2] ^= w2 ^ w10;
  g[3] ^= w3 ^ w11;
  g[4] ^= w4 ^ w12;
  g[5] ^= w5 ^ w13;
  g[6] ^= w6 ^ w14;
  g[7] ^= w7 ^ w15;
}

// sigma schedule packed 4 bits per entry (entry i of round r at bits 4i..4i+3), for per-lane lookups
#define LMN_B2_SIGMA_PACK(s0, s1, s2, s3, s4, s5, s6, s7, s8, s9, s10, s11, s12, s13, s14, s15)                    \
  ((uint64_t)(s0) | (uint64_t)(s1) << 4 | (uint64_t)(s2) << 8 | (uint64_t)(s3) << 12 | (uint64_t)(s4) << 16 |      \
   (uint64_t)(s5) << 20 | (uint64_t)(s6) << 24 | (uint64_t)(s7) << 28 | (uint64_t)(s8) << 32 |                     \
   (uint64_t)(s9) << 36 | (uint64_t)(s10) << 40 | (uint64_t)(s11) << 44 | (uint64_t)(s12) << 48 |                  \
   (uint64_t)(s13) << 52 | (uint64_t)(s14) << 56 | (uint64_t)(s15) << 60)

LMN_HD void b2_init(uint32_t h[8]) {
  h[0] = 0x6A09E667u ^ 0x01010020u;
  h[1] = 0xBB67AE85u;
  h[2] = 0x3C6EF372u;
  h[3] = 0xA54FF53Au;
  h[4] = 0x510E527Fu;
  h[5] = 0x9B05688Cu;
  h[6] = 0x1F83D9ABu;
  h[7] = 0x5BE0CD19u;
}

// Hash a message given as `nwords` little-endian 32-bit words (host helper).
inline Hash32 b2_hash_words(const uint32_t* words, size_t nwords) {
  uint32_t h[8];
  b2_init(h);
  size_t nblocks = nwords == 0 ? 1 : (nwords + 15) / 16;
  for (size_t b = 0; b < nblocks; ++b) {
    uint32_t m[16];
    for (int i = 0; i < 16; ++i) {
      size_t k = 16 * b + i;
      m[i] = k < nwords ? words[k] : 0u;
    }
    bool last = b + 1 == nblocks;
    uint32_t t = last ? (uint32_t)(4 * nwords) : (uint32_t)(64 * (b + 1));
    b2_compress(h, m, t, last ? 0xffffffffu : 0u);
  }
  Hash32 r;
  for (int i = 0; i < 8; ++i) r.w[i] = h[i];
  return r;
}

}  // namespace lmn

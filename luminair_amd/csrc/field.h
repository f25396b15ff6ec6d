// M31 / CM31 / QM31 arithmetic shared by host orchestration and gfx950 kernels.
// Replaces stwo `core/fields/{m31,cm31,qm31}.rs` (un-vendored dependency of
// /root/reference/crates/prover): P = 2^31-1, CM31 = M31[i]/(i^2+1),
// QM31 = CM31[u]/(u^2-(2+i)); (a,b,c,d) = (a+bi)+(c+di)u.
#pragma once
#include "platform.h"

namespace lmn {

constexpr uint32_t P31 = 0x7fffffffu;

LMN_HD uint32_t m_add(uint32_t a, uint32_t b) {
  uint32_t s = a + b;
  return s >= P31 ? s - P31 : s;
}
LMN_HD uint32_t m_sub(uint32_t a, uint32_t b) {
  // a >= b: d = a-b < P <= d+P; a < b: d wrapped (> 2^32-P), d+P wraps to a-b+P < P.  sub, add, min.
  uint32_t d = a - b, e = d + P31;
  return d < e ? d : e;
}
LMN_HD uint32_t m_neg(uint32_t a) { return a ? P31 - a : 0u; }
LMN_HD uint32_t m_mul(uint32_t a, uint32_t b) {
  uint64_t p = (uint64_t)a * (uint64_t)b;
  uint32_t s = (uint32_t)(p & P31) + (uint32_t)(p >> 31);
  return s >= P31 ? s - P31 : s;
}
LMN_HD uint32_t m_sqr(uint32_t a) { return m_mul(a, a); }
LMN_HD uint32_t m_dbl(uint32_t a) { return m_add(a, a); }
// reduce any 64-bit value to canonical M31
LMN_HD uint32_t m_red64(uint64_t p) {
  uint64_t s = (p & P31) + (p >> 31);       // < 2^32
  uint32_t t = (uint32_t)(s & P31) + (uint32_t)(s >> 31);
  return t >= P31 ? t - P31 : t;
}
// a^(2^n)
LMN_HD uint32_t m_sqn(uint32_t a, int n) {
  for (int i = 0; i < n; ++i) a = m_sqr(a);
  return a;
}
// a^(P-2) by the addition chain 2^31-3 = 2*(2^30-1) - 1 ... (37 multiplications)
LMN_HD uint32_t m_inv(uint32_t a) {
  uint32_t t0 = m_mul(m_sqn(a, 2), a);      // a^5
  uint32_t t1 = m_mul(m_sqn(t0, 1), t0);    // a^15
  uint32_t t2 = m_mul(m_sqn(t1, 3), t0);    // a^125
  uint32_t t3 = m_mul(m_sqn(t2, 1), t0);    // a^255
  uint32_t t4 = m_mul(m_sqn(t3, 8), t3);    // a^65535
  uint32_t t5 = m_mul(m_sqn(t4, 8), t3);    // a^16777215
  uint32_t t6 = m_mul(m_sqn(t5, 7), t2);    // a^2147483645 = a^(P-2)
  return t6;
}

struct CM31 {
  uint32_t a, b;
};
LMN_HD CM31 c_add(CM31 x, CM31 y) { return {m_add(x.a, y.a), m_add(x.b, y.b)}; }
LMN_HD CM31 c_sub(CM31 x, CM31 y) { return {m_sub(x.a, y.a), m_sub(x.b, y.b)}; }
LMN_HD CM31 c_mul(CM31 x, CM31 y) {
  return {m_sub(m_mul(x.a, y.a), m_mul(x.b, y.b)), m_add(m_mul(x.a, y.b), m_mul(x.b, y.a))};
}
LMN_HD CM31 c_mul_m(CM31 x, uint32_t m) { return {m_mul(x.a, m), m_mul(x.b, m)}; }
LMN_HD uint32_t c_norm(CM31 x) { return m_add(m_sqr(x.a), m_sqr(x.b)); }
LMN_HD CM31 c_inv(CM31 x) {
  uint32_t n = m_inv(c_norm(x));
  return {m_mul(x.a, n), m_mul(m_neg(x.b), n)};
}
// (2+i)*x
LMN_HD CM31 c_mul_r(CM31 x) { return {m_sub(m_dbl(x.a), x.b), m_add(x.a, m_dbl(x.b))}; }

struct QM31 {
  uint32_t a, b, c, d;
};
LMN_HD QM31 q_zero() { return {0, 0, 0, 0}; }
LMN_HD QM31 q_one() { return {1, 0, 0, 0}; }
LMN_HD QM31 q_from_m(uint32_t m) { return {m, 0, 0, 0}; }
LMN_HD bool q_eq(QM31 x, QM31 y) { return x.a == y.a && x.b == y.b && x.c == y.c && x.d == y.d; }
LMN_HD bool q_is_zero(QM31 x) { return (x.a | x.b | x.c | x.d) == 0; }
LMN_HD QM31 q_add(QM31 x, QM31 y) { return {m_add(x.a, y.a), m_add(x.b, y.b), m_add(x.c, y.c), m_add(x.d, y.d)}; }
LMN_HD QM31 q_sub(QM31 x, QM31 y) { return {m_sub(x.a, y.a), m_sub(x.b, y.b), m_sub(x.c, y.c), m_sub(x.d, y.d)}; }
LMN_HD QM31 q_neg(QM31 x) { return {m_neg(x.a), m_neg(x.b), m_neg(x.c), m_neg(x.d)}; }
LMN_HD QM31 q_add_m(QM31 x, uint32_t m) { return {m_add(x.a, m), x.b, x.c, x.d}; }
LMN_HD QM31 q_sub_m(QM31 x, uint32_t m) { return {m_sub(x.a, m), x.b, x.c, x.d}; }
LMN_HD QM31 q_mul_m(QM31 x, uint32_t m) { return {m_mul(x.a, m), m_mul(x.b, m), m_mul(x.c, m), m_mul(x.d, m)}; }
// (A + B u)(C + D u) with u^2 = 2 + i:  lo = A C + (2+i) B D,  hi = A D + B C.  Written out per
// coordinate, every output is a sum of exactly four products once the signs and the factor (2+i) are
// folded into the left operands (P - v for negation, canonical doubles, three M31 sums):
//   lo.re = xa ya - xb yb + (2xc - xd) yc - (2xd + xc) yd
//   lo.im = xa yb + xb ya + (xc + 2xd) yc + (2xc - xd) yd
//   hi.re = xa yc - xb yd + xc ya - xd yb          hi.im = xa yd + xb yc + xc yb + xd ya
// Four products of 31-bit values fit one 64-bit accumulator (4 (2^31-1)^2 < 2^64), so each coordinate is
// four v_mad_u64_u32 and ONE reduction instead of four multiply-reduce and three add-reduce steps.
LMN_HD QM31 q_mul(QM31 x, QM31 y) {
  const uint32_t nb = P31 - x.b, nd = P31 - x.d;               // -xb, -xd (P itself stands for 0)
  const uint32_t c2 = m_dbl(x.c), d2 = m_dbl(x.d);
  const uint32_t e1 = m_sub(c2, x.d);                           // 2xc - xd
  const uint32_t e2 = P31 - m_add(d2, x.c);                     // -(2xd + xc)
  const uint32_t f1 = m_add(x.c, d2);                           // xc + 2xd
  const uint64_t lre = (uint64_t)x.a * y.a + (uint64_t)nb * y.b + (uint64_t)e1 * y.c + (uint64_t)e2 * y.d;
  const uint64_t lim = (uint64_t)x.a * y.b + (uint64_t)x.b * y.a + (uint64_t)f1 * y.c + (uint64_t)e1 * y.d;
  const uint64_t hre = (uint64_t)x.a * y.c + (uint64_t)nb * y.d + (uint64_t)x.c * y.a + (uint64_t)nd * y.b;
  const uint64_t him = (uint64_t)x.a * y.d + (uint64_t)x.b * y.c + (uint64_t)x.c * y.b + (uint64_t)x.d * y.a;
  return {m_red64(lre), m_red64(lim), m_red64(hre), m_red64(him)};
}
LMN_HD QM31 q_sqr(QM31 x) { return q_mul(x, x); }

// Lazy dot-product accumulator: sum_k c_k * f_k with c_k in QM31, f_k in M31 (canonical factors).
// Each product is < 2^62, so one 64-bit lane per coordinate takes three products on top of a folded
// value (< 2^34) before it must be folded again: one v_mad_u64_u32 per multiply-accumulate instead of
// a full multiply-reduce-add-reduce chain.  Results are identical after qacc_reduce.
struct QAcc {
  uint64_t a, b, c, d;
};
LMN_HD QAcc qacc_zero() { return {0ull, 0ull, 0ull, 0ull}; }
LMN_HD void qacc_mad(QAcc& s, const QM31& c, uint32_t f) {
  s.a += (uint64_t)c.a * f;
  s.b += (uint64_t)c.b * f;
  s.c += (uint64_t)c.c * f;
  s.d += (uint64_t)c.d * f;
}
LMN_HD uint64_t m_fold64(uint64_t s) { return (s & P31) + (s >> 31); }  // any s -> < 2^31 + 2^33, same residue
LMN_HD void qacc_fold(QAcc& s) {
  s.a = m_fold64(s.a);
  s.b = m_fold64(s.b);
  s.c = m_fold64(s.c);
  s.d = m_fold64(s.d);
}
LMN_HD QM31 qacc_reduce(const QAcc& s) { return {m_red64(s.a), m_red64(s.b), m_red64(s.c), m_red64(s.d)}; }
LMN_HD QM31 q_mul_c(QM31 x, CM31 c) {
  CM31 lo = c_mul({x.a, x.b}, c), hi = c_mul({x.c, x.d}, c);
  return {lo.a, lo.b, hi.a, hi.b};
}
LMN_HD QM31 q_conj(QM31 x) { return {x.a, x.b, m_neg(x.c), m_neg(x.d)}; }
LMN_HD QM31 q_inv(QM31 x) {
  CM31 A{x.a, x.b}, B{x.c, x.d};
  CM31 den = c_sub(c_mul(A, A), c_mul_r(c_mul(B, B)));
  CM31 di = c_inv(den);
  CM31 lo = c_mul(A, di), hi = c_mul(B, di);
  return {lo.a, lo.b, m_neg(hi.a), m_neg(hi.b)};
}
LMN_HD QM31 q_pow(QM31 x, uint64_t e) {
  QM31 r = q_one();
  while (e) {
    if (e & 1) r = q_mul(r, x);
    x = q_sqr(x);
    e >>= 1;
  }
  return r;
}
// SecureField::from_partial_evals: e0 + e1*i + e2*u + e3*iu
LMN_HD QM31 q_from_partial_evals(QM31 e0, QM31 e1, QM31 e2, QM31 e3) {
  QM31 r = e0;
  r = q_add(r, q_mul(e1, QM31{0, 1, 0, 0}));
  r = q_add(r, q_mul(e2, QM31{0, 0, 1, 0}));
  r = q_add(r, q_mul(e3, QM31{0, 0, 0, 1}));
  return r;
}

}  // namespace lmn

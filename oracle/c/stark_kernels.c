/* oracle/c/stark_kernels.c — plain-C restatement of the per-row work of the LuminAIR `prove` path.
 *
 * TEST INFRASTRUCTURE ONLY (checker + `cpu_baseline` leg of bench.py); never linked into the product.
 * It restates, as scalar loops (+ OpenMP over rows/columns), the algorithms stwo runs behind
 *   /root/reference/crates/prover/src/prover.rs:56-59,179,298 (interpolate / evaluate / Blake2s Merkle),
 *   crates/air/src/components/add/witness.rs:126-167        (LogupTraceGenerator),
 *   crates/air/src/components/{add,mul,recip}/component.rs, inputs/components.rs (constraint quotients),
 *   prover.rs:312 stwo::prover::prove                       (eval_at_point, FRI quotients, folds),
 * following SURVEY.md Appendix A.  The stwo sources are un-vendored (Cargo.toml:21-30), so the
 * published algorithm is restated; oracle/cbackend.py checks every function here against the numpy
 * oracle, which is pinned bit-for-bit on the reference's known-answer proof.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define P 0x7fffffffu
typedef uint32_t u32;
typedef uint64_t u64;

static inline u32 madd(u32 a, u32 b) { u32 s = a + b; return s >= P ? s - P : s; }
static inline u32 msub(u32 a, u32 b) { return a >= b ? a - b : a + P - b; }
static inline u32 mneg(u32 a) { return a ? P - a : 0; }
static inline u32 mmul(u32 a, u32 b) {
  u64 p = (u64)a * b;
  u32 s = (u32)(p & P) + (u32)(p >> 31);
  return s >= P ? s - P : s;
}
static u32 mpow(u32 a, u32 e) {
  u32 r = 1;
  while (e) {
    if (e & 1) r = mmul(r, a);
    a = mmul(a, a);
    e >>= 1;
  }
  return r;
}
static inline u32 minv(u32 a) { return mpow(a, P - 2); }

typedef struct { u32 a, b; } cm;
typedef struct { u32 a, b, c, d; } qm;
static inline cm cadd(cm x, cm y) { cm r = {madd(x.a, y.a), madd(x.b, y.b)}; return r; }
static inline cm csub(cm x, cm y) { cm r = {msub(x.a, y.a), msub(x.b, y.b)}; return r; }
static inline cm cmul(cm x, cm y) {
  cm r = {msub(mmul(x.a, y.a), mmul(x.b, y.b)), madd(mmul(x.a, y.b), mmul(x.b, y.a))};
  return r;
}
static inline cm cmulr(cm x) { cm r = {msub(madd(x.a, x.a), x.b), madd(x.a, madd(x.b, x.b))}; return r; } /* (2+i)x */
static inline cm cinv(cm x) {
  u32 n = minv(madd(mmul(x.a, x.a), mmul(x.b, x.b)));
  cm r = {mmul(x.a, n), mmul(mneg(x.b), n)};
  return r;
}
static inline qm qadd(qm x, qm y) { qm r = {madd(x.a, y.a), madd(x.b, y.b), madd(x.c, y.c), madd(x.d, y.d)}; return r; }
static inline qm qsub(qm x, qm y) { qm r = {msub(x.a, y.a), msub(x.b, y.b), msub(x.c, y.c), msub(x.d, y.d)}; return r; }
static inline qm qmulm(qm x, u32 m) { qm r = {mmul(x.a, m), mmul(x.b, m), mmul(x.c, m), mmul(x.d, m)}; return r; }
static inline qm qmul(qm x, qm y) {
  cm A = {x.a, x.b}, B = {x.c, x.d}, C = {y.a, y.b}, D = {y.c, y.d};
  cm lo = cadd(cmul(A, C), cmulr(cmul(B, D)));
  cm hi = cadd(cmul(A, D), cmul(B, C));
  qm r = {lo.a, lo.b, hi.a, hi.b};
  return r;
}
static inline qm qmulc(qm x, cm c) {
  cm A = {x.a, x.b}, B = {x.c, x.d};
  cm lo = cmul(A, c), hi = cmul(B, c);
  qm r = {lo.a, lo.b, hi.a, hi.b};
  return r;
}
static inline qm qinv(qm x) {
  cm A = {x.a, x.b}, B = {x.c, x.d};
  cm den = csub(cmul(A, A), cmulr(cmul(B, B)));
  cm di = cinv(den);
  cm lo = cmul(A, di), hi = cmul(B, di);
  qm r = {lo.a, lo.b, mneg(hi.a), mneg(hi.b)};
  return r;
}
static inline qm qfromm(u32 m) { qm r = {m, 0, 0, 0}; return r; }
static const qm QZERO = {0, 0, 0, 0};

/* ---------------------------------------------------------------------------------------------
 * Circle FFT, one layer at a time (Appendix A.2).  data: ncols columns of 2^log_n words.
 * tw[i] = twiddles of layer i (2^(log_n-1-i) words): forward twiddles, or their inverses when
 * inverse != 0 (the caller passes the matching table).  Inverse also scales by 2^-log_n.
 * ------------------------------------------------------------------------------------------- */
/* The layers are grouped into two cache-resident passes (the same split the GPU kernels use): the low layers
 * 0..CH-1 stay inside one contiguous chunk of 2^CH words; the high layers CH..log_n-1 pair rows that are 2^CH words
 * apart, so a tile of FFT_G consecutive words x all 2^(log_n-CH) rows is gathered into a local buffer, transformed and
 * scattered back.  Every butterfly is the same arithmetic as the one-layer-at-a-time form (tests/test_oracle_c.py
 * compares it with the numpy restatement); only the order of independent butterflies changes.
 * HOT marks the functions compiled for AVX-512 / AVX2 / baseline x86-64 and dispatched by cpuid at load time: the
 * library is built in one container and travels to other hosts. */
#ifdef ORC_SCALAR /* liboracle_kernels_scalar.so: one scalar instance of every function, vectoriser off (Makefile) */
#define HOT
#else
#define HOT __attribute__((target_clones("arch=skylake-avx512", "avx2", "default")))
#endif
#define FFT_G 16
#define PAR_MIN (1L << 15) /* below this many words an OpenMP team costs more than it saves */
#include <omp.h>
/* A loop goes parallel only when every thread of the (full) team gets at least 2^12 work units: waking a 256-thread
 * team for a 2^12-point loop costs more than the loop itself.  (Always the full team or none: libgomp tears threads
 * down and re-creates them when consecutive regions ask for different team sizes.) */
static inline int par_ok(long work) {
  const long per_team = (long)omp_get_max_threads() << 12;
  return work >= (per_team > (1L << 18) ? per_team : (1L << 18));
}

/* ---- M31 over LW lanes: plain loops over fixed-length arrays, written branch-free so that the HOT clones compile
 * each line to vector instructions (the same shape stwo's SimdBackend gives its 16-lane PackedM31). */
#define LW 16
static inline u32 red1(u32 s) { u32 t = s - P; return t < s ? t : s; }           /* s in [0, 2P) -> [0, P): min(s, s - P) unsigned */
static inline u32 vmadd(u32 a, u32 b) { return red1(a + b); }
static inline u32 vmsub(u32 a, u32 b) { u32 d = a - b; u32 e = d + P; return e < d ? e : d; } /* a - b wraps iff a < b: then d + P wraps back below d */
static inline u32 vmmul(u32 a, u32 b) {
  u64 p = (u64)a * b;
  return red1((u32)(p & P) + (u32)(p >> 31));
}

/* out[i] = 1 / v[i] for i < n, n a multiple of LW: LW independent Montgomery chains side by side (element i belongs to
 * chain i % LW), one inversion per chain by exponentiation, every step a vector line.  tmp: n words. */
HOT static void batch_inverse_lanes(const u32* v, long n, u32* out, u32* tmp) {
  u32 acc[LW], inv[LW];
  for (int l = 0; l < LW; ++l) acc[l] = 1;
  for (long i = 0; i < n; i += LW)
    for (int l = 0; l < LW; ++l) { tmp[i + l] = acc[l]; acc[l] = vmmul(acc[l], v[i + l]); }
  for (int l = 0; l < LW; ++l) inv[l] = 1;
  { /* inv = acc^(P-2): P - 2 = 0x7ffffffd */
    u32 b[LW];
    for (int l = 0; l < LW; ++l) b[l] = acc[l];
    for (u32 e = P - 2; e; e >>= 1) {
      if (e & 1) for (int l = 0; l < LW; ++l) inv[l] = vmmul(inv[l], b[l]);
      for (int l = 0; l < LW; ++l) b[l] = vmmul(b[l], b[l]);
    }
  }
  for (long i = n - LW; i >= 0; i -= LW)
    for (int l = 0; l < LW; ++l) { u32 x = v[i + l]; out[i + l] = vmmul(inv[l], tmp[i + l]); inv[l] = vmmul(inv[l], x); }
}

static inline void bfly_fwd(u32* lo, u32* hi, u32 w) {
  u32 x = mmul(*hi, w), a = *lo;
  *lo = madd(a, x);
  *hi = msub(a, x);
}
static inline void bfly_inv(u32* lo, u32* hi, u32 w) {
  u32 a = *lo, b = *hi;
  *lo = madd(a, b);
  *hi = mmul(msub(a, b), w);
}

/* layers [0, ch) of the chunk starting at word `base` of a column (chunk index = base >> ch) */
HOT static void fft_low_chunk(u32* col, long base, int ch, const u32* const* tw, int inverse) {
  u32* d = col + base;
  for (int s = 0; s < ch; ++s) {
    const int i = inverse ? s : ch - 1 - s;
    const long span = 1L << i, nh = 1L << (ch - 1 - i);
    const u32* t = tw[i] + ((base >> ch) << (ch - 1 - i));
    for (long h = 0; h < nh; ++h) {
      const u32 w = t[h];
      u32* lo = d + (h << (i + 1));
      u32* hi = lo + span;
      if (inverse)
        for (long l = 0; l < span; ++l) bfly_inv(lo + l, hi + l, w);
      else
        for (long l = 0; l < span; ++l) bfly_fwd(lo + l, hi + l, w);
    }
  }
}

/* layers [ch, log_n) on the tile of FFT_G words starting at low offset l0: buf[row][j], row = index >> ch */
HOT static void fft_high_tile(u32* col, long l0, int ch, int log_n, const u32* const* tw, int inverse, u32* buf) {
  const int T = log_n - ch;
  const long rows = 1L << T;
  for (long r = 0; r < rows; ++r)
    for (int j = 0; j < FFT_G; ++j) buf[r * FFT_G + j] = col[(r << ch) + l0 + j];
  for (int s = 0; s < T; ++s) {
    const int k = inverse ? s : T - 1 - s; /* row bit paired by layer ch + k */
    const long span = 1L << k, nh = 1L << (T - 1 - k);
    const u32* t = tw[ch + k];
    for (long h = 0; h < nh; ++h) {
      const u32 w = t[h];
      for (long l = 0; l < span; ++l) {
        u32* lo = buf + ((h << (k + 1)) + l) * FFT_G;
        u32* hi = lo + span * FFT_G;
        if (inverse)
          for (int j = 0; j < FFT_G; ++j) bfly_inv(lo + j, hi + j, w);
        else
          for (int j = 0; j < FFT_G; ++j) bfly_fwd(lo + j, hi + j, w);
      }
    }
  }
  for (long r = 0; r < rows; ++r)
    for (int j = 0; j < FFT_G; ++j) col[(r << ch) + l0 + j] = buf[r * FFT_G + j];
}

HOT static void scale_words(u32* d, long n, u32 sc) {
  for (long k = 0; k < n; ++k) d[k] = mmul(d[k], sc);
}

void orc_circle_fft(u32* data, long ncols, int log_n, const u32* const* tw, int inverse) {
  const long n = 1L << log_n;
  const u32 sc = minv(mpow(2, (u32)log_n));
  int ch = log_n <= 12 ? log_n : (log_n - 12 > 12 ? log_n - 12 : 12);
  if (ch < 4 && log_n >= 4) ch = 4; /* a high tile needs FFT_G consecutive words */
  if (log_n < 4) ch = log_n;
  const long nchunks = n >> ch, ntiles = ch >= 4 ? (1L << ch) / FFT_G : 0;
  const int T = log_n - ch;
#pragma omp parallel if (par_ok(ncols * n * (long)log_n / 8))
  {
    u32* buf = T > 0 ? (u32*)malloc(sizeof(u32) * FFT_G << T) : NULL;
    for (int pass = 0; pass < 2; ++pass) {
      const int low = inverse ? pass == 0 : pass == 1;
      if (low) {
#pragma omp for schedule(static) collapse(2)
        for (long c = 0; c < ncols; ++c)
          for (long k = 0; k < nchunks; ++k) fft_low_chunk(data + c * n, k << ch, ch, tw, inverse);
      } else if (T > 0) {
#pragma omp for schedule(static) collapse(2)
        for (long c = 0; c < ncols; ++c)
          for (long k = 0; k < ntiles; ++k) fft_high_tile(data + c * n, k * FFT_G, ch, log_n, tw, inverse, buf);
      }
    }
    if (inverse) {
      const long blk = 1L << 14, nblk = (ncols * n + blk - 1) / blk;
#pragma omp for schedule(static)
      for (long k = 0; k < nblk; ++k) {
        const long a = k * blk, e = a + blk < ncols * n ? a + blk : ncols * n;
        scale_words(data + a, e - a, sc);
      }
    }
    free(buf);
  }
}

/* ---------------------------------------------------------------------------------------------
 * Blake2s (RFC 7693) over rows of 32-bit words: out[i] = blake2s(words[i][0..w)) (Appendix A.4).
 * ------------------------------------------------------------------------------------------- */
static const u32 IV[8] = {0x6A09E667u, 0xBB67AE85u, 0x3C6EF372u, 0xA54FF53Au,
                          0x510E527Fu, 0x9B05688Cu, 0x1F83D9ABu, 0x5BE0CD19u};
static const unsigned char SIGMA[10][16] = {
    {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3},
    {11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4}, {7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8},
    {9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13}, {2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9},
    {12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11}, {13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10},
    {6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5}, {10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0}};
static inline u32 rotr(u32 x, int n) { return (x >> n) | (x << (32 - n)); }
static void compress(u32 h[8], const u32 m[16], u32 t, u32 f) {
  u32 v[16];
  for (int i = 0; i < 8; ++i) { v[i] = h[i]; v[i + 8] = IV[i]; }
  v[12] ^= t;
  v[14] ^= f;
#define G(a, b, c, d, x, y)                                           \
  v[a] += v[b] + (x); v[d] = rotr(v[d] ^ v[a], 16); v[c] += v[d]; v[b] = rotr(v[b] ^ v[c], 12); \
  v[a] += v[b] + (y); v[d] = rotr(v[d] ^ v[a], 8);  v[c] += v[d]; v[b] = rotr(v[b] ^ v[c], 7);
  for (int r = 0; r < 10; ++r) {
    const unsigned char* s = SIGMA[r];
    G(0, 4, 8, 12, m[s[0]], m[s[1]]) G(1, 5, 9, 13, m[s[2]], m[s[3]])
    G(2, 6, 10, 14, m[s[4]], m[s[5]]) G(3, 7, 11, 15, m[s[6]], m[s[7]])
    G(0, 5, 10, 15, m[s[8]], m[s[9]]) G(1, 6, 11, 12, m[s[10]], m[s[11]])
    G(2, 7, 8, 13, m[s[12]], m[s[13]]) G(3, 4, 9, 14, m[s[14]], m[s[15]])
  }
#undef G
  for (int i = 0; i < 8; ++i) h[i] ^= v[i] ^ v[i + 8];
}
/* LW independent hashes side by side (one per SIMD lane): the same compression function written over arrays so
 * that the compiler turns every line into one vector instruction (vprord etc. under AVX-512). */
HOT static void compress_lanes(u32 h[8][LW], u32 m[16][LW], u32 t, u32 f) {
  u32 v[16][LW];
  for (int i = 0; i < 8; ++i)
    for (int l = 0; l < LW; ++l) { v[i][l] = h[i][l]; v[i + 8][l] = IV[i]; }
  for (int l = 0; l < LW; ++l) { v[12][l] ^= t; v[14][l] ^= f; }
#define GL(a, b, c, d, x, y)                                                                  \
  for (int l = 0; l < LW; ++l) {                                                              \
    v[a][l] += v[b][l] + m[x][l]; v[d][l] = rotr(v[d][l] ^ v[a][l], 16);                      \
    v[c][l] += v[d][l];           v[b][l] = rotr(v[b][l] ^ v[c][l], 12);                      \
    v[a][l] += v[b][l] + m[y][l]; v[d][l] = rotr(v[d][l] ^ v[a][l], 8);                       \
    v[c][l] += v[d][l];           v[b][l] = rotr(v[b][l] ^ v[c][l], 7);                       \
  }
  for (int r = 0; r < 10; ++r) {
    const unsigned char* s = SIGMA[r];
    GL(0, 4, 8, 12, s[0], s[1]) GL(1, 5, 9, 13, s[2], s[3])
    GL(2, 6, 10, 14, s[4], s[5]) GL(3, 7, 11, 15, s[6], s[7])
    GL(0, 5, 10, 15, s[8], s[9]) GL(1, 6, 11, 12, s[10], s[11])
    GL(2, 7, 8, 13, s[12], s[13]) GL(3, 4, 9, 14, s[14], s[15])
  }
#undef GL
  for (int i = 0; i < 8; ++i)
    for (int l = 0; l < LW; ++l) h[i][l] ^= v[i][l] ^ v[i + 8][l];
}

/* nodes [i0, i0 + LW) of one Merkle layer (all lanes valid) */
HOT static void merkle_lanes(const u32* prev, const u32* const* cols, int ncols, long i0, u32* out) {
  const int npre = prev ? 16 : 0;
  const int w = npre + ncols;
  const int nblocks = w == 0 ? 1 : (w + 15) / 16;
  u32 h[8][LW], m[16][LW];
  for (int k = 0; k < 8; ++k)
    for (int l = 0; l < LW; ++l) h[k][l] = IV[k] ^ (k == 0 ? 0x01010020u : 0u);
  for (int b = 0; b < nblocks; ++b) {
    for (int k = 0; k < 16; ++k) {
      const int j = 16 * b + k;
      if (j < npre) {
        for (int l = 0; l < LW; ++l) m[k][l] = prev[16 * (i0 + l) + j];
      } else if (j < w) {
        const u32* c = cols[j - npre] + i0;
        for (int l = 0; l < LW; ++l) m[k][l] = c[l];
      } else {
        for (int l = 0; l < LW; ++l) m[k][l] = 0;
      }
    }
    const int last = b + 1 == nblocks;
    compress_lanes(h, m, last ? (u32)(4 * w) : (u32)(64 * (b + 1)), last ? 0xffffffffu : 0);
  }
  for (int l = 0; l < LW; ++l)
    for (int k = 0; k < 8; ++k) out[8 * (i0 + l) + k] = h[k][l];
}

void orc_blake2s_rows(const u32* words, long n, int w, u32* out) {
  const int nblocks = w == 0 ? 1 : (w + 15) / 16;
#pragma omp parallel for schedule(static) if (par_ok(n * (long)(w + 8) * 8))
  for (long i = 0; i < n; ++i) {
    u32 h[8];
    for (int k = 0; k < 8; ++k) h[k] = IV[k];
    h[0] ^= 0x01010020u;
    const u32* row = words + i * (long)w;
    for (int b = 0; b < nblocks; ++b) {
      u32 m[16];
      for (int k = 0; k < 16; ++k) { int j = 16 * b + k; m[k] = j < w ? row[j] : 0; }
      int last = b + 1 == nblocks;
      compress(h, m, last ? (u32)(4 * w) : (u32)(64 * (b + 1)), last ? 0xffffffffu : 0);
    }
    memcpy(out + 8 * i, h, 32);
  }
}

/* ---------------------------------------------------------------------------------------------
 * Logup (Appendix A.6): S_j[r] = S_{j-1}[r] + mult_j[r] / (val_j[r] + alpha*id_j[r] - z).
 * cols: k relations x (val, id, mult) pointers, each n words.  out: k secure columns as 4k base
 * columns of n words (running sums, NOT yet prefix-summed); claimed = sum over rows of S_{k-1}.
 * ------------------------------------------------------------------------------------------- */
/* One chunk of LCH rows: the k denominators of a row are QM31; 1/x = conj-style formula over the CM31 norm
 * N = A^2 - (2+i) B^2 of x = A + B u, whose own norm (an M31) is what gets inverted - all LCH x k of them together
 * (stwo batches the inversions of a logup column the same way), the rest lane-wise. */
#define LCH 1024
HOT static void logup_chunk(const u32* const* val, const u32* const* id, const u32* const* mult, int k, long r0, long n,
                            const u32* zs, const u32* alphas, const int* neg, u32* out, u64 loc[4], u32* scratch) {
  u32* xa = scratch;                 /* k x LCH each: denominator coordinates, CM31 norm, its M31 norm and inverse */
  u32* xb = xa + (long)k * LCH;
  u32* xc = xb + (long)k * LCH;
  u32* xd = xc + (long)k * LCH;
  u32* na = xd + (long)k * LCH;
  u32* nb = na + (long)k * LCH;
  u32* nn = nb + (long)k * LCH;
  u32* ni = nn + (long)k * LCH;
  u32* tmp = ni + (long)k * LCH;
  u32 *sa = tmp + (long)k * LCH, *sb = sa + LCH, *sc = sb + LCH, *sd = sc + LCH;
  for (int j = 0; j < k; ++j) {
    const u32 z0 = zs[4 * j], z1 = zs[4 * j + 1], z2 = zs[4 * j + 2], z3 = zs[4 * j + 3];
    const u32 a0 = alphas[4 * j], a1 = alphas[4 * j + 1], a2 = alphas[4 * j + 2], a3 = alphas[4 * j + 3];
    const u32* v = val[j] + r0;
    const u32* idp = id[j] ? id[j] + r0 : NULL;
    u32 *pa = xa + (long)j * LCH, *pb = xb + (long)j * LCH, *pc = xc + (long)j * LCH, *pd = xd + (long)j * LCH;
    u32 *qa = na + (long)j * LCH, *qb = nb + (long)j * LCH, *qn = nn + (long)j * LCH;
    for (int t = 0; t < LCH; ++t) {
      const u32 w = idp ? idp[t] : 0;
      const u32 a = vmsub(vmadd(v[t], vmmul(a0, w)), z0), b = vmsub(vmmul(a1, w), z1);
      const u32 c = vmsub(vmmul(a2, w), z2), d = vmsub(vmmul(a3, w), z3);
      pa[t] = a; pb[t] = b; pc[t] = c; pd[t] = d;
      /* A^2 = (a^2 - b^2, 2ab), B^2 = (c^2 - d^2, 2cd), (2+i)(x, y) = (2x - y, x + 2y) */
      const u32 asq = vmsub(vmmul(a, a), vmmul(b, b)), abb = vmmul(vmadd(a, a), b);
      const u32 csq = vmsub(vmmul(c, c), vmmul(d, d)), cdd = vmmul(vmadd(c, c), d);
      const u32 ra = vmsub(vmadd(csq, csq), cdd), rb = vmadd(csq, vmadd(cdd, cdd));
      const u32 Na = vmsub(asq, ra), Nb = vmsub(abb, rb);
      qa[t] = Na; qb[t] = Nb;
      qn[t] = vmadd(vmmul(Na, Na), vmmul(Nb, Nb));
    }
  }
  batch_inverse_lanes(nn, (long)k * LCH, ni, tmp);
  for (int t = 0; t < LCH; ++t) sa[t] = sb[t] = sc[t] = sd[t] = 0;
  for (int j = 0; j < k; ++j) {
    const u32 *pa = xa + (long)j * LCH, *pb = xb + (long)j * LCH, *pc = xc + (long)j * LCH, *pd = xd + (long)j * LCH;
    const u32 *qa = na + (long)j * LCH, *qb = nb + (long)j * LCH, *qi = ni + (long)j * LCH;
    const u32* m = mult[j] + r0;
    const int ng = neg[j];
    u32* o = out + (long)(4 * j) * n + r0;
    for (int t = 0; t < LCH; ++t) {
      /* di = conj(N) / |N|; 1/x = (A di, -B di) */
      const u32 da = vmmul(qa[t], qi[t]), db = vmmul(vmsub(0, qb[t]), qi[t]);
      const u32 ia = vmsub(vmmul(pa[t], da), vmmul(pb[t], db)), ib = vmadd(vmmul(pa[t], db), vmmul(pb[t], da));
      const u32 ic = vmsub(0, vmsub(vmmul(pc[t], da), vmmul(pd[t], db))), idd = vmsub(0, vmadd(vmmul(pc[t], db), vmmul(pd[t], da)));
      const u32 mm = ng ? vmsub(0, m[t]) : m[t];
      sa[t] = vmadd(sa[t], vmmul(ia, mm));
      sb[t] = vmadd(sb[t], vmmul(ib, mm));
      sc[t] = vmadd(sc[t], vmmul(ic, mm));
      sd[t] = vmadd(sd[t], vmmul(idd, mm));
      o[t] = sa[t]; o[n + t] = sb[t]; o[2 * n + t] = sc[t]; o[3 * n + t] = sd[t];
    }
  }
  u64 l0 = 0, l1 = 0, l2 = 0, l3 = 0;
  for (int t = 0; t < LCH; ++t) { l0 += sa[t]; l1 += sb[t]; l2 += sc[t]; l3 += sd[t]; }
  loc[0] += l0; loc[1] += l1; loc[2] += l2; loc[3] += l3;
}

void orc_logup_columns(const u32* const* val, const u32* const* id /* entries may be NULL */,
                       const u32* const* mult, int k, long n, const u32* zs /* 4 per relation */,
                       const u32* alphas /* 4 per relation */, const int* neg, u32* out, u32 claimed[4]) {
  u64 acc[4] = {0, 0, 0, 0};
  const long nchunks = n / LCH;
#pragma omp parallel if (par_ok(n * (long)k * 64))
  {
    u64 loc[4] = {0, 0, 0, 0};
    u32* scratch = (u32*)malloc(sizeof(u32) * ((size_t)k * 9 + 4) * LCH);
#pragma omp for schedule(static)
    for (long c = 0; c < nchunks; ++c) logup_chunk(val, id, mult, k, c * LCH, n, zs, alphas, neg, out, loc, scratch);
    free(scratch);
#pragma omp critical
    for (int t = 0; t < 4; ++t) acc[t] += loc[t] % P;
  }
  for (long r = nchunks * LCH; r < n; ++r) { /* tables below LCH rows: one inversion per row and relation */
    qm S = QZERO;
    for (int j = 0; j < k; ++j) {
      const qm Z = {zs[4 * j], zs[4 * j + 1], zs[4 * j + 2], zs[4 * j + 3]};
      const qm A = {alphas[4 * j], alphas[4 * j + 1], alphas[4 * j + 2], alphas[4 * j + 3]};
      qm den = qfromm(val[j][r]);
      if (id[j]) den = qadd(den, qmulm(A, id[j][r]));
      den = qsub(den, Z);
      u32 m = mult[j][r];
      if (neg[j]) m = mneg(m);
      S = qadd(S, qmulm(qinv(den), m));
      u32* o = out + (long)(4 * j) * n + r;
      o[0] = S.a; o[n] = S.b; o[2 * n] = S.c; o[3 * n] = S.d;
    }
    acc[0] += S.a; acc[1] += S.b; acc[2] += S.c; acc[3] += S.d;
  }
  for (int t = 0; t < 4; ++t) claimed[t] = (u32)(acc[t] % P);
}

/* last column: T[order[i]] = sum_{i' <= i} (S[order[i']] - shift), order = coset-order storage indices */
void orc_logup_prefix(u32* col4, long n, const int64_t* order, const u32 shift[4]) {
  for (int t = 0; t < 4; ++t) {
    u32* c = col4 + (long)t * n;
    u32 run = 0;
    for (long i = 0; i < n; ++i) {
      long s = order[i];
      run = madd(run, msub(c[s], shift[t]));
      c[s] = run;
    }
  }
}

/* ---------------------------------------------------------------------------------------------
 * Constraint quotients of one component on its eval domain (Appendix A.7).
 * kind: TraceTable variant index (all 17).  Relations are given by eval-domain column pointers (value, optional id, multiplicity)
 * with their element set (z, alpha) and numerator sign.  main: n_cols columns of E words; inter: 4*n_rel columns;
 * prev_idx[s] = storage index of the previous trace row of s; coeff: alpha powers (QM31) in
 * constraint order; zinv[s]: 1/Z per row; out: 4 x E (+= when accumulate).
 * ------------------------------------------------------------------------------------------- */
static int local_constraints(int kind, const u32* c, u32* out) {
  int k = 0;
  if (kind == 0 || kind == 1) {
    u32 is_last = c[4], nl = msub(1, is_last);
    out[k++] = mmul(is_last, msub(is_last, 1));
    if (kind == 0) {
      out[k++] = msub(c[11], madd(c[9], c[10]));
    } else {
      out[k++] = msub(mmul(c[9], c[10]), madd(mmul(c[11], 4096), c[12]));
      out[k++] = 0; /* second eval_fixed_mul slot: zero for rem == 0 (KAT-pinned), see oracle/air.py */
    }
    out[k++] = mmul(nl, msub(c[5], c[0]));
    out[k++] = mmul(nl, msub(c[6], c[1]));
    out[k++] = mmul(nl, msub(c[7], c[2]));
    out[k++] = mmul(nl, msub(msub(c[8], c[3]), 1));
  } else if (kind == 2) {
    u32 is_last = c[3], nl = msub(1, is_last);
    out[k++] = mmul(is_last, msub(is_last, 1));
    out[k++] = msub(mmul(c[10], c[10]), madd(mmul(c[7], c[8]), c[9]));
    out[k++] = mmul(nl, msub(c[4], c[0]));
    out[k++] = mmul(nl, msub(c[5], c[1]));
    out[k++] = mmul(nl, msub(msub(c[6], c[2]), 1));
  } else if (kind == 15) {
    u32 is_last = c[2], nl = msub(1, is_last);
    out[k++] = mmul(is_last, msub(is_last, 1));
    out[k++] = mmul(nl, msub(c[3], c[0]));
    out[k++] = mmul(nl, msub(msub(c[4], c[1]), 1));
  } else if (kind == 7) { /* Sqrt: eval_fixed_sqrt unpinned, input*scale = out^2 + rem */
    u32 is_last = c[3], nl = msub(1, is_last);
    out[k++] = mmul(is_last, msub(is_last, 1));
    out[k++] = msub(mmul(c[7], c[10]), madd(mmul(c[8], c[8]), c[9]));
    out[k++] = mmul(nl, msub(c[4], c[0]));
    out[k++] = mmul(nl, msub(c[5], c[1]));
    out[k++] = mmul(nl, msub(msub(c[6], c[2]), 1));
  } else if (kind == 8) { /* Rem: eval_fixed_rem unpinned, lhs = rhs*quotient + rem */
    u32 is_last = c[4], nl = msub(1, is_last);
    out[k++] = mmul(is_last, msub(is_last, 1));
    out[k++] = msub(c[9], madd(mmul(c[10], c[12]), c[11]));
    out[k++] = mmul(nl, msub(c[5], c[0]));
    out[k++] = mmul(nl, msub(c[6], c[1]));
    out[k++] = mmul(nl, msub(c[7], c[2]));
    out[k++] = mmul(nl, msub(msub(c[8], c[3]), 1));
  } else if (kind == 14 || kind == 4 || kind == 10 || kind == 12) {
    /* RangeCheckLookup / SinLookup / Exp2Lookup / Log2Lookup: no local constraints */
  } else if (kind == 13) { /* LessThan, less_than/component.rs:48-185 */
    u32 is_last = c[4], nl = msub(1, is_last), borrow = c[13];
    out[k++] = mmul(is_last, msub(is_last, 1));
    out[k++] = mmul(borrow, msub(borrow, 1));
    out[k++] = msub(c[11], mmul(msub(1, borrow), 4096));
    out[k++] = msub(madd(c[9], c[12]), c[10]); /* borrow * (2^31-1) == 0 in M31 */
    {
      u32 rec = madd(madd(mmul(c[17], 1u << 24), mmul(c[16], 1u << 16)), madd(mmul(c[15], 1u << 8), c[14]));
      out[k++] = msub(c[12], rec);
    }
    out[k++] = mmul(nl, msub(c[5], c[0]));
    out[k++] = mmul(nl, msub(c[6], c[1]));
    out[k++] = mmul(nl, msub(c[7], c[2]));
    out[k++] = mmul(nl, msub(msub(c[8], c[3]), 1));
  } else { /* 5 SumReduce, 6 MaxReduce, 16 Contiguous, 3 Sin, 9 Exp2, 11 Log2: shared id/idx prefix, columns 0..6 */
    u32 is_last = c[3], nl = msub(1, is_last);
    out[k++] = mmul(is_last, msub(is_last, 1));
    if (kind == 5) {
      u32 ils = c[11];
      out[k++] = mmul(ils, msub(ils, 1));
      out[k++] = msub(c[10], madd(c[9], c[7]));
      out[k++] = mmul(msub(c[8], c[10]), ils);
    } else if (kind == 6) {
      u32 ils = c[11], im = c[12];
      out[k++] = mmul(ils, msub(ils, 1));
      out[k++] = mmul(im, msub(im, 1));
      out[k++] = mmul(im, msub(c[10], c[7]));
      out[k++] = mmul(msub(1, im), msub(c[10], c[9]));
      out[k++] = mmul(msub(c[8], c[10]), ils);
    }
    out[k++] = mmul(nl, msub(c[4], c[0]));
    out[k++] = mmul(nl, msub(c[5], c[1]));
    out[k++] = mmul(nl, msub(msub(c[6], c[2]), 1));
  }
  return k;
}

void orc_composition(int kind, int n_cols, int n_rel, const u32* const* rel_val, const u32* const* rel_id,
                     const u32* const* rel_mult, const u32* rel_z, const u32* rel_alpha, const int* rel_neg,
                     const u32* main, const u32* inter, long E, const int64_t* prev_idx, const u32 shift[4],
                     const u32* coeff /* 4 words each */, const u32* zinv, u32* out, int accumulate) {
  const qm SH = {shift[0], shift[1], shift[2], shift[3]};
#pragma omp parallel for schedule(static) if (par_ok(E * 64))
  for (long s = 0; s < E; ++s) {
    u32 c[32], lc[16];
    for (int k = 0; k < n_cols; ++k) c[k] = main[(long)k * E + s];
    int nl = local_constraints(kind, c, lc);
    qm acc = QZERO;
    int k = 0;
    for (; k < nl; ++k) {
      qm cf = {coeff[4 * k], coeff[4 * k + 1], coeff[4 * k + 2], coeff[4 * k + 3]};
      acc = qadd(acc, qmulm(cf, lc[k]));
    }
    qm prev = QZERO;
    for (int j = 0; j < n_rel; ++j, ++k) {
      const u32* b = inter + (long)(4 * j) * E;
      const qm Z = {rel_z[4 * j], rel_z[4 * j + 1], rel_z[4 * j + 2], rel_z[4 * j + 3]};
      const qm A = {rel_alpha[4 * j], rel_alpha[4 * j + 1], rel_alpha[4 * j + 2], rel_alpha[4 * j + 3]};
      qm cur = {b[s], b[E + s], b[2 * E + s], b[3 * E + s]};
      qm den = qfromm(rel_val[j][s]);
      if (rel_id[j]) den = qadd(den, qmulm(A, rel_id[j][s]));
      den = qsub(den, Z);
      u32 m = rel_mult[j][s];
      if (rel_neg[j]) m = mneg(m);
      qm diff;
      if (j < n_rel - 1) {
        diff = qsub(cur, prev);
      } else {
        long ps = prev_idx[s];
        qm pr = {b[ps], b[E + ps], b[2 * E + ps], b[3 * E + ps]};
        diff = qadd(qsub(qsub(cur, pr), prev), SH);
      }
      qm cons = qsub(qmul(diff, den), qfromm(m));
      qm cf = {coeff[4 * k], coeff[4 * k + 1], coeff[4 * k + 2], coeff[4 * k + 3]};
      acc = qadd(acc, qmul(cons, cf));
      prev = cur;
    }
    acc = qmulm(acc, zinv[s]);
    u32* o = out + s;
    if (accumulate) {
      acc.a = madd(acc.a, o[0]); acc.b = madd(acc.b, o[E]); acc.c = madd(acc.c, o[2 * E]); acc.d = madd(acc.d, o[3 * E]);
    }
    o[0] = acc.a; o[E] = acc.b; o[2 * E] = acc.c; o[3 * E] = acc.d;
  }
}

/* ---------------------------------------------------------------------------------------------
 * eval_at_point: sum_j coeff_j * prod_k maps[k]^(bit k of j)  (Appendix A.2 basis; Horner-style fold)
 * ------------------------------------------------------------------------------------------- */
/* The basis value of coefficient index j factors into a table over the low EV_LB bits of j and a table over the high
 * bits: f(p) = sum_hi hiT[hi] * (sum_lo c[hi, lo] * loT[lo]).  The inner sums are M31 x QM31 dot products accumulated
 * lazily in 64 bits (each product folded once to < 2^32, so 2^EV_LB terms stay far below 2^64). */
#define EV_LB 10
HOT static void eval_dot(const u32* c, long lo_n, const u32* la, const u32* lb, const u32* lc, const u32* ld, u64 acc[4]) {
  u64 s0 = 0, s1 = 0, s2 = 0, s3 = 0;
  for (long k = 0; k < lo_n; ++k) {
    const u64 v = c[k];
    const u64 p0 = v * la[k], p1 = v * lb[k], p2 = v * lc[k], p3 = v * ld[k];
    s0 += (p0 & P) + (p0 >> 31);
    s1 += (p1 & P) + (p1 >> 31);
    s2 += (p2 & P) + (p2 >> 31);
    s3 += (p3 & P) + (p3 >> 31);
  }
  acc[0] = s0; acc[1] = s1; acc[2] = s2; acc[3] = s3;
}
void orc_eval_at_point(const u32* coeffs, int log_n, const u32* maps /* log_n x 4 */, u32 out[4]) {
  const long n = 1L << log_n;
  const int lbits = log_n < EV_LB ? log_n : EV_LB;
  const long lo_n = 1L << lbits, hi_n = n >> lbits;
  u32* lo = (u32*)malloc(sizeof(u32) * 4 * (size_t)lo_n);
  qm* hiT = (qm*)malloc(sizeof(qm) * (size_t)hi_n);
  u32 *la = lo, *lb = lo + lo_n, *lc = lo + 2 * lo_n, *ld = lo + 3 * lo_n;
  la[0] = 1; lb[0] = lc[0] = ld[0] = 0;
  for (int k = 0; k < lbits; ++k) {
    const qm m = {maps[4 * k], maps[4 * k + 1], maps[4 * k + 2], maps[4 * k + 3]};
    for (long j = 0; j < (1L << k); ++j) {
      const qm t = {la[j], lb[j], lc[j], ld[j]};
      const qm r = qmul(t, m);
      const long d = j + (1L << k);
      la[d] = r.a; lb[d] = r.b; lc[d] = r.c; ld[d] = r.d;
    }
  }
  hiT[0] = qfromm(1);
  for (int k = lbits; k < log_n; ++k) {
    const qm m = {maps[4 * k], maps[4 * k + 1], maps[4 * k + 2], maps[4 * k + 3]};
    const long half = 1L << (k - lbits);
    for (long j = 0; j < half; ++j) hiT[j + half] = qmul(hiT[j], m);
  }
  u64 tot[4] = {0, 0, 0, 0};
#pragma omp parallel if (par_ok(n * 8))
  {
    u64 loc[4] = {0, 0, 0, 0};
#pragma omp for schedule(static)
    for (long h = 0; h < hi_n; ++h) {
      u64 acc[4];
      eval_dot(coeffs + h * lo_n, lo_n, la, lb, lc, ld, acc);
      const qm inner = {(u32)(acc[0] % P), (u32)(acc[1] % P), (u32)(acc[2] % P), (u32)(acc[3] % P)};
      const qm t = qmul(inner, hiT[h]);
      loc[0] += t.a; loc[1] += t.b; loc[2] += t.c; loc[3] += t.d;
    }
#pragma omp critical
    for (int t = 0; t < 4; ++t) tot[t] += loc[t] % P;
  }
  for (int t = 0; t < 4; ++t) out[t] = (u32)(tot[t] % P);
  free(lo);
  free(hiT);
}

/* ---------------------------------------------------------------------------------------------
 * FRI quotients over one LDE domain (Appendix A.8).  Batches: for batch b, entries
 * [bstart[b], bstart[b+1]) of (col index, a, b, c) line coefficients (already alpha-weighted);
 * point (px, py) as QM31; batch_coeff = alpha^|batch|.  xs/ys: domain points in storage order.
 * ------------------------------------------------------------------------------------------- */
/* One chunk of QCH rows, all batches.  The denominators are inverted together (stwo's batch inverse of the CM31
 * denominators: here through their norms, batch_inverse_lanes); the sums over a batch's entries run lane-wise.
 * sum_e (C_e f_e(s) - (A_e y + B_e)) = sum_e C_e f_e(s) - (SA y + SB) with SA, SB the batch's sums of A_e, B_e. */
#define QCH 1024
HOT static void quotients_chunk(const u32* const* cols, long s0, long L, int nbatch, const int* bstart, const int* col_idx,
                                const u32* lc, const u32* sa /* 4 per batch */, const u32* sb, const u32* pts,
                                const u32* batch_coeff, const u32* xs, const u32* ys, u32* out, u32* scratch) {
  u32* dre = scratch;                      /* nbatch x QCH: denominator real / imaginary part, then its norm's inverse */
  u32* dim = dre + (long)nbatch * QCH;
  u32* nrm = dim + (long)nbatch * QCH;
  u32* ninv = nrm + (long)nbatch * QCH;
  u32* tmp = ninv + (long)nbatch * QCH;
  u32* ra = tmp + (long)nbatch * QCH;      /* row accumulator, 4 x QCH */
  u32 *rb = ra + QCH, *rc = rb + QCH, *rd = rc + QCH;
  u32 *na = rd + QCH, *nb = na + QCH, *nc = nb + QCH, *nd = nc + QCH;   /* numerator, 4 x QCH */
  const u32* x = xs + s0;
  const u32* y = ys + s0;
  for (int b = 0; b < nbatch; ++b) {
    const u32* p = pts + 8 * b;
    const u32 prx_a = p[0], prx_b = p[1], pix_a = p[2], pix_b = p[3], pry_a = p[4], pry_b = p[5], piy_a = p[6], piy_b = p[7];
    u32* re = dre + (long)b * QCH;
    u32* im = dim + (long)b * QCH;
    u32* nn = nrm + (long)b * QCH;
    for (int k = 0; k < QCH; ++k) {
      /* den = (prx - x) * piy - (pry - y) * pix  in CM31, dx = (prx_a - x, prx_b), dy = (pry_a - y, pry_b) */
      const u32 dxa = vmsub(prx_a, x[k]), dya = vmsub(pry_a, y[k]);
      const u32 t1a = vmsub(vmmul(dxa, piy_a), vmmul(prx_b, piy_b)), t1b = vmadd(vmmul(dxa, piy_b), vmmul(prx_b, piy_a));
      const u32 t2a = vmsub(vmmul(dya, pix_a), vmmul(pry_b, pix_b)), t2b = vmadd(vmmul(dya, pix_b), vmmul(pry_b, pix_a));
      const u32 a = vmsub(t1a, t2a), bb = vmsub(t1b, t2b);
      re[k] = a;
      im[k] = bb;
      nn[k] = vmadd(vmmul(a, a), vmmul(bb, bb));
    }
  }
  batch_inverse_lanes(nrm, (long)nbatch * QCH, ninv, tmp);
  for (int k = 0; k < QCH; ++k) ra[k] = rb[k] = rc[k] = rd[k] = 0;
  for (int b = 0; b < nbatch; ++b) {
    for (int k = 0; k < QCH; ++k) na[k] = nb[k] = nc[k] = nd[k] = 0;
    for (int e = bstart[b]; e < bstart[b + 1]; ++e) {
      const u32 ca = lc[4 * e], cb = lc[4 * e + 1], cc = lc[4 * e + 2], cd = lc[4 * e + 3];
      const u32* f = cols[col_idx[e]] + s0;
      for (int k = 0; k < QCH; ++k) {
        const u32 v = f[k];
        na[k] = vmadd(na[k], vmmul(ca, v));
        nb[k] = vmadd(nb[k], vmmul(cb, v));
        nc[k] = vmadd(nc[k], vmmul(cc, v));
        nd[k] = vmadd(nd[k], vmmul(cd, v));
      }
    }
    const u32* A = sa + 4 * b;
    const u32* B = sb + 4 * b;
    const u32* bc = batch_coeff + 4 * b;
    const u32* re = dre + (long)b * QCH;
    const u32* im = dim + (long)b * QCH;
    const u32* ni = ninv + (long)b * QCH;
    for (int k = 0; k < QCH; ++k) {
      const u32 yy = y[k];
      /* num -= SA * y + SB */
      const u32 n0 = vmsub(na[k], vmadd(vmmul(A[0], yy), B[0])), n1 = vmsub(nb[k], vmadd(vmmul(A[1], yy), B[1]));
      const u32 n2 = vmsub(nc[k], vmadd(vmmul(A[2], yy), B[2])), n3 = vmsub(nd[k], vmadd(vmmul(A[3], yy), B[3]));
      /* 1/den = conj(den) / norm */
      const u32 ia = vmmul(re[k], ni[k]), ib = vmmul(vmsub(0, im[k]), ni[k]);
      /* (n0 + n1 i) * inv, (n2 + n3 i) * inv */
      const u32 q0 = vmsub(vmmul(n0, ia), vmmul(n1, ib)), q1 = vmadd(vmmul(n0, ib), vmmul(n1, ia));
      const u32 q2 = vmsub(vmmul(n2, ia), vmmul(n3, ib)), q3 = vmadd(vmmul(n2, ib), vmmul(n3, ia));
      /* row = row * bc + q   (QM31 product with the constant bc = (e + f i) + (g + h i) u, u^2 = 2 + i) */
      const u32 a_ = ra[k], b_ = rb[k], c_ = rc[k], d_ = rd[k];
      const u32 e_ = bc[0], f_ = bc[1], g_ = bc[2], h_ = bc[3];
      /* A*C */
      const u32 ac_a = vmsub(vmmul(a_, e_), vmmul(b_, f_)), ac_b = vmadd(vmmul(a_, f_), vmmul(b_, e_));
      /* B*D */
      const u32 bd_a = vmsub(vmmul(c_, g_), vmmul(d_, h_)), bd_b = vmadd(vmmul(c_, h_), vmmul(d_, g_));
      /* (2 + i) * BD */
      const u32 r_a = vmsub(vmadd(bd_a, bd_a), bd_b), r_b = vmadd(bd_a, vmadd(bd_b, bd_b));
      /* A*D + B*C */
      const u32 ad_a = vmsub(vmmul(a_, g_), vmmul(b_, h_)), ad_b = vmadd(vmmul(a_, h_), vmmul(b_, g_));
      const u32 bc_a = vmsub(vmmul(c_, e_), vmmul(d_, f_)), bc_b = vmadd(vmmul(c_, f_), vmmul(d_, e_));
      ra[k] = vmadd(vmadd(ac_a, r_a), q0);
      rb[k] = vmadd(vmadd(ac_b, r_b), q1);
      rc[k] = vmadd(vmadd(ad_a, bc_a), q2);
      rd[k] = vmadd(vmadd(ad_b, bc_b), q3);
    }
  }
  for (int k = 0; k < QCH; ++k) {
    out[s0 + k] = ra[k];
    out[L + s0 + k] = rb[k];
    out[2 * L + s0 + k] = rc[k];
    out[3 * L + s0 + k] = rd[k];
  }
}

void orc_quotients(const u32* const* cols, long L, int nbatch, const int* bstart, const int* col_idx,
                   const u32* la, const u32* lb, const u32* lc /* 4 words per entry */, const u32* pts /* 8/batch */,
                   const u32* batch_coeff /* 4/batch */, const u32* xs, const u32* ys, u32* out) {
  /* per batch: SA = sum of the entries' a coefficients, SB = sum of the b coefficients */
  u32* sa = (u32*)calloc((size_t)nbatch * 8, sizeof(u32));
  u32* sb = sa + (size_t)nbatch * 4;
  for (int b = 0; b < nbatch; ++b)
    for (int e = bstart[b]; e < bstart[b + 1]; ++e)
      for (int t = 0; t < 4; ++t) {
        sa[4 * b + t] = madd(sa[4 * b + t], la[4 * e + t]);
        sb[4 * b + t] = madd(sb[4 * b + t], lb[4 * e + t]);
      }
  const long nchunks = L / QCH;
#pragma omp parallel if (par_ok(L * 64))
  {
    u32* scratch = (u32*)malloc(sizeof(u32) * ((size_t)nbatch * 5 + 8) * QCH);
#pragma omp for schedule(static)
    for (long c = 0; c < nchunks; ++c)
      quotients_chunk(cols, c * QCH, L, nbatch, bstart, col_idx, lc, sa, sb, pts, batch_coeff, xs, ys, out, scratch);
    free(scratch);
  }
  /* domains below QCH rows (and a ragged tail, which power-of-two domains do not have): one row at a time */
  for (long s = nchunks * QCH; s < L; ++s) {
    const u32 x = xs[s], y = ys[s];
    qm row = QZERO;
    for (int b = 0; b < nbatch; ++b) {
      qm num = QZERO;
      for (int e = bstart[b]; e < bstart[b + 1]; ++e) {
        qm C = {lc[4 * e], lc[4 * e + 1], lc[4 * e + 2], lc[4 * e + 3]};
        num = qadd(num, qmulm(C, cols[col_idx[e]][s]));
      }
      qm SA = {sa[4 * b], sa[4 * b + 1], sa[4 * b + 2], sa[4 * b + 3]}, SB = {sb[4 * b], sb[4 * b + 1], sb[4 * b + 2], sb[4 * b + 3]};
      num = qsub(num, qadd(qmulm(SA, y), SB));
      const u32* p = pts + 8 * b;
      cm prx = {p[0], p[1]}, pix = {p[2], p[3]}, pry = {p[4], p[5]}, piy = {p[6], p[7]};
      cm dx = {msub(prx.a, x), prx.b}, dy = {msub(pry.a, y), pry.b};
      cm den = csub(cmul(dx, piy), cmul(dy, pix));
      qm bc = {batch_coeff[4 * b], batch_coeff[4 * b + 1], batch_coeff[4 * b + 2], batch_coeff[4 * b + 3]};
      row = qadd(qmul(row, bc), qmulc(num, cinv(den)));
    }
    out[s] = row.a; out[L + s] = row.b; out[2 * L + s] = row.c; out[3 * L + s] = row.d;
  }
  free(sa);
}

/* FRI fold of adjacent pairs: dst[i] = [dst[i]*alpha^2 +] (a+b) + alpha*((a-b)*itw[i])  (Appendix A.8) */
void orc_fold(u32* dst, const u32* src, long src_len, const u32* itw, const u32 alpha[4], int accumulate) {
  const qm AL = {alpha[0], alpha[1], alpha[2], alpha[3]};
  const qm AL2 = qmul(AL, AL);
  const long n = src_len / 2, L = src_len;
#pragma omp parallel for schedule(static) if (par_ok(n * 16))
  for (long i = 0; i < n; ++i) {
    qm a = {src[2 * i], src[L + 2 * i], src[2 * L + 2 * i], src[3 * L + 2 * i]};
    qm b = {src[2 * i + 1], src[L + 2 * i + 1], src[2 * L + 2 * i + 1], src[3 * L + 2 * i + 1]};
    qm r = qadd(qadd(a, b), qmul(AL, qmulm(qsub(a, b), itw[i])));
    if (accumulate) {
      qm d = {dst[i], dst[n + i], dst[2 * n + i], dst[3 * n + i]};
      r = qadd(qmul(d, AL2), r);
    }
    dst[i] = r.a; dst[n + i] = r.b; dst[2 * n + i] = r.c; dst[3 * n + i] = r.d;
  }
}

/* One Merkle layer straight from column pointers (Appendix A.4):
 * out[i] = blake2s(prev[2i] || prev[2i+1] || cols[0][i] .. cols[ncols-1][i]); prev may be NULL. */
void orc_merkle_layer(const u32* prev, const u32* const* cols, int ncols, long size, u32* out) {
  const int npre = prev ? 16 : 0;
  const int w = npre + ncols;
  const int nblocks = w == 0 ? 1 : (w + 15) / 16;
  const long groups = size / LW;
#pragma omp parallel for schedule(static) if (par_ok(size * (long)(w + 8) * 8))
  for (long g = 0; g < groups; ++g) merkle_lanes(prev, cols, ncols, g * LW, out);
  for (long i = groups * LW; i < size; ++i) { /* ragged tail (and layers narrower than LW): one hash at a time */
    u32 h[8];
    for (int k = 0; k < 8; ++k) h[k] = IV[k];
    h[0] ^= 0x01010020u;
    for (int b = 0; b < nblocks; ++b) {
      u32 m[16];
      for (int k = 0; k < 16; ++k) {
        int j = 16 * b + k;
        m[k] = j < npre ? prev[16 * i + j] : (j < w ? cols[j - npre][i] : 0);
      }
      int last = b + 1 == nblocks;
      compress(h, m, last ? (u32)(4 * w) : (u32)(64 * (b + 1)), last ? 0xffffffffu : 0);
    }
    memcpy(out + 8 * i, h, 32);
  }
}

/* `write_trace` (add/witness.rs:33-108): AoS rows -> SoA columns of `size` rows, rows past n_rows = the padding row.
 * Returns 1 if a word is not a canonical M31. */
int orc_transpose_pad(const u32* rows, long n_rows, int ncols, long size, const u32* pad, u32* cols) {
  int bad = 0;
#pragma omp parallel for schedule(static) reduction(| : bad) if (par_ok(size * (long)ncols))
  for (long r0 = 0; r0 < size; r0 += 64) {
    const long r1 = r0 + 64 < size ? r0 + 64 : size;
    for (int c = 0; c < ncols; ++c)
      for (long r = r0; r < r1; ++r) {
        const u32 v = r < n_rows ? rows[r * (long)ncols + c] : pad[c];
        bad |= v >= P;
        cols[(long)c * size + r] = v;
      }
  }
  return bad;
}

/* out[i] = 1/v[i] (Montgomery batch inversion in chunks, chunks in parallel) */
void orc_batch_inverse(const u32* v, long n, u32* out) {
  const long chunk = 4096;
#pragma omp parallel for schedule(static) if (par_ok(n * 4))
  for (long c0 = 0; c0 < n; c0 += chunk) {
    long c1 = c0 + chunk < n ? c0 + chunk : n;
    u32 pre[4096];
    u32 acc = 1;
    for (long i = c0; i < c1; ++i) { pre[i - c0] = acc; acc = mmul(acc, v[i]); }
    u32 inv = minv(acc);
    for (long i = c1; i-- > c0;) { out[i] = mmul(inv, pre[i - c0]); inv = mmul(inv, v[i]); }
  }
}

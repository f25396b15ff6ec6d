// gfx950 kernels, part 2: the generic circle FFT passes (run-time tile geometry; the fixed shapes the prover's columns
// mostly have are in fft_fixed.hip) and the twiddle tables.
#include "kernels_common.h"

namespace lmn {

// =============================================================================================
// a4  Circle FFT.  Layer i pairs indices differing in bit i; twiddle index = idx >> (i+1).
//     A pass runs layers [lo, hi) on LDS tiles of 2^(hi-lo) rows x 2^cb contiguous words.
// =============================================================================================
// src may differ from data (out-of-place first pass); words at index >= src_len read as zero
// (zero-extension of a coefficient vector onto a larger domain, i.e. the LDE).
template <bool INV>
LMN_KERNEL k_fft_pass(uint32_t* data, uint64_t col_stride, const uint32_t* src,
                      uint64_t src_stride, uint64_t src_len, int lo, int hi, int cb, TwPtrs tw, uint32_t scale) {
  LMN_DYN_SMEM(uint32_t, sm);
  const int rbits = hi - lo;
  const int C = 1 << cb;
  const int tile_elems = 1 << (rbits + cb);
  uint32_t* col = data + (uint64_t)blockIdx.y * col_stride;
  const uint32_t* scol = src + (uint64_t)blockIdx.y * src_stride;
  const uint32_t tile = blockIdx.x;
  const uint32_t q = tile & ((1u << (lo - cb)) - 1u);
  const uint32_t H = tile >> (lo - cb);
  const uint64_t base = ((uint64_t)H << hi) + ((uint64_t)q << cb);
  for (int e = threadIdx.x; e < tile_elems; e += blockDim.x) {
    int m = e >> cb, c = e & (C - 1);
    uint64_t gi = base + ((uint64_t)m << lo) + c;
    sm[e] = gi < src_len ? scol[gi] : 0u;
  }
  __syncthreads();
  for (int step = 0; step < rbits; ++step) {
    const int i = INV ? lo + step : hi - 1 - step;
    const int bit = i - lo;
    const uint32_t* __restrict__ t = tw.l[i];
    const uint32_t hbase = H << (hi - i - 1);
    for (int b = threadIdx.x; b < tile_elems / 2; b += blockDim.x) {
      int c = b & (C - 1);
      int p = b >> cb;
      int m0 = ((p >> bit) << (bit + 1)) | (p & ((1 << bit) - 1));
      int m1 = m0 | (1 << bit);
      uint32_t w = t[hbase + (uint32_t)(m0 >> (bit + 1))];
      int i0 = (m0 << cb) | c, i1 = (m1 << cb) | c;
      uint32_t v0 = sm[i0], v1 = sm[i1];
      if (INV) {
        sm[i0] = m_add(v0, v1);
        sm[i1] = m_mul(m_sub(v0, v1), w);
      } else {
        uint32_t x = m_mul(v1, w);
        sm[i0] = m_add(v0, x);
        sm[i1] = m_sub(v0, x);
      }
    }
    __syncthreads();
  }
  for (int e = threadIdx.x; e < tile_elems; e += blockDim.x) {
    int m = e >> cb, c = e & (C - 1);
    uint32_t v = sm[e];
    if (INV && scale != 1u) v = m_mul(v, scale);
    col[base + ((uint64_t)m << lo) + c] = v;
  }
}

// ---------------------------------------------------------------------------------------------
// Staged variant: each lane keeps 2^R points (R <= 4) in registers and runs R butterfly layers
// on them before exchanging through LDS, so a 12-layer tile needs 2 LDS exchanges instead of 12
// and the per-layer index arithmetic disappears.  Tile geometry as in k_fft_pass.
// ---------------------------------------------------------------------------------------------
constexpr int FFT_MAX_STAGES = 4;
struct FftStagePlan {
  int lo, hi, cb;
  int nst;
  int first[FFT_MAX_STAGES];  // first layer of each stage, ascending
  int R[FFT_MAX_STAGES];      // layers per stage (1..4)
  int xcd_swizzle;
};

LMN_HD uint32_t fft_lds_pad(uint32_t e) { return e + (e >> 5); }

// out[k] = x[k] * w[k] in M31 for N independent products, issued in phases (field.h m_mul, same arithmetic).  Leaves the
// wave in the first-port phase.
template <int N>
LMN_D void m_mul_phased(const uint32_t (&x)[N], const uint32_t (&w)[N], uint32_t (&out)[N]) {
  uint64_t pr[N];
  uint32_t hi[N], s[N], s2[N];
  LMN_PHASE_PORT0();
#pragma unroll
  for (int k = 0; k < N; ++k) pr[k] = (uint64_t)x[k] * (uint64_t)w[k];
#pragma unroll
  for (int k = 0; k < N; ++k) hi[k] = (uint32_t)(pr[k] >> 31);
  LMN_PHASE_ANY();
#pragma unroll
  for (int k = 0; k < N; ++k) {
    s[k] = ((uint32_t)pr[k] & P31) + hi[k];
    s2[k] = s[k] - P31;
  }
  LMN_PHASE_PORT0();
#pragma unroll
  for (int k = 0; k < N; ++k) out[k] = s[k] < s2[k] ? s[k] : s2[k];
}

// The 2^R - 1 twiddles of a register stage (layer r needs 2^(R-1-r) of them: butterfly k of the layer uses entry k >> r).
// Loaded in one go BEFORE the stage's data so that a single memory latency covers all R layers - loading them layer by
// layer put one exposed L2 round trip in front of every layer (the kernels run at 2.6 - 4 waves per SIMD).
template <int R>
struct StageTwiddles {
  uint32_t t[(1 << R) - 1];  // layers 0 .. R-1 one after the other
  static constexpr int at(int r) { return (1 << R) - (1 << (R - r)); }  // first entry of layer r
};

template <int R, bool INV>
LMN_D void load_stage_twiddles(StageTwiddles<R>& T, const TwPtrs& tw, int first_layer, int hi, uint32_t H, uint32_t mhigh) {
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int L = first_layer + r;
    const uint32_t* __restrict__ t = tw.l[L];
    const uint32_t hb = (H << (hi - L - 1)) + (mhigh << (R - 1 - r));
    LMN_ASSUME(hb < (1u << 28));  // lets the compiler use 32-bit offsets from the uniform table pointer
#pragma unroll
    for (int q = 0; q < (1 << (R - 1 - r)); ++q) T.t[StageTwiddles<R>::at(r) + q] = t[hb + (uint32_t)q];
  }
}

template <int R, bool INV>
LMN_D void radix_butterflies(uint32_t (&v)[1 << R], const StageTwiddles<R>& T) {
  constexpr int NB = 1 << (R - 1);  // independent butterflies per layer
#pragma unroll
  for (int rr = 0; rr < R; ++rr) {
    const int r = INV ? rr : R - 1 - rr;
    uint32_t w[NB], a[NB], b[NB], x[NB], u[NB], u2[NB], d[NB], d2[NB];
#pragma unroll
    for (int k = 0; k < NB; ++k) {
      const int j = ((k >> r) << (r + 1)) | (k & ((1 << r) - 1));  // k-th index with bit r clear; j >> (r+1) == k >> r
      w[k] = T.t[StageTwiddles<R>::at(r) + (k >> r)];
      a[k] = v[j];
      b[k] = v[j | (1 << r)];
    }
    if (INV) {
      LMN_PHASE_ANY();
#pragma unroll
      for (int k = 0; k < NB; ++k) {
        u[k] = a[k] + b[k];
        u2[k] = u[k] - P31;
        d[k] = a[k] - b[k];
        d2[k] = d[k] + P31;
      }
      LMN_PHASE_PORT0();
#pragma unroll
      for (int k = 0; k < NB; ++k) {
        a[k] = u[k] < u2[k] ? u[k] : u2[k];   // m_add(a, b)
        d[k] = d[k] < d2[k] ? d[k] : d2[k];   // m_sub(a, b)
      }
      m_mul_phased<NB>(d, w, x);
#pragma unroll
      for (int k = 0; k < NB; ++k) {
        const int j = ((k >> r) << (r + 1)) | (k & ((1 << r) - 1));
        v[j] = a[k];
        v[j | (1 << r)] = x[k];
      }
    } else {
      m_mul_phased<NB>(b, w, x);
      LMN_PHASE_ANY();
#pragma unroll
      for (int k = 0; k < NB; ++k) {
        u[k] = a[k] + x[k];
        u2[k] = u[k] - P31;
        d[k] = a[k] - x[k];
        d2[k] = d[k] + P31;
      }
      LMN_PHASE_PORT0();
#pragma unroll
      for (int k = 0; k < NB; ++k) {
        const int j = ((k >> r) << (r + 1)) | (k & ((1 << r) - 1));
        v[j] = u[k] < u2[k] ? u[k] : u2[k];              // m_add(a, x)
        v[j | (1 << r)] = d[k] < d2[k] ? d[k] : d2[k];   // m_sub(a, x)
      }
    }
  }
  LMN_PHASE_ANY();
}

// sm_in / sm_out: LDS tile read by a stage that does not load from global / written by one that does not store to
// global (the same buffer in the plain passes).  keep != nullptr: a to_global stage also leaves its (scaled) values in
// that LDS tile (the fused interpolate + extend pass continues from them).
template <int R, bool INV>
LMN_D void fft_stage(const uint32_t* sm_in, uint32_t* sm_out, uint32_t* col, const uint32_t* scol, uint64_t src_len,
                     uint64_t base, int lo, int hi, int cb, int first_layer, uint32_t H, bool from_global, bool to_global,
                     const TwPtrs& tw, uint32_t scale, uint32_t* keep = nullptr) {
  const int p = first_layer - lo + cb;           // bit position of the stage's first layer in the tile index
  const uint32_t tile_elems = 1u << (hi - lo + cb);
  const uint32_t ngroups = tile_elems >> R;
  const uint32_t cmask = (1u << cb) - 1u;
  // 32-bit offsets from the (block-uniform) tile base keep the address arithmetic off the 64-bit path
  const uint32_t* __restrict__ tsrc = scol + base;
  uint32_t* __restrict__ tdst = col + base;
  const uint64_t span = src_len > base ? src_len - base : 0;
  const uint32_t lim = span > 0x10000000ull ? 0x10000000u : (uint32_t)span;  // readable words from tsrc
  // The 2^R points of a group sit at tile index e0 + (j << p) (bits [p, p+R) of e0 are zero), so both
  // address maps split into a per-lane base plus a wave-uniform term in j: global offset
  // ((e >> cb) << lo) + (e & cmask) has stride 2^(p - cb + lo) (p >= cb always), and the padded LDS index
  // pad(e) = e + (e >> 5) satisfies pad(e0 + (j << p)) = pad(e0) + pad(j << p) because the two addends
  // occupy disjoint bits (no carry into bit 5).  One add per point instead of re-deriving each address.
  const uint32_t gstride = 1u << (p - cb + lo);
  const uint32_t tile_span = (((tile_elems - 1u) >> cb) << lo) + cmask + 1u;  // words of [tsrc, ...) the tile touches
  const bool full = tile_span <= lim;                                           // block-uniform: no zero-extension here
  for (uint32_t g = threadIdx.x; g < ngroups; g += blockDim.x) {
    const uint32_t e0 = ((g >> p) << (p + R)) | (g & ((1u << p) - 1u));
    const uint32_t off0 = ((e0 >> cb) << lo) + (e0 & cmask);
    LMN_ASSUME(off0 < 0x10000000u);
    const uint32_t m0 = e0 >> cb;
    StageTwiddles<R> T;
    load_stage_twiddles<R, INV>(T, tw, first_layer, hi, H, m0 >> (first_layer - lo + R));
    uint32_t v[1 << R];
    if (from_global) {
      if (p == 0 && cb == 0 && R >= 2) {
        if (e0 < lim) {
          const uint4* q = reinterpret_cast<const uint4*>(tsrc + e0);
#pragma unroll
          for (int k = 0; k < (1 << R) / 4; ++k) {
            uint4 x = q[k];
            v[4 * k] = x.x;
            v[4 * k + 1] = x.y;
            v[4 * k + 2] = x.z;
            v[4 * k + 3] = x.w;
          }
        } else {
#pragma unroll
          for (int j = 0; j < (1 << R); ++j) v[j] = 0u;
        }
      } else if (full) {
#pragma unroll
        for (int j = 0; j < (1 << R); ++j) v[j] = tsrc[off0 + (uint32_t)j * gstride];
      } else {
#pragma unroll
        for (int j = 0; j < (1 << R); ++j) {
          const uint32_t off = off0 + (uint32_t)j * gstride;
          v[j] = off < lim ? tsrc[off] : 0u;
        }
      }
    } else {
      const uint32_t pb = fft_lds_pad(e0);
#pragma unroll
      for (int j = 0; j < (1 << R); ++j) v[j] = sm_in[pb + fft_lds_pad((uint32_t)j << p)];
    }
    radix_butterflies<R, INV>(v, T);
    if (to_global) {
      if (INV && scale != 1u) {
        uint32_t sc[1 << R], pr[1 << R];
#pragma unroll
        for (int j = 0; j < (1 << R); ++j) sc[j] = scale;
        m_mul_phased<(1 << R)>(v, sc, pr);
#pragma unroll
        for (int j = 0; j < (1 << R); ++j) v[j] = pr[j];
        LMN_PHASE_ANY();
      }
      if (p == 0 && cb == 0 && R >= 2) {
        uint4* q = reinterpret_cast<uint4*>(tdst + e0);
#pragma unroll
        for (int k = 0; k < (1 << R) / 4; ++k) q[k] = make_uint4(v[4 * k], v[4 * k + 1], v[4 * k + 2], v[4 * k + 3]);
      } else {
#pragma unroll
        for (int j = 0; j < (1 << R); ++j) tdst[off0 + (uint32_t)j * gstride] = v[j];
      }
      if (keep) {
        const uint32_t pb = fft_lds_pad(e0);
#pragma unroll
        for (int j = 0; j < (1 << R); ++j) keep[pb + fft_lds_pad((uint32_t)j << p)] = v[j];
      }
    } else {
      const uint32_t pb = fft_lds_pad(e0);
#pragma unroll
      for (int j = 0; j < (1 << R); ++j) sm_out[pb + fft_lds_pad((uint32_t)j << p)] = v[j];
    }
  }
}

template <bool INV>
LMN_D void fft_stage_dispatch(int R, const uint32_t* sm_in, uint32_t* sm_out, uint32_t* col, const uint32_t* scol,
                              uint64_t src_len, uint64_t base, int lo, int hi, int cb, int first_layer, uint32_t H,
                              bool from_global, bool to_global, const TwPtrs& tw, uint32_t scale, uint32_t* keep = nullptr) {
  switch (R) {
    case 1: fft_stage<1, INV>(sm_in, sm_out, col, scol, src_len, base, lo, hi, cb, first_layer, H, from_global, to_global, tw, scale, keep); break;
    case 2: fft_stage<2, INV>(sm_in, sm_out, col, scol, src_len, base, lo, hi, cb, first_layer, H, from_global, to_global, tw, scale, keep); break;
    case 3: fft_stage<3, INV>(sm_in, sm_out, col, scol, src_len, base, lo, hi, cb, first_layer, H, from_global, to_global, tw, scale, keep); break;
    default: fft_stage<4, INV>(sm_in, sm_out, col, scol, src_len, base, lo, hi, cb, first_layer, H, from_global, to_global, tw, scale, keep); break;
  }
}

template <bool INV>
LMN_KERNEL k_fft_staged(uint32_t* data, uint64_t col_stride, const uint32_t* src, uint64_t src_stride,
                        uint64_t src_len, FftStagePlan pl, TwPtrs tw, uint32_t scale, int ncols, int cpb,
                        uint32_t h_off) {
  LMN_DYN_SMEM(uint32_t, sm);
  // XCD-aware tile order: consecutive workgroups are dealt round-robin to the 8 XCDs (observed, used
  // for speed only), so give each XCD a contiguous run of tiles: neighbouring strided tiles share
  // 128-byte lines and then hit the same L2.
  uint32_t tile = blockIdx.x;
  if (pl.xcd_swizzle && (gridDim.x & 7u) == 0u) tile = (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  const uint32_t q = tile & ((1u << (pl.lo - pl.cb)) - 1u);
  const uint32_t Hl = tile >> (pl.lo - pl.cb);
  const uint64_t base = ((uint64_t)Hl << pl.hi) + ((uint64_t)q << pl.cb);
  // h_off != 0: the data is one aligned block of a larger domain (row-block sharding): addresses stay local, the
  // twiddle index carries the block's position
  const uint32_t H = Hl + h_off;
  for (int cc = 0; cc < cpb; ++cc) {
    const int c = blockIdx.y * cpb + cc;
    if (c >= ncols) break;
    uint32_t* col = data + (uint64_t)c * col_stride;
    const uint32_t* scol = src + (uint64_t)c * src_stride;
    for (int k = 0; k < pl.nst; ++k) {
      const int s = INV ? k : pl.nst - 1 - k;
      const bool fg = k == 0, tg = k == pl.nst - 1;
      fft_stage_dispatch<INV>(pl.R[s], sm, sm, col, scol, src_len, base, pl.lo, pl.hi, pl.cb, pl.first[s], H, fg, tg, tw, scale);
      __syncthreads();
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Fused "interpolate then extend" pass (PolyOps::interpolate followed by PolyOps::evaluate on the blown-up domain, as
// every committed column goes through: prover.rs:56-59,179,298).  The last pass of the inverse transform on 2^n points
// (layers [lo, n), strided tile) and the first pass of the forward transform onto 2^(n+1) points touch the SAME
// coefficient positions: the zero-extended coefficient vector makes the forward layer n the identity on both halves,
// and layers [lo, n) of each half h are the tile's own layers with twiddle row H = h.  So one workgroup loads the
// tile once, finishes the interpolation (coefficients go to HBM - the OODS evaluation needs them - and stay in LDS),
// and runs the forward layers twice from LDS, writing both halves of the extended evaluation: one launch and one
// read of the coefficients less per column than two separate passes.
// ---------------------------------------------------------------------------------------------
LMN_KERNEL k_fft_interp_extend(uint32_t* coeffs, uint64_t coeff_stride, const uint32_t* src, uint64_t src_stride,
                               uint32_t* lde, uint64_t lde_stride, FftStagePlan pl, TwPtrs itw, TwPtrs tw, uint32_t scale,
                               int ncols, int cpb) {
  LMN_DYN_SMEM(uint32_t, sm);
  const uint32_t tile_elems = 1u << (pl.hi - pl.lo + pl.cb);
  uint32_t* A = sm;                                   // the coefficient tile (kept for the second half)
  uint32_t* B = sm + fft_lds_pad(tile_elems) + 1u;    // exchange buffer of the stages
  uint32_t tile = blockIdx.x;
  if (pl.xcd_swizzle && (gridDim.x & 7u) == 0u) tile = (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  const uint64_t base = (uint64_t)tile << pl.cb;      // hi = n: the tile spans the whole column in its strided rows
  const uint64_t n_words = 1ull << pl.hi;
  for (int cc = 0; cc < cpb; ++cc) {
    const int c = blockIdx.y * cpb + cc;
    if (c >= ncols) break;
    uint32_t* ccol = coeffs + (uint64_t)c * coeff_stride;
    const uint32_t* scol = src + (uint64_t)c * src_stride;
    uint32_t* lcol = lde + (uint64_t)c * lde_stride;
    // inverse layers [lo, n): stages ascending; the last one scales, stores the coefficients and keeps them in A
    for (int k = 0; k < pl.nst; ++k) {
      const bool fg = k == 0, tg = k == pl.nst - 1;
      fft_stage_dispatch<true>(pl.R[k], B, B, ccol, scol, n_words, base, pl.lo, pl.hi, pl.cb, pl.first[k], 0u, fg, tg, itw, scale,
                               tg ? A : nullptr);
      __syncthreads();
    }
    // forward layers [lo, n) of half h of the 2^(n+1)-point transform: stages descending, first one reads A
    for (uint32_t h = 0; h < 2; ++h) {
      for (int k = 0; k < pl.nst; ++k) {
        const int s = pl.nst - 1 - k;
        const bool tg = k == pl.nst - 1;
        fft_stage_dispatch<false>(pl.R[s], k == 0 ? A : B, B, lcol + h * n_words, lcol, 0, base, pl.lo, pl.hi, pl.cb, pl.first[s], h,
                                  false, tg, tw, 1u);
        __syncthreads();
      }
    }
  }
}

struct FftPass {
  int lo, hi, cb;
};
constexpr int FFT_LOW_BITS = 12;    // contiguous low pass: 2^12 words = 16 KiB LDS
constexpr int FFT_HIGH_BITS = 10;   // strided high passes: 2^10 rows x 16 words = 64 KiB LDS
constexpr int FFT_HIGH_CB = 4;

static int plan_passes(int log_n, FftPass* out) {
  // LMN_FFT_SPLIT (experiment): 0 = strided passes of equal depth (default), 1 = deepest first (10, then the rest),
  // 2 = deepest last
  static const int split = getenv("LMN_FFT_SPLIT") ? atoi(getenv("LMN_FFT_SPLIT")) : 0;
  int n = 0;
  int lo = 0;
  int hi = log_n < FFT_LOW_BITS ? log_n : FFT_LOW_BITS;
  out[n++] = {0, hi, 0};
  lo = hi;
  while (lo < log_n) {
    int rem = log_n - lo;
    int npass = (rem + FFT_HIGH_BITS - 1) / FFT_HIGH_BITS;
    int take = (rem + npass - 1) / npass;
    if (split == 1 && npass > 1) take = FFT_HIGH_BITS;
    if (split == 2 && npass > 1) take = rem - (npass - 1) * FFT_HIGH_BITS;
    // three-pass sizes (2^23 points and more) have strided passes of 5 - 7 layers: 32-word runs (whole 128-byte lines)
    // keep their tiles at 4 - 16 KiB; the single strided pass of the smaller sizes keeps 16-word runs (up to 2^10 rows)
    static const int cb3 = getenv("LMN_FFT_CB3") ? atoi(getenv("LMN_FFT_CB3")) : 5;
    out[n++] = {lo, lo + take, take == 5 ? 5 : (npass > 1 && take <= 7 ? cb3 : FFT_HIGH_CB)};
    lo += take;
  }
  return n;
}

static uint32_t inv_pow2(int log_n) {
  // 2^-log_n mod P = 2^(31 - log_n mod 31)
  int e = (31 - (log_n % 31)) % 31;
  return 1u << e;
}

static void split_stages(FftStagePlan& pl) {
  int rbits = pl.hi - pl.lo;
  int nst = (rbits + 3) / 4;
  pl.nst = nst;
  int f = pl.lo;
  for (int k = 0; k < nst; ++k) {
    int r = (rbits - (f - pl.lo) + (nst - k) - 1) / (nst - k);  // balanced split, each <= 4
    pl.first[k] = f;
    pl.R[k] = r;
    f += r;
  }
}

// one k_fft_staged launch: layers [p.lo, p.hi) of a 2^log_n transform
template <bool INV>
static void launch_staged_pass(uint32_t* data, uint64_t col_stride, const uint32_t* psrc, uint64_t pstride, uint64_t plen,
                               const FftPass& p, int log_n, const TwPtrs& tw, uint32_t scale, int ncols, lmn_stream_t s,
                               uint32_t block_index = 0) {
  const int rbits = p.hi - p.lo;
  const unsigned tiles = 1u << (log_n - rbits - p.cb);
  FftStagePlan pl{};
  pl.lo = p.lo;
  pl.hi = p.hi;
  pl.cb = p.cb;
  split_stages(pl);
  static const int env_xcd = getenv("LMN_FFT_XCD") ? atoi(getenv("LMN_FFT_XCD")) : 1;
  pl.xcd_swizzle = (env_xcd && p.cb > 0) ? 1 : 0;
  uint32_t tile_elems = 1u << (rbits + p.cb);
  size_t smem = (size_t)4 * (tile_elems + (tile_elems >> 5) + 1);
  // several columns per block when there are plenty of tiles: twiddles stay hot in L1/L2
  int cpb = tiles >= 2048 ? 3 : (tiles >= 512 ? 2 : 1);
  static const int env_cpb = getenv("LMN_FFT_CPB") ? atoi(getenv("LMN_FFT_CPB")) : 0;
  static const int env_thr = getenv("LMN_FFT_THREADS") ? atoi(getenv("LMN_FFT_THREADS")) : 0;
  if (env_cpb > 0) cpb = env_cpb;
  if (cpb > ncols) cpb = ncols;
  // the shapes of the prover's committed columns have compile-time-specialised kernels (fft_fixed.hip)
  const bool full = plen >= (1ull << log_n);
  const bool half_top = !INV && plen == (1ull << (log_n - 1)) && p.hi == log_n && p.lo > 0;   // LDE by one bit, top pass
  if (full || half_top) {
    const uint32_t scale_log = scale == 1u ? 0u : (uint32_t)__builtin_ctz(scale);
    if (launch_fft_fixed_pass(INV, data, col_stride, psrc, pstride, p.lo, rbits, p.cb, log_n, tw, scale_log, ncols, cpb,
                              block_index << (log_n - p.hi), pl.xcd_swizzle, !full, s))
      return;
  }
  unsigned gy = (unsigned)((ncols + cpb - 1) / cpb);
  int threads = (int)std::min<uint32_t>(TPB, std::max<uint32_t>(64u, tile_elems >> 4));
  if (env_thr > 0) threads = env_thr;
  LMN_LAUNCH(k_fft_staged<INV>, dim3(tiles, gy), dim3(threads), smem, s, data, col_stride, psrc, pstride, plen, pl, tw,
             scale, ncols, cpb, block_index << (log_n - p.hi));
}

template <bool INV>
static int run_fft(uint32_t* data, uint64_t col_stride, const uint32_t* src, uint64_t src_stride, int log_src,
                    int ncols, int log_n, const TwPtrs& tw, lmn_stream_t s, uint32_t block_index = 0) {
  if (log_n < 1) throw LmnError(-100, "fft: log_n < 1");
  static const bool use_v1 = getenv("LMN_FFT_V1") != nullptr;
  FftPass passes[8];
  int np = plan_passes(log_n, passes);
  for (int k = 0; k < np; ++k) {
    const FftPass& p = INV ? passes[k] : passes[np - 1 - k];
    int rbits = p.hi - p.lo;
    unsigned tiles = 1u << (log_n - rbits - p.cb);
    bool last = INV && k == np - 1;
    uint32_t scale = last ? inv_pow2(log_n) : 1u;
    const uint32_t* psrc = k == 0 ? src : data;
    uint64_t pstride = k == 0 ? src_stride : col_stride;
    uint64_t plen = k == 0 ? (1ull << log_src) : (1ull << log_n);
    if (use_v1) {
      size_t smem = (size_t)4 << (rbits + p.cb);
      LMN_LAUNCH(k_fft_pass<INV>, dim3(tiles, ncols), dim3(TPB), smem, s, data, col_stride, psrc, pstride, plen,
                 p.lo, p.hi, p.cb, tw, scale);
      continue;
    }
    launch_staged_pass<INV>(data, col_stride, psrc, pstride, plen, p, log_n, tw, scale, ncols, s, block_index);
  }
  return np;
}

// interpolate (2^log_n evaluations -> coefficients, kept) + extend onto the 2^(log_n + 1) domain in three launches:
// inverse low pass, the fused strided pass (k_fft_interp_extend), forward low pass.  Applies when both transforms
// have exactly one strided pass of at most FFT_HIGH_BITS layers.
bool fft_interp_extend_supported(int log_n) {
  static const bool off = getenv("LMN_NO_FFT_FUSION") != nullptr;
  return !off && log_n > FFT_LOW_BITS && log_n - FFT_LOW_BITS <= FFT_HIGH_BITS - 1;
}
// the fused strided pass on coefficients that have been through the inverse low pass, then the forward low pass
static void interp_extend_tail(uint32_t* coeffs, uint64_t coeff_stride, uint32_t* lde, uint64_t lde_stride, int ncols, int log_n,
                               const TwPtrs& itw, const TwPtrs& tw_ext, lmn_stream_t s) {
  const FftPass low{0, FFT_LOW_BITS, 0};
  FftStagePlan pl{};
  pl.lo = FFT_LOW_BITS;
  pl.hi = log_n;
  pl.cb = FFT_HIGH_CB;
  split_stages(pl);
  pl.xcd_swizzle = 1;
  const uint32_t tile_elems = 1u << (log_n - FFT_LOW_BITS + FFT_HIGH_CB);
  const size_t smem = (size_t)8 * (tile_elems + (tile_elems >> 5) + 1);
  const unsigned tiles = 1u << (FFT_LOW_BITS - FFT_HIGH_CB);
  int cpb = 1;
  const int threads = (int)std::min<uint32_t>(TPB, std::max<uint32_t>(64u, tile_elems >> 4));
#if !defined(LMN_EMU) && !defined(LMN_BATCH)
  if (smem > 64 * 1024) allow_big_lds((const void*)k_fft_interp_extend, 160 * 1024);
#endif
  if (!launch_interp_extend_fixed(coeffs, coeff_stride, lde, lde_stride, log_n, itw, tw_ext, ncols, s))
    LMN_LAUNCH(k_fft_interp_extend, dim3(tiles, (unsigned)((ncols + cpb - 1) / cpb)), dim3(threads), smem, s, coeffs,
               coeff_stride, coeffs, coeff_stride, lde, lde_stride, pl, itw, tw_ext, inv_pow2(log_n), ncols, cpb);
  launch_staged_pass<false>(lde, lde_stride, lde, lde_stride, 2ull << log_n, low, log_n + 1, tw_ext, 1u, ncols, s);
}

int launch_interp_extend(uint32_t* coeffs, uint64_t coeff_stride, const uint32_t* evals, uint64_t evals_stride, uint32_t* lde,
                         uint64_t lde_stride, int ncols, int log_n, const TwPtrs& itw, const TwPtrs& tw_ext, lmn_stream_t s) {
  if (LMN_ABLATED(2u)) return 3;
  if (!fft_interp_extend_supported(log_n)) throw LmnError(-100, "interp_extend: unsupported size");
  const FftPass low{0, FFT_LOW_BITS, 0};
  launch_staged_pass<true>(coeffs, coeff_stride, evals, evals_stride, 1ull << log_n, low, log_n, itw, 1u, ncols, s);
  interp_extend_tail(coeffs, coeff_stride, lde, lde_stride, ncols, log_n, itw, tw_ext, s);
  return 3;
}

// launch_interp_extend whose evaluations are still the table's AoS ROWS: the transpose happens inside the inverse low pass
// (fft_fixed.hip k_fft_rows_fx).  false: this size / build has no such pass - transpose, then launch_interp_extend.
bool launch_interp_extend_rows(uint32_t* coeffs, uint64_t coeff_stride, const uint32_t* rows, uint64_t n_rows, const PadRow& pad,
                               uint32_t* bad_flag, uint32_t bad_value, uint32_t* lde, uint64_t lde_stride, int ncols, int log_n,
                               const TwPtrs& itw, const TwPtrs& tw_ext, lmn_stream_t s) {
  static_assert(FFT_LOW_BITS == 12, "k_fft_rows_fx is the 12-layer contiguous pass");
  if (!fft_interp_extend_supported(log_n)) return false;
  if (LMN_ABLATED(2u)) return true;
  if (!launch_fft_rows_fixed(coeffs, coeff_stride, rows, n_rows, ncols, log_n, pad, bad_flag, bad_value, itw, s)) return false;
  interp_extend_tail(coeffs, coeff_stride, lde, lde_stride, ncols, log_n, itw, tw_ext, s);
  return true;
}

int launch_ifft(uint32_t* dst, uint64_t dst_stride, const uint32_t* src, uint64_t src_stride, int ncols, int log_n,
                const TwPtrs& itw, lmn_stream_t s) {
  if (LMN_ABLATED(2u)) return 1;
  return run_fft<true>(dst, dst_stride, src, src_stride, log_n, ncols, log_n, itw, s);
}
int launch_fft(uint32_t* dst, uint64_t dst_stride, const uint32_t* src, uint64_t src_stride, int log_src, int ncols,
               int log_n, const TwPtrs& tw, lmn_stream_t s) {
  if (LMN_ABLATED(2u)) return 1;
  return run_fft<false>(dst, dst_stride, src, src_stride, log_src, ncols, log_n, tw, s);
}

// Row-block LDE (single-commitment sharding): block `b` of 2^g equal blocks of the forward transform
// onto a 2^n domain.  The top g layers pair indices that differ in the block bits only and their twiddle index
// (idx >> (i+1)) depends on the block bits only, so element t of block b after those layers is a 2^g-point
// transform across the blocks at fixed t - computed here from the (zero-extended) coefficients; the remaining
// layers then run inside the block (k_fft_staged with the block's twiddle offset).
template <int G>
LMN_KERNEL k_fft_top_block(uint32_t* __restrict__ dst, uint64_t dst_stride, const uint32_t* __restrict__ src,
                           uint64_t src_stride, uint64_t src_len, int log_n, uint32_t block, TwPtrs tw) {
  const uint64_t S = 1ull << (log_n - G);
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= S) return;
  const uint32_t* scol = src + (uint64_t)blockIdx.y * src_stride;
  uint32_t v[1 << G];
#pragma unroll
  for (int c = 0; c < (1 << G); ++c) {
    const uint64_t idx = (uint64_t)c * S + t;
    v[c] = idx < src_len ? scol[idx] : 0u;
  }
#pragma unroll
  for (int k = G - 1; k >= 0; --k) {
    const uint32_t* __restrict__ tl = tw.l[log_n - G + k];
#pragma unroll
    for (int c = 0; c < (1 << G); ++c) {
      if (c & (1 << k)) continue;
      const uint32_t w = tl[c >> (k + 1)];
      const uint32_t a = v[c], x = m_mul(v[c | (1 << k)], w);
      v[c] = m_add(a, x);
      v[c | (1 << k)] = m_sub(a, x);
    }
  }
  uint32_t out = v[0];
#pragma unroll
  for (int c = 1; c < (1 << G); ++c) out = block == (uint32_t)c ? v[c] : out;
  dst[(uint64_t)blockIdx.y * dst_stride + t] = out;
}

int launch_fft_block(uint32_t* dst, uint64_t dst_stride, const uint32_t* src, uint64_t src_stride, int log_src, int ncols,
                     int log_n, int log_blocks, uint32_t block, const TwPtrs& tw, lmn_stream_t s) {
  if (log_blocks < 1 || log_blocks > 3 || log_n - log_blocks < 1 || block >= (1u << log_blocks))
    throw LmnError(-100, "fft_block: bad arguments");
  const int lb = log_n - log_blocks;
  dim3 g(cdiv(1ull << lb, TPB), ncols), b(TPB);
  const uint64_t slen = 1ull << log_src;
  switch (log_blocks) {
    case 1: LMN_LAUNCH(k_fft_top_block<1>, g, b, 0, s, dst, dst_stride, src, src_stride, slen, log_n, block, tw); break;
    case 2: LMN_LAUNCH(k_fft_top_block<2>, g, b, 0, s, dst, dst_stride, src, src_stride, slen, log_n, block, tw); break;
    default: LMN_LAUNCH(k_fft_top_block<3>, g, b, 0, s, dst, dst_stride, src, src_stride, slen, log_n, block, tw); break;
  }
  return 1 + run_fft<false>(dst, dst_stride, dst, dst_stride, lb, ncols, lb, tw, s, block);
}

// one global-memory layer per launch: the obviously-correct reference used by the self-test
LMN_KERNEL k_fft_layer_simple(uint32_t* __restrict__ data, uint64_t col_stride, int log_n, int i,
                              const uint32_t* __restrict__ t, int inverse, uint32_t scale) {
  uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint64_t half = 1ull << (log_n - 1);
  if (b >= half) return;
  uint32_t* col = data + (uint64_t)blockIdx.y * col_stride;
  uint64_t h = b >> i, l = b & ((1ull << i) - 1);
  uint64_t i0 = (h << (i + 1)) + l, i1 = i0 + (1ull << i);
  uint32_t w = t[h];
  uint32_t v0 = col[i0], v1 = col[i1];
  if (inverse) {
    uint32_t a = m_add(v0, v1), d = m_mul(m_sub(v0, v1), w);
    if (scale != 1u) {
      a = m_mul(a, scale);
      d = m_mul(d, scale);
    }
    col[i0] = a;
    col[i1] = d;
  } else {
    uint32_t x = m_mul(v1, w);
    col[i0] = m_add(v0, x);
    col[i1] = m_sub(v0, x);
  }
}

LMN_KERNEL k_pack_blocks(uint32_t* __restrict__ cols, uint64_t col_stride, uint32_t* __restrict__ packed, uint32_t block_rows,
                         int ncols, int nsel, PackSel sel, int unpack) {
  const uint32_t h = blockIdx.y % (uint32_t)nsel, c = (blockIdx.y / (uint32_t)nsel) % (uint32_t)ncols;
  const uint32_t s = blockIdx.y / (uint32_t)(nsel * ncols);
  uint32_t* cp = cols + (uint64_t)c * col_stride + (uint64_t)sel.blk[s][h] * block_rows;
  uint32_t* pp = packed + (((uint64_t)s * ncols + c) * nsel + h) * block_rows;
  if (unpack) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < block_rows; i += gridDim.x * blockDim.x) cp[i] = pp[i];
  } else {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < block_rows; i += gridDim.x * blockDim.x) pp[i] = cp[i];
  }
}
void launch_pack_blocks(const uint32_t* src, uint64_t src_stride, uint32_t* dst, uint32_t block_rows, int ncols, int nsel,
                        int world, const PackSel& sel, lmn_stream_t s) {
  if (ncols <= 0) return;
  LMN_LAUNCH(k_pack_blocks, dim3(std::min<unsigned>(cdiv(block_rows, TPB), 64u), (unsigned)(world * ncols * nsel)), dim3(TPB),
             0, s, const_cast<uint32_t*>(src), src_stride, dst, block_rows, ncols, nsel, sel, 0);
}
void launch_unpack_blocks(const uint32_t* packed, uint32_t* cols, uint64_t col_stride, uint32_t block_rows, int ncols, int world,
                          lmn_stream_t s) {
  if (ncols <= 0) return;
  PackSel sel{};
  for (int r = 0; r < world; ++r) sel.blk[r][0] = (uint32_t)r;
  LMN_LAUNCH(k_pack_blocks, dim3(std::min<unsigned>(cdiv(block_rows, TPB), 64u), (unsigned)(world * ncols)), dim3(TPB), 0, s,
             cols, col_stride, const_cast<uint32_t*>(packed), block_rows, ncols, 1, sel, 1);
}

void launch_fft_simple(uint32_t* data, uint64_t col_stride, int ncols, int log_n, const TwPtrs& tw, bool inverse,
                       lmn_stream_t s) {
  uint64_t half = 1ull << (log_n - 1);
  for (int k = 0; k < log_n; ++k) {
    int i = inverse ? k : log_n - 1 - k;
    uint32_t scale = (inverse && k == log_n - 1) ? inv_pow2(log_n) : 1u;
    LMN_LAUNCH(k_fft_layer_simple, dim3(cdiv(half, TPB), ncols), dim3(TPB), 0, s, data, col_stride, log_n, i,
               tw.l[i], inverse ? 1 : 0, scale);
  }
}

LMN_KERNEL k_extend(const uint32_t* __restrict__ src, uint64_t src_stride, uint64_t src_len,
                    uint32_t* __restrict__ dst, uint64_t dst_stride, uint64_t dst_len) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= dst_len) return;
  uint32_t v = i < src_len ? src[(uint64_t)blockIdx.y * src_stride + i] : 0u;
  dst[(uint64_t)blockIdx.y * dst_stride + i] = v;
}

void launch_extend(const uint32_t* src, uint64_t src_stride, int log_src, uint32_t* dst, uint64_t dst_stride,
                   int log_dst, int ncols, lmn_stream_t s) {
  uint64_t dl = 1ull << log_dst;
  LMN_LAUNCH(k_extend, dim3(cdiv(dl, TPB), ncols), dim3(TPB), 0, s, src, src_stride, 1ull << log_src, dst,
             dst_stride, dl);
}

// =============================================================================================
// a11  twiddle tables: the points of a half coset by double-and-add from the step's doublings (<= 26 group additions per
// entry) instead of a serial walk on the host - 2^26 entries and their inverses in milliseconds
// =============================================================================================
LMN_KERNEL k_twiddles(int bits, TwGen g, int coord, uint32_t* __restrict__ tw, uint32_t* __restrict__ itw,
                      uint32_t* __restrict__ tw2, uint32_t* __restrict__ itw2) {
  const uint32_t h = blockIdx.x * blockDim.x + threadIdx.x;
  if (h >= (1u << bits)) return;
  const uint32_t j = bits ? (__brev(h) >> (32 - bits)) : 0u;
  uint32_t x = g.ix, y = g.iy;
  for (int k = 0; k < bits; ++k) {
    if ((j >> k) & 1u) {
      const uint32_t nx = m_sub(m_mul(x, g.sx[k]), m_mul(y, g.sy[k]));
      y = m_add(m_mul(x, g.sy[k]), m_mul(y, g.sx[k]));
      x = nx;
    }
  }
  const uint32_t v = coord ? x : y, vi = m_inv(v);
  tw[h] = v;
  itw[h] = vi;
  tw2[h] = 2u * v;
  itw2[h] = 2u * vi;
}

void launch_twiddles(int bits, const TwGen& g, int coord, uint32_t* tw, uint32_t* itw, uint32_t* tw2, uint32_t* itw2,
                     lmn_stream_t s) {
  if (bits < 0 || bits > 29) throw LmnError(-100, "twiddles: bad size");
  LMN_LAUNCH(k_twiddles, dim3(cdiv(1ull << bits, TPB)), dim3(TPB), 0, s, bits, g, coord, tw, itw, tw2, itw2);
}

}  // namespace lmn

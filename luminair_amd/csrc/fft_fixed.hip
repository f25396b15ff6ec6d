// Fixed-shape circle-FFT passes for gfx950 (stwo PolyOps::interpolate / evaluate behind
// /root/reference/crates/prover/src/prover.rs:56-59,179,298 and the composition commit behind :312).
//
// The generic k_fft_staged (kernels_fft.hip) takes its tile geometry at run time; every address, LDS index and twiddle index is
// then computed with vector instructions and the emitted code spends ~19 VALU instructions per butterfly for 12 of
// arithmetic.  The shapes the prover's committed columns actually have are few - the contiguous 12-layer low pass and the
// strided 16-word-run passes of 5..10 layers - so they are instantiated here with everything but the tile's position
// known at compile time:
//   * LDS and global offsets of a lane's 2^R points are immediates / scalar registers (one base per lane);
//   * the twiddles of a tile's TOP stage depend on the tile only, so they are scalar loads and SGPR operands;
//     the other stages load theirs as aligned runs (1, 2, 4, 8 words) with wide loads;
//   * the M31 product uses DOUBLED twiddles (TwPtrs::d): x * 2w = 2^32 * floor(x w / 2^31) + 2 * (x w mod 2^31), so the
//     partially reduced product is (lo >> 1) + hi - two plain instructions instead of funnel shift + mask + add -
//     and the butterfly is 11 instructions (mad, shr, add, sub, min | add, sub, min | sub, add, min);
//   * the 2^-n scaling of the inverse transform is a 31-bit rotation (funnel shift + mask) instead of a multiplication.
// All values stay canonical (< P) between layers, exactly as in k_fft_staged: the two kernels are interchangeable bit for
// bit (op_fft_selftest compares both with the one-layer-per-launch kernels).
// Instruction classes are issued in s_setprio phases as in k_fft_staged (issue_phases.h).
#include "fft_fixed.h"

#include <algorithm>

#include "issue_phases.h"
#include "launch_util.h"

namespace lmn {

LMN_HD constexpr uint32_t fx_pad(uint32_t e) { return e + (e >> 5); }

#if !defined(LMN_EMU) && !defined(LMN_BATCH)
#define LMN_BOUNDS(n) __launch_bounds__(n)
#else
#define LMN_BOUNDS(n)
#endif

// A tile's words in global memory, addressed as uniform base + scalar word offset + per-lane word offset.  On the device
// this is a raw buffer resource: the scalar part travels in an SGPR (soffset) and the lane part in one VGPR, so the 2^R
// strided accesses of a register stage need no per-access address arithmetic (plain pointers made the compiler add the
// scalar offset to a 64-bit per-lane address with two vector instructions per access).
struct GTile {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(LMN_EMU)
  __amdgpu_buffer_rsrc_t r;
#else
  uint32_t* p;
#endif
};
LMN_D GTile gtile(const uint32_t* p) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(LMN_EMU)
  // raw buffer, stride 0, no bounds (the prover's columns are far below 4 GiB), gfx9 dword-3 flags: 32-bit data format
  return GTile{__builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(p), (short)0, (int)0xffffffff, 0x00020000)};
#else
  return GTile{const_cast<uint32_t*>(p)};
#endif
}
LMN_D uint32_t gtile_load(const GTile& t, uint32_t lane_word, uint32_t uniform_word) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(LMN_EMU)
  return __builtin_amdgcn_raw_buffer_load_b32(t.r, (int)(lane_word << 2), (int)(uniform_word << 2), 0);
#else
  return t.p[(uint64_t)uniform_word + lane_word];
#endif
}
LMN_D void gtile_store(const GTile& t, uint32_t lane_word, uint32_t uniform_word, uint32_t v) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(LMN_EMU)
  __builtin_amdgcn_raw_buffer_store_b32(v, t.r, (int)(lane_word << 2), (int)(uniform_word << 2), 0);
#else
  t.p[(uint64_t)uniform_word + lane_word] = v;
#endif
}

// N consecutive words from an N-word-aligned position of a 256-byte-aligned table
template <int N>
LMN_D void fx_load_run(const uint32_t* __restrict__ p, uint32_t* out) {
#if !defined(LMN_EMU)
  if constexpr (N >= 4) {
#pragma unroll
    for (int k = 0; k < N / 4; ++k) {
      const uint4 x = reinterpret_cast<const uint4*>(p)[k];
      out[4 * k] = x.x;
      out[4 * k + 1] = x.y;
      out[4 * k + 2] = x.z;
      out[4 * k + 3] = x.w;
    }
  } else if constexpr (N == 2) {
    const uint2 x = *reinterpret_cast<const uint2*>(p);
    out[0] = x.x;
    out[1] = x.y;
  } else {
    out[0] = p[0];
  }
#else
  for (int k = 0; k < N; ++k) out[k] = p[k];
#endif
}

// out[k] = x[k] * w[k] (canonical) for N independent products, w2[k] = 2 w[k].  Leaves the wave in the first-port phase.
template <int N>
LMN_D void m_mul2_phased(const uint32_t (&x)[N], const uint32_t (&w2)[N], uint32_t (&out)[N]) {
  uint64_t pr[N];
  uint32_t s[N], s2[N];
  LMN_PHASE_PORT0();
#pragma unroll
  for (int k = 0; k < N; ++k) pr[k] = (uint64_t)x[k] * (uint64_t)w2[k];
  LMN_PHASE_ANY();
#pragma unroll
  for (int k = 0; k < N; ++k) {
    s[k] = ((uint32_t)pr[k] >> 1) + (uint32_t)(pr[k] >> 32);   // x w mod 2^31 + floor(x w / 2^31) <= 2P - 2
    s2[k] = s[k] - P31;
  }
  LMN_PHASE_PORT0();
#pragma unroll
  for (int k = 0; k < N; ++k) out[k] = s[k] < s2[k] ? s[k] : s2[k];
}

template <int R>
constexpr int fx_tw_at(int r) { return (1 << R) - (1 << (R - r)); }  // first entry of layer r in a stage's twiddle array

// R butterfly layers on the 2^R points a lane holds; t2 = the stage's 2^R - 1 doubled twiddles (layer r: 2^(R-1-r) of them)
// SKIP_TOP (forward only): the stage's top layer is the identity (its inputs' upper half was zero and has been filled with
// a copy of the lower half)
template <int R, bool INV, bool SKIP_TOP = false>
LMN_D void fx_butterflies(uint32_t (&v)[1 << R], const uint32_t (&t2)[(1 << R) - 1]) {
  constexpr int NB = 1 << (R - 1);
  static_assert(!(SKIP_TOP && INV), "only the forward transform is zero-extended");
#pragma unroll
  for (int rr = SKIP_TOP ? 1 : 0; rr < R; ++rr) {
    const int r = INV ? rr : R - 1 - rr;
    uint32_t w[NB], a[NB], b[NB], x[NB], u[NB], u2[NB], d[NB], d2[NB];
#pragma unroll
    for (int k = 0; k < NB; ++k) {
      const int j = ((k >> r) << (r + 1)) | (k & ((1 << r) - 1));
      w[k] = t2[fx_tw_at<R>(r) + (k >> r)];
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
        a[k] = u[k] < u2[k] ? u[k] : u2[k];
        d[k] = d[k] < d2[k] ? d[k] : d2[k];
      }
      m_mul2_phased<NB>(d, w, x);
#pragma unroll
      for (int k = 0; k < NB; ++k) {
        const int j = ((k >> r) << (r + 1)) | (k & ((1 << r) - 1));
        v[j] = a[k];
        v[j | (1 << r)] = x[k];
      }
    } else {
      m_mul2_phased<NB>(b, w, x);
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
        v[j] = u[k] < u2[k] ? u[k] : u2[k];
        v[j | (1 << r)] = d[k] < d2[k] ? d[k] : d2[k];
      }
    }
  }
  LMN_PHASE_ANY();
}

// The doubled twiddles of a register stage, layer r = 0 .. R-1: the aligned run of 2^(R-1-r) entries starting at
// (H << (HB - r - 1)) + (mhigh << (R-1-r)) of the table of layer `layer0 + r` (HB = layers from the stage's first up to
// the top of the tile).
template <int R, int r, int HB>
LMN_D void fx_load_twiddles(uint32_t (&t2)[(1 << R) - 1], const uint32_t* const (&twd)[MAX_LOG], int layer0, uint32_t H,
                            uint32_t mhigh) {
  const uint32_t* __restrict__ t = twd[layer0 + r];
  const uint32_t hb = (H << (HB - r - 1)) + (mhigh << (R - 1 - r));
  LMN_ASSUME(hb < (1u << 28));
  fx_load_run<(1 << (R - 1 - r))>(t + hb, &t2[fx_tw_at<R>(r)]);
  if constexpr (r + 1 < R) fx_load_twiddles<R, r + 1, HB>(t2, twd, layer0, H, mhigh);
}

// x * 2^e mod P for canonical x, 0 < e < 31: rotation of the 31-bit value
LMN_D uint32_t m_rot(uint32_t x, uint32_t e) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(LMN_EMU)
  return __builtin_amdgcn_alignbit(x, x + x, 32u - e) & P31;
#else
  return (uint32_t)((((uint64_t)x << e) | ((uint64_t)x >> (31u - e))) & P31);
#endif
}

// Tile shape: 2^RBITS rows x 2^CB contiguous words; the RBITS layers are split into stages of at most 4 (balanced, as
// split_stages in kernels_fft.hip).  LO0: the pass starts at layer 0 of a contiguous tile (CB = 0).
template <int RBITS_, int CB_, bool LO0_>
struct FxShape {
  static constexpr int RBITS = RBITS_, CB = CB_, TB = RBITS_ + CB_;
  static constexpr bool LO0 = LO0_;
  static constexpr int NT = 1 << (TB - 4);            // one lane per 16 points
  static constexpr int NST = (RBITS + 3) / 4;
  static constexpr int R(int k) {
    int f = 0, r = 0;
    for (int i = 0; i <= k; ++i) {
      r = (RBITS - f + (NST - i) - 1) / (NST - i);
      f += r;
    }
    return r;
  }
  static constexpr int F(int k) {   // first row bit of stage k
    int f = 0;
    for (int i = 0; i < k; ++i) f += R(i);
    return f;
  }
  static constexpr uint32_t LDS_WORDS = (1u << TB) + ((1u << TB) >> 5) + 1u;
};

// One register stage: R layers starting at bit P of the tile index.  FG: points come from global memory (tsrc),
// otherwise from sm_in; TG: results go to global memory (tdst) - scaled by 2^scale_log when INV - and, with KEEP, also
// into the LDS tile `keep`; otherwise to sm_out.  tsrc / tdst point at the tile's first word; H = tile row of the
// twiddle index (block offset included).
// ZX (forward, the tile's top stage, from global memory): the source holds only the lower half of the rows - the
// coefficients of a polynomial of half the domain's size (PolyOps::evaluate onto the blown-up domain): the upper half
// reads as zero, the top layer's butterflies are copies.
// tid: the lane's index inside the NT lanes that work on this tile (threadIdx.x, or its low bits when a block runs several
// tiles side by side: k_fft_rows_fx)
template <int R, bool INV, int P, class S, bool FG, bool TG, bool KEEP, bool ZX = false>
LMN_D void fx_stage(const uint32_t* sm_in, uint32_t* sm_out, uint32_t* tdst, const uint32_t* tsrc,
                    int lo, uint32_t H, const uint32_t* const (&twd)[MAX_LOG], uint32_t scale_log, uint32_t* keep, uint32_t tid) {
  constexpr int TB = S::TB, CB = S::CB, RBITS = S::RBITS;
  constexpr bool LO0 = S::LO0;
  constexpr uint32_t NG = 1u << (TB - R);
  constexpr uint32_t NT = (uint32_t)S::NT;
  constexpr int ITER = (int)(NG / NT);
  constexpr int L0 = P - CB;                  // the stage's first layer, relative to lo
  constexpr uint32_t cmask = (1u << CB) - 1u;
  static_assert(NG >= NT && P >= CB && P + R <= TB, "stage outside the tile");
#pragma unroll
  for (int it = 0; it < ITER; ++it) {
    const uint32_t g = tid + (uint32_t)it * NT;
    const uint32_t e0 = ((g >> P) << (P + R)) | (g & ((1u << P) - 1u));
    const uint32_t mhigh = e0 >> (P + R);     // row bits above the stage (0 in the tile's top stage: uniform twiddles)
    uint32_t t2[(1 << R) - 1];
    fx_load_twiddles<R, 0, RBITS - L0>(t2, twd, (LO0 ? 0 : lo) + L0, H, mhigh);
    uint32_t v[1 << R];
    const uint32_t off0 = LO0 ? e0 : (((e0 >> CB) << lo) + (e0 & cmask));
    LMN_ASSUME(off0 < 0x10000000u);
    const int gshift = LO0 ? P : (P - CB + lo);   // global stride of the stage's points = 2^gshift words
    if constexpr (FG) {
      if constexpr (LO0 && P == 0 && R >= 2) {
        fx_load_run<(1 << R)>(tsrc + e0, v);
      } else if constexpr (ZX) {
        static_assert(!INV && P + R == TB, "zero extension: forward transform, top stage");
        const GTile gt = gtile(tsrc);
#pragma unroll
        for (int j = 0; j < (1 << (R - 1)); ++j) {
          v[j] = gtile_load(gt, off0, (uint32_t)j << gshift);
          v[j + (1 << (R - 1))] = v[j];
        }
      } else {
        const GTile gt = gtile(tsrc);
#pragma unroll
        for (int j = 0; j < (1 << R); ++j) v[j] = gtile_load(gt, off0, (uint32_t)j << gshift);
      }
    } else {
      const uint32_t pb = fx_pad(e0);
#pragma unroll
      for (int j = 0; j < (1 << R); ++j) v[j] = sm_in[pb + fx_pad((uint32_t)j << P)];
    }
    fx_butterflies<R, INV, ZX>(v, t2);
    if constexpr (TG) {
      if constexpr (INV) {
        if (scale_log != 0u) {
#pragma unroll
          for (int j = 0; j < (1 << R); ++j) v[j] = m_rot(v[j], scale_log);
        }
      }
      if constexpr (LO0 && P == 0 && R >= 2) {
#if !defined(LMN_EMU)
        uint4* q = reinterpret_cast<uint4*>(tdst + e0);
#pragma unroll
        for (int k = 0; k < (1 << R) / 4; ++k) q[k] = make_uint4(v[4 * k], v[4 * k + 1], v[4 * k + 2], v[4 * k + 3]);
#else
        for (int j = 0; j < (1 << R); ++j) tdst[e0 + j] = v[j];
#endif
      } else {
        const GTile gt = gtile(tdst);
#pragma unroll
        for (int j = 0; j < (1 << R); ++j) gtile_store(gt, off0, (uint32_t)j << gshift, v[j]);
      }
      if constexpr (KEEP) {
        const uint32_t pb = fx_pad(e0);
#pragma unroll
        for (int j = 0; j < (1 << R); ++j) keep[pb + fx_pad((uint32_t)j << P)] = v[j];
      }
    } else {
      const uint32_t pb = fx_pad(e0);
#pragma unroll
      for (int j = 0; j < (1 << R); ++j) sm_out[pb + fx_pad((uint32_t)j << P)] = v[j];
    }
  }
}

// The stages of one tile in execution order (inverse: ascending layers; forward: descending), a barrier after each.
// FG0: the first stage reads global memory (else sm_first); TGL: the last stage writes global memory.
// tid / active: see fx_stage; a lane group without a tile (active = false, block-group-uniform) only keeps the barriers.
template <class S, bool INV, int STEP, bool FG0, bool TGL, bool KEEPL, bool ZX0 = false>
LMN_D void fx_steps(const uint32_t* sm_first, uint32_t* sm, uint32_t* tdst, const uint32_t* tsrc, int lo, uint32_t H,
                    const uint32_t* const (&twd)[MAX_LOG], uint32_t scale_log, uint32_t* keep, uint32_t tid, bool active = true) {
  constexpr int k = INV ? STEP : S::NST - 1 - STEP;
  constexpr bool fg = FG0 && STEP == 0;
  constexpr bool tg = TGL && STEP == S::NST - 1;
  if (active)
    fx_stage<S::R(k), INV, S::CB + S::F(k), S, fg, tg, (tg && KEEPL), (ZX0 && STEP == 0)>(STEP == 0 ? sm_first : sm, sm, tdst, tsrc,
                                                                                           lo, H, twd, scale_log, keep, tid);
  __syncthreads();
  if constexpr (STEP + 1 < S::NST)
    fx_steps<S, INV, STEP + 1, FG0, TGL, KEEPL, ZX0>(sm_first, sm, tdst, tsrc, lo, H, twd, scale_log, keep, tid, active);
}

struct TwD {   // kernel argument: the doubled tables only
  const uint32_t* l[MAX_LOG];
};
static TwD doubled(const TwPtrs& tw) {
  TwD t{};
  for (int i = 0; i < MAX_LOG; ++i) t.l[i] = tw.d[i];
  return t;
}

template <bool INV, int RBITS, int CB, bool LO0, bool ZX = false>
LMN_KERNEL LMN_BOUNDS((FxShape<RBITS, CB, LO0>::NT))
k_fft_fx(uint32_t* data, uint64_t col_stride, const uint32_t* src, uint64_t src_stride, int lo_arg, TwD tw, uint32_t scale_log,
         int ncols, int cpb, uint32_t h_off, int xcd_swizzle) {
  using S = FxShape<RBITS, CB, LO0>;
  LMN_DYN_SMEM(uint32_t, sm);
  const int lo = LO0 ? 0 : lo_arg;
  // XCD-aware tile order for the strided passes (as k_fft_staged): each XCD gets a contiguous run of tiles
  uint32_t tile = blockIdx.x;
  if (xcd_swizzle && (gridDim.x & 7u) == 0u) tile = (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  const uint32_t q = tile & ((1u << (lo - CB)) - 1u);
  const uint32_t Hl = tile >> (lo - CB);
  const uint64_t base = ((uint64_t)Hl << (lo + RBITS)) + ((uint64_t)q << CB);
  const uint32_t H = Hl + h_off;
  for (int cc = 0; cc < cpb; ++cc) {
    const int c = (int)blockIdx.y * cpb + cc;
    if (c >= ncols) break;
    uint32_t* col = data + (uint64_t)c * col_stride + base;
    const uint32_t* scol = src + (uint64_t)c * src_stride + base;
    fx_steps<S, INV, 0, true, true, false, ZX>(sm, sm, col, scol, lo, H, tw.l, scale_log, nullptr, threadIdx.x);
  }
}

// k_fft_interp_extend (kernels_fft.hip) with a fixed tile: layers [12, n) of the inverse transform on 2^n points, then the
// same layers of both halves of the forward transform onto 2^(n+1) points from the coefficient tile kept in LDS.
template <int RBITS>
LMN_KERNEL LMN_BOUNDS((FxShape<RBITS, 4, false>::NT))
k_fft_interp_extend_fx(uint32_t* coeffs, uint64_t coeff_stride, uint32_t* lde, uint64_t lde_stride, TwD itw, TwD tw,
                       uint32_t scale_log) {
  using S = FxShape<RBITS, 4, false>;
  constexpr int LO = 12;
  LMN_DYN_SMEM(uint32_t, sm);
  uint32_t* A = sm;                       // the coefficient tile, kept for both halves
  uint32_t* B = sm + S::LDS_WORDS;        // exchange buffer of the stages
  uint32_t tile = blockIdx.x;
  if ((gridDim.x & 7u) == 0u) tile = (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  const uint64_t base = (uint64_t)tile << 4;
  const uint64_t n_words = 1ull << (LO + RBITS);
  uint32_t* ccol = coeffs + (uint64_t)blockIdx.y * coeff_stride + base;
  uint32_t* lcol = lde + (uint64_t)blockIdx.y * lde_stride + base;
  fx_steps<S, true, 0, true, true, true>(B, B, ccol, ccol, LO, 0u, itw.l, scale_log, A, threadIdx.x);
  for (uint32_t h = 0; h < 2; ++h)
    fx_steps<S, false, 0, false, true, false>(A, B, lcol + h * n_words, nullptr, LO, h, tw.l, 0u, nullptr, threadIdx.x);
}

// =============================================================================================
// a3 + a4 fused: the AoS -> SoA transpose of `write_trace` (add/witness.rs:33-108, air/src/utils.rs:59-64) inside the first
// pass of the interpolation.  A workgroup owns rows [tile * 4096, +4096) of up to ROWS_FX_COLS consecutive columns: it reads
// its share of the table's rows (coalesced: ROWS_FX_COLS consecutive words of every row), pads beyond the table's last row,
// rejects non-canonical words, parks the values column-major in LDS - where k_transpose_pad would have written them to
// HBM and the first inverse pass read them back (2 x 4 bytes per cell, 120 MB of a 2^20-row Add proof's 3.6 GB) - and
// runs the contiguous tile's 12 inverse layers on them in place, four columns side by side (one FxShape<12, 0, true> lane
// group of 256 each).  Output: what k_fft_fx<true, 12, 0, true> leaves in `coeffs`.
// =============================================================================================
// COLS columns per workgroup (COLS x 4225 words of LDS: 135 KB for 8, 68 KB - two workgroups per CU, one loading while
// the other transforms - for 4), COLS / 2 tiles transformed side by side.  The column groups of a row tile read the
// same rows: the (tile, group) order puts them on the same XCD back to back (workgroup b runs on XCD b mod 8), so that the
// second group's rows come out of that XCD's L2.
template <int COLS>
LMN_KERNEL LMN_BOUNDS((COLS / 2) * 256)
k_fft_rows_fx(uint32_t* coeffs, uint64_t col_stride, const uint32_t* __restrict__ rows, uint64_t n_rows, int ncols, PadRow pad,
              uint32_t* __restrict__ bad_flag, uint32_t bad_value, TwD itw, uint32_t n_groups) {
  using S = FxShape<12, 0, true>;
  constexpr uint32_t NT = (uint32_t)S::NT, TILE = 1u << 12, GROUPS = COLS / 2;
  static_assert(NT == 256, "one lane group per tile");
  LMN_DYN_SMEM(uint32_t, sm);
  // workgroup b: XCD x = b mod 8, s = b / 8 its position in that XCD's queue -> tile (s / n_groups) * 8 + x, group s mod n_groups
  uint32_t tile = blockIdx.x / n_groups, ygrp = blockIdx.x % n_groups;
  if ((gridDim.x & 7u) == 0u && ((gridDim.x / n_groups) & 7u) == 0u) {
    const uint32_t x = blockIdx.x & 7u, sq = blockIdx.x >> 3;
    tile = (sq / n_groups) * 8u + x;
    ygrp = sq % n_groups;
  }
  const int c0 = (int)ygrp * COLS;
  const int ncb = ncols - c0 < COLS ? ncols - c0 : COLS;
  const uint64_t row0 = (uint64_t)tile << 12;
  // ---- load: lane -> (row, column of the group), the column index fastest: a wave reads 64 / COLS rows x COLS consecutive words
  {
    constexpr int BATCH = 8;
    constexpr uint32_t CELLS = TILE * COLS, STEP = GROUPS * NT;
    for (uint32_t k0 = threadIdx.x; k0 < CELLS; k0 += BATCH * STEP) {
      uint32_t v[BATCH];
#pragma unroll
      for (int j = 0; j < BATCH; ++j) {
        const uint32_t k = k0 + (uint32_t)j * STEP;
        const uint32_t r = k / COLS, c = k % COLS;
        const uint64_t gr = row0 + r;
        v[j] = 0u;
        if ((int)c < ncb) v[j] = gr < n_rows ? rows[gr * (uint64_t)ncols + (uint32_t)c0 + c] : pad.v[(uint32_t)c0 + c];
      }
#pragma unroll
      for (int j = 0; j < BATCH; ++j) {
        const uint32_t k = k0 + (uint32_t)j * STEP;
        const uint32_t r = k / COLS, c = k % COLS;
        if (v[j] >= P31) *bad_flag = bad_value;   // the boundary takes raw u32 words: reject non-canonical M31 values
        if ((int)c < ncb) sm[c * S::LDS_WORDS + fx_pad(r)] = v[j];
      }
    }
  }
  __syncthreads();
  // ---- 12 inverse layers per column, in place in its LDS tile; the last stage writes the column's tile to `coeffs`
#if defined(__HIP_DEVICE_COMPILE__) && !defined(LMN_EMU)
  const uint32_t grp = __builtin_amdgcn_readfirstlane(threadIdx.x / NT);   // wave-uniform: the column's base stays in SGPRs
#else
  const uint32_t grp = threadIdx.x / NT;
#endif
  const uint32_t tid = threadIdx.x % NT;
  for (int round = 0; round * (int)GROUPS < COLS; ++round) {
    const int cl = round * (int)GROUPS + (int)grp;
    const bool active = cl < ncb;
    uint32_t* t = sm + (uint32_t)(active ? cl : 0) * S::LDS_WORDS;
    uint32_t* col = coeffs + (uint64_t)(c0 + (active ? cl : 0)) * col_stride + row0;
    fx_steps<S, true, 0, false, true, false>(t, t, col, nullptr, 0, tile, itw.l, 0u, nullptr, tid, active);
  }
}

template <int COLS>
static void launch_rows_fx(uint32_t* coeffs, uint64_t col_stride, const uint32_t* rows, uint64_t n_rows, int ncols, int log_n,
                           const PadRow& pad, uint32_t* bad_flag, uint32_t bad_value, const TwPtrs& itw, lmn_stream_t s) {
  using S = FxShape<12, 0, true>;
  const size_t smem = (size_t)4 * COLS * S::LDS_WORDS;
  const uint32_t n_groups = (uint32_t)((ncols + COLS - 1) / COLS);
#if !defined(LMN_EMU) && !defined(LMN_BATCH)
  if (smem > 64 * 1024) allow_big_lds((const void*)k_fft_rows_fx<COLS>, 160 * 1024);
#endif
  LMN_LAUNCH((k_fft_rows_fx<COLS>), dim3((1u << (log_n - 12)) * n_groups), dim3((COLS / 2) * 256), smem, s, coeffs, col_stride,
             rows, n_rows, ncols, pad, bad_flag, bad_value, doubled(itw), n_groups);
}

bool launch_fft_rows_fixed(uint32_t* coeffs, uint64_t col_stride, const uint32_t* rows, uint64_t n_rows, int ncols, int log_n,
                           const PadRow& pad, uint32_t* bad_flag, uint32_t bad_value, const TwPtrs& itw, lmn_stream_t s) {
  static const bool off = getenv("LMN_NO_FFT_FIXED") != nullptr;
  if (off || !itw.d[0] || log_n < 13 || ncols < 1 || ncols > 32) return false;
  const int cols = getenv("LMN_ROWS_FX_COLS") ? atoi(getenv("LMN_ROWS_FX_COLS")) : 8;
  if (cols == 4)
    launch_rows_fx<4>(coeffs, col_stride, rows, n_rows, ncols, log_n, pad, bad_flag, bad_value, itw, s);
  else
    launch_rows_fx<8>(coeffs, col_stride, rows, n_rows, ncols, log_n, pad, bad_flag, bad_value, itw, s);
  return true;
}

// experiment knob (docs/SWITCHES.md): bytes of unused dynamic LDS per transform workgroup - caps how many of them a CU holds
static size_t fx_lds_pad() {
  static const size_t pad = getenv("LMN_FFT_LDS_PAD") ? (size_t)atol(getenv("LMN_FFT_LDS_PAD")) : 0;
  return pad;
}

template <bool INV, int RBITS, int CB, bool LO0, bool ZX = false>
static void launch_fx(uint32_t* data, uint64_t col_stride, const uint32_t* src, uint64_t src_stride, int lo, int log_n,
                      const TwPtrs& tw, uint32_t scale_log, int ncols, int cpb, uint32_t h_off, int xcd, lmn_stream_t s) {
  using S = FxShape<RBITS, CB, LO0>;
  const unsigned tiles = 1u << (log_n - S::TB);
  const size_t smem = (size_t)4 * S::LDS_WORDS + fx_lds_pad();
#if !defined(LMN_EMU) && !defined(LMN_BATCH)
  if (smem > 64 * 1024) allow_big_lds((const void*)k_fft_fx<INV, RBITS, CB, LO0, ZX>, 160 * 1024);
#endif
  LMN_LAUNCH((k_fft_fx<INV, RBITS, CB, LO0, ZX>), dim3(tiles, (unsigned)((ncols + cpb - 1) / cpb)), dim3(S::NT), smem, s, data,
             col_stride, src, src_stride, lo, doubled(tw), scale_log, ncols, cpb, h_off, xcd);
}

template <bool INV>
static bool dispatch_fx(uint32_t* data, uint64_t col_stride, const uint32_t* src, uint64_t src_stride, int lo, int rbits, int cb,
                        int log_n, const TwPtrs& tw, uint32_t scale_log, int ncols, int cpb, uint32_t h_off, int xcd,
                        bool zext, lmn_stream_t s) {
#define LMN_FX_CASE(RB, CBV, L0V)                                                                                            \
  if constexpr (!INV && !L0V) {                                                                                              \
    if (zext) {                                                                                                              \
      launch_fx<INV, RB, CBV, L0V, true>(data, col_stride, src, src_stride, lo, log_n, tw, scale_log, ncols, cpb, h_off, xcd, \
                                         s);                                                                                 \
      return true;                                                                                                           \
    }                                                                                                                        \
  }                                                                                                                          \
  if (zext) return false;                                                                                                    \
  launch_fx<INV, RB, CBV, L0V>(data, col_stride, src, src_stride, lo, log_n, tw, scale_log, ncols, cpb, h_off, xcd, s);      \
  return true
  if (lo == 0 && cb == 0) {
    if (rbits == 12) { LMN_FX_CASE(12, 0, true); }
    return false;
  }
  if (cb == 5 && lo >= 5) {
    switch (rbits) {
      case 5: LMN_FX_CASE(5, 5, false);
      case 6: LMN_FX_CASE(6, 5, false);
      case 7: LMN_FX_CASE(7, 5, false);
      default: return false;
    }
  }
  if (cb == 4 && lo >= 4) {
    switch (rbits) {
      case 6: LMN_FX_CASE(6, 4, false);
      case 7: LMN_FX_CASE(7, 4, false);
      case 8: LMN_FX_CASE(8, 4, false);
      case 9: LMN_FX_CASE(9, 4, false);
      case 10: LMN_FX_CASE(10, 4, false);
      default: return false;
    }
  }
  return false;
#undef LMN_FX_CASE
}

bool launch_fft_fixed_pass(bool inverse, uint32_t* data, uint64_t col_stride, const uint32_t* src, uint64_t src_stride, int lo,
                           int rbits, int cb, int log_n, const TwPtrs& tw, uint32_t scale_log, int ncols, int cpb,
                           uint32_t h_off, int xcd_swizzle, bool zero_extended_top, lmn_stream_t s) {
  static const bool off = getenv("LMN_NO_FFT_FIXED") != nullptr;
  if (off || !tw.d[0]) return false;
  if (zero_extended_top && (inverse || lo + rbits != log_n)) return false;
  return inverse ? dispatch_fx<true>(data, col_stride, src, src_stride, lo, rbits, cb, log_n, tw, scale_log, ncols, cpb, h_off,
                                     xcd_swizzle, false, s)
                 : dispatch_fx<false>(data, col_stride, src, src_stride, lo, rbits, cb, log_n, tw, scale_log, ncols, cpb, h_off,
                                      xcd_swizzle, zero_extended_top, s);
}

template <int RBITS>
static void launch_ie(uint32_t* coeffs, uint64_t coeff_stride, uint32_t* lde, uint64_t lde_stride, const TwPtrs& itw,
                      const TwPtrs& tw_ext, uint32_t scale_log, int ncols, lmn_stream_t s) {
  using S = FxShape<RBITS, 4, false>;
  const size_t smem = (size_t)8 * S::LDS_WORDS + fx_lds_pad();
#if !defined(LMN_EMU) && !defined(LMN_BATCH)
  if (smem > 64 * 1024) allow_big_lds((const void*)k_fft_interp_extend_fx<RBITS>, 160 * 1024);
#endif
  LMN_LAUNCH((k_fft_interp_extend_fx<RBITS>), dim3(1u << (12 - 4), (unsigned)ncols), dim3(S::NT), smem, s, coeffs, coeff_stride,
             lde, lde_stride, doubled(itw), doubled(tw_ext), scale_log);
}

bool launch_interp_extend_fixed(uint32_t* coeffs, uint64_t coeff_stride, uint32_t* lde, uint64_t lde_stride, int log_n,
                                const TwPtrs& itw, const TwPtrs& tw_ext, int ncols, lmn_stream_t s) {
  static const bool off = getenv("LMN_NO_FFT_FIXED") != nullptr;
  if (off || !itw.d[0] || !tw_ext.d[0]) return false;
  const uint32_t scale_log = (uint32_t)((31 - (log_n % 31)) % 31);   // 2^-log_n = 2^(31 - log_n mod 31)
  switch (log_n - 12) {
    case 6: launch_ie<6>(coeffs, coeff_stride, lde, lde_stride, itw, tw_ext, scale_log, ncols, s); return true;
    case 7: launch_ie<7>(coeffs, coeff_stride, lde, lde_stride, itw, tw_ext, scale_log, ncols, s); return true;
    case 8: launch_ie<8>(coeffs, coeff_stride, lde, lde_stride, itw, tw_ext, scale_log, ncols, s); return true;
    case 9: launch_ie<9>(coeffs, coeff_stride, lde, lde_stride, itw, tw_ext, scale_log, ncols, s); return true;
    default: return false;
  }
}

}  // namespace lmn

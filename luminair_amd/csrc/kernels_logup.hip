// gfx950 kernels, part 4: the logup interaction trace: fractions with one batched inverse per
// row, claimed-sum reduction, coset-order prefix sum over bit-reversed storage.
#include "kernels_common.h"

namespace lmn {

// =============================================================================================
// a6  LogUp: S_j[r] = S_{j-1}[r] + mult_j[r] / (val_j[r] + alpha*id_j[r] - z)
// =============================================================================================
int logup_num_blocks(uint32_t n) { return (int)cdiv(n, TPB); }

template <int K>
LMN_KERNEL k_logup_fracs(LogupArgs a) {
  LMN_SHARED uint64_t red[TPB * 4];
  uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
  QM31 S = q_zero();
  if (r < a.n) {
    QM31 den[K], pre[K];
    // the row's cells: column-major evaluations, or - a.rows - the table's own rows (launch-uniform choice)
    const bool aos = a.rows != nullptr, real = r < a.n_real;
    const uint32_t* __restrict__ rowp = a.rows + (uint64_t)(real ? r : 0u) * a.row_words;
    uint32_t mults[K];
#pragma unroll
    for (int j = 0; j < K; ++j) {
      uint32_t vj, ij = 0u;
      if (aos) {
        vj = real ? rowp[a.vcol[j]] : a.pad_val[j];
        if (a.icol[j] >= 0) ij = real ? rowp[a.icol[j]] : a.pad_id[j];
        mults[j] = real ? rowp[a.mcol[j]] : a.pad_mult[j];
      } else {
        vj = ld_ub(a.val[j], r);   // (uniform column bases + the row as a 32-bit lane offset: kernels_common.h)
        if (a.id[j]) ij = ld_ub(a.id[j], r);
        mults[j] = ld_ub(a.mult[j], r);
      }
      QM31 d = q_from_m(vj);
      // relation elements: kernel arguments, or the device-resident draws (a.d_elems is launch-uniform)
      const QM31 ez = a.d_elems ? a.d_elems->z[a.es[j]] : a.z[j];
      if (aos ? a.icol[j] >= 0 : a.id[j] != nullptr) d = q_add(d, q_mul_m(a.d_elems ? a.d_elems->alpha[a.es[j]] : a.alpha[j], ij));
      d = q_sub(d, ez);
      den[j] = d;
      pre[j] = j == 0 ? d : q_mul(pre[j - 1], d);
    }
    QM31 inv = q_inv(pre[K - 1]);
    QM31 invs[K];
#pragma unroll
    for (int j = K - 1; j >= 0; --j) {
      invs[j] = j == 0 ? inv : q_mul(inv, pre[j - 1]);
      inv = q_mul(inv, den[j]);
    }
#pragma unroll
    for (int j = 0; j < K; ++j) {
      uint32_t mlt = mults[j];
      if (a.neg[j]) mlt = m_neg(mlt);
      S = q_add(S, q_mul_m(invs[j], mlt));
      if (j < K - 1) {
        uint32_t* o = a.inter + (uint64_t)(4 * j) * a.n;
        st_ub(o, r, S.a);
        st_ub(o + (uint64_t)a.n, r, S.b);
        st_ub(o + (uint64_t)2 * a.n, r, S.c);
        st_ub(o + (uint64_t)3 * a.n, r, S.d);
      }
    }
    a.last_tmp[r] = S;
  }
  red[threadIdx.x * 4 + 0] = S.a;
  red[threadIdx.x * 4 + 1] = S.b;
  red[threadIdx.x * 4 + 2] = S.c;
  red[threadIdx.x * 4 + 3] = S.d;
  __syncthreads();
  for (int st = TPB / 2; st > 0; st >>= 1) {
    if ((int)threadIdx.x < st)
      for (int k = 0; k < 4; ++k) red[threadIdx.x * 4 + k] += red[(threadIdx.x + st) * 4 + k];
    __syncthreads();
  }
  if (threadIdx.x < 4) a.partials[blockIdx.x * 4 + threadIdx.x] = m_red64(red[threadIdx.x]);
}

void launch_logup_fracs(const LogupArgs& a, lmn_stream_t s) {
  if (LMN_ABLATED(32u)) return;
  dim3 g(logup_num_blocks(a.n)), b(TPB);
  switch (a.k) {
    case 1: LMN_LAUNCH(k_logup_fracs<1>, g, b, 0, s, a); break;
    case 2: LMN_LAUNCH(k_logup_fracs<2>, g, b, 0, s, a); break;
    case 3: LMN_LAUNCH(k_logup_fracs<3>, g, b, 0, s, a); break;
    case 7: LMN_LAUNCH(k_logup_fracs<7>, g, b, 0, s, a); break;
    default: throw LmnError(-100, "logup: unsupported relation count");
  }
}

LMN_KERNEL k_logup_reduce(const uint32_t* __restrict__ partials, int nblocks, uint32_t n_inv, QM31* out) {
  LMN_SERIAL_KERNEL();
  LMN_SHARED uint64_t red[TPB * 4];
  uint64_t acc[4] = {0, 0, 0, 0};
  for (int b = threadIdx.x; b < nblocks; b += blockDim.x)
    for (int k = 0; k < 4; ++k) acc[k] += partials[b * 4 + k];
  for (int k = 0; k < 4; ++k) red[threadIdx.x * 4 + k] = m_red64(acc[k]);
  __syncthreads();
  for (int st = TPB / 2; st > 0; st >>= 1) {
    if ((int)threadIdx.x < st)
      for (int k = 0; k < 4; ++k) red[threadIdx.x * 4 + k] += red[(threadIdx.x + st) * 4 + k];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    QM31 c{m_red64(red[0]), m_red64(red[1]), m_red64(red[2]), m_red64(red[3])};
    out[0] = c;
    out[1] = q_mul_m(c, n_inv);
  }
}

void launch_logup_reduce(const uint32_t* partials, int nblocks, uint32_t n_inv, QM31* claimed_out, lmn_stream_t s) {
  LMN_LAUNCH(k_logup_reduce, dim3(1), dim3(TPB), 0, s, partials, nblocks, n_inv, claimed_out);
}

// coset-order position -> storage index (bit-reversed circle-domain order), SURVEY Appendix A.2
LMN_D uint32_t coset_pos_to_storage(uint32_t i, int log_size) {
  uint32_t n = 1u << log_size;
  uint32_t cd = (i & 1u) ? n - ((i + 1u) >> 1) : (i >> 1);
  return log_size == 0 ? 0u : (__brev(cd) >> (32 - log_size));
}

constexpr int SCAN_PER_THREAD = 4;
constexpr int SCAN_PER_BLOCK = TPB * SCAN_PER_THREAD;
int logup_scan_num_blocks(int log_size);

// block-local inclusive scan of thread sums in LDS (Hillis-Steele over TPB QM31 values)
LMN_D QM31 block_scan_inclusive(QM31 v, QM31* sh) {
  sh[threadIdx.x] = v;
  __syncthreads();
  for (int off = 1; off < TPB; off <<= 1) {
    QM31 add = q_zero();
    if ((int)threadIdx.x >= off) add = sh[threadIdx.x - off];
    __syncthreads();
    sh[threadIdx.x] = q_add(sh[threadIdx.x], add);
    __syncthreads();
  }
  return sh[threadIdx.x];
}

// mode 0: write block totals; mode 2: block totals of the unshifted values; mode 1: write scanned values (+ exclusive block offsets)
LMN_KERNEL k_logup_scan(const QM31* __restrict__ last_tmp, const QM31* __restrict__ claimed_shift, int log_size,
                        uint32_t* __restrict__ out_cols, QM31* blocksums, int mode) {
  LMN_SHARED QM31 sh[TPB];
  const uint32_t n = 1u << log_size;
  const QM31 shift = mode == 2 ? q_zero() : claimed_shift[1];   // mode 2: totals of the raw values (the shift does not exist yet)
  uint32_t i0 = (blockIdx.x * TPB + threadIdx.x) * SCAN_PER_THREAD;
  QM31 v[SCAN_PER_THREAD];
  uint32_t st[SCAN_PER_THREAD];
  QM31 run = q_zero();
  for (int k = 0; k < SCAN_PER_THREAD; ++k) {
    uint32_t i = i0 + k;
    if (i < n) {
      st[k] = coset_pos_to_storage(i, log_size);
      run = q_add(run, q_sub(last_tmp[st[k]], shift));
    } else {
      st[k] = 0xffffffffu;
    }
    v[k] = run;
  }
  QM31 incl = block_scan_inclusive(run, sh);
  if (mode != 1) {
    if (threadIdx.x == TPB - 1) blocksums[blockIdx.x] = incl;
    return;
  }
  QM31 offset = q_sub(incl, run);  // exclusive prefix of this thread within the block
  if (blockIdx.x > 0) offset = q_add(offset, blocksums[blockIdx.x - 1]);
  for (int k = 0; k < SCAN_PER_THREAD; ++k) {
    if (st[k] == 0xffffffffu) continue;
    QM31 t = q_add(v[k], offset);
    uint32_t* o = out_cols + st[k];
    o[0] = t.a;
    o[(uint64_t)n] = t.b;
    o[(uint64_t)2 * n] = t.c;
    o[(uint64_t)3 * n] = t.d;
  }
}

// inclusive scan of the block totals in place (single block of up to 1024 lanes; lane t owns a contiguous run)
constexpr int SCAN_SUMS_THREADS = 1024;
// `claim_out` given: the totals are those of the UNSHIFTED values (every block holds `per_block` of the n values, the last
// one possibly fewer).  Their sum is the claimed sum, shift = sum / n; both are written to claim_out[0 .. 1], and the scan
// is that of the shifted values: prefix(b) - (values up to and including block b) * shift.  One launch instead of a
// reduction over the fraction kernel's partial sums in front of the totals.
LMN_KERNEL k_scan_blocksums(QM31* blocksums, int nblocks, QM31* claim_out, uint32_t n_inv, uint32_t per_block, uint32_t n) {
  LMN_SERIAL_KERNEL();
  LMN_SHARED QM31 sh[SCAN_SUMS_THREADS / 64];
  const int T = (int)blockDim.x;
  const int per = (nblocks + T - 1) / T;
  const int b0 = threadIdx.x * per;
  // pass 1: the lane's total (loads in independent batches of 8, so that they overlap)
  QM31 run = q_zero();
  QM31 first[8];   // the lane's first batch stays in registers for pass 2 (up to 8 192 totals: the only batch)
  for (int k0 = 0; k0 < per; k0 += 8) {
    QM31 v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int b = b0 + k0 + j;
      v[j] = (k0 + j < per && b < nblocks) ? blocksums[b] : q_zero();
      if (k0 == 0) first[j] = v[j];
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) run = q_add(run, v[j]);
  }
  // the lanes' totals: an inclusive scan inside each wave (six shuffle steps, no barrier), then the wave totals
  // (at most 16) through LDS - one barrier instead of the twenty a block-wide Hillis-Steele scan takes
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  QM31 inc = run;
#pragma unroll
  for (uint32_t off = 1; off < 64u; off <<= 1) {
    const QM31 o{lmn_shfl_up(inc.a, off), lmn_shfl_up(inc.b, off), lmn_shfl_up(inc.c, off), lmn_shfl_up(inc.d, off)};
    if (lane >= off) inc = q_add(inc, o);
  }
  if (lane == 63u) sh[wave] = inc;
  __syncthreads();
  QM31 before = q_zero();
  for (uint32_t w = 0; w < wave; ++w) before = q_add(before, sh[w]);
  QM31 shift = q_zero();
  if (claim_out) {
    QM31 total = before;
    for (uint32_t w = wave; w < (uint32_t)(T + 63) / 64u; ++w) total = q_add(total, sh[w]);
    shift = q_mul_m(total, n_inv);
    if (threadIdx.x == 0) {
      claim_out[0] = total;
      claim_out[1] = shift;
    }
  }
  // pass 2: inclusive prefix inside the lane's run, starting from the lanes before it
  QM31 acc = q_add(before, q_sub(inc, run));
  for (int k0 = 0; k0 < per; k0 += 8) {
    QM31 v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int b = b0 + k0 + j;
      v[j] = k0 == 0 ? first[j] : ((k0 + j < per && b < nblocks) ? blocksums[b] : q_zero());
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int b = b0 + k0 + j;
      acc = q_add(acc, v[j]);
      if (k0 + j < per && b < nblocks) {
        const uint64_t upto = ((uint64_t)b + 1u) * per_block;
        blocksums[b] = claim_out ? q_sub(acc, q_mul_m(shift, (uint32_t)(upto < n ? upto : n))) : acc;
      }
    }
  }
}

// ---- coalesced coset-order scan (log_size >= SCAN2_MIN_LOG)
// Coset positions 2m and 2m+1 hold circle-domain indices m and n-1-m = ~m, i.e. storage indices s = brev(m) (even)
// and ~s (odd).  With m = mh * 2^A + ml, u = s >> 1 = brev_A(ml) << (k-1-A) | brev(mh): a block of 2^A consecutive
// m (fixed mh) is a stride-2^(k-1-A) comb in storage.  A workgroup therefore takes the 2 * 2^C blocks whose brev(mh)
// is (G << C) | x for G in {g, ~g} and all x < 2^C: their even elements are the even words of 2^A contiguous runs
// of 2^(C+1) storage indices in region g, their odd partners the odd words of the runs in region ~g, and vice versa -
// every line the workgroup touches is used completely, loads (AoS QM31) and stores (4 SoA columns) are contiguous
// runs of 2^(C+1) elements.  Three launches: block totals, scan of the 2^(k-1-A) totals, prefix + write
// (48 B per row of HBM traffic for 32 B of algorithmic bytes; the scattered version moved ~160 B per row).
constexpr int SCAN2_A = 6, SCAN2_C = 4;
constexpr int SCAN2_MIN_LOG = SCAN2_A + SCAN2_C + 2;
constexpr int SCAN2_ELEMS = 2 << (SCAN2_A + SCAN2_C + 1);  // QM31 values per workgroup (4096 = 64 KB)
static_assert(SCAN2_ELEMS == TPB * 16 && (1 << (SCAN2_C + 1)) * 8 == TPB, "one 8-lane group per block of positions");

template <int MODE>
LMN_KERNEL k_logup_scan2(const QM31* __restrict__ last_tmp, const QM31* __restrict__ claimed_shift, int log_size,
                         uint32_t* __restrict__ out_cols, QM31* __restrict__ blocksums) {
  LMN_SERIAL_KERNEL();
  LMN_DYN_SMEM(QM31, T);
  constexpr int A = SCAN2_A, C = SCAN2_C;
  const int gbits = log_size - 1 - A - C;                 // bits of the region index G
  const uint32_t n = 1u << log_size;
  const uint32_t g = blockIdx.x, gmask = (1u << gbits) - 1u;
  const QM31 shift = MODE == 2 ? q_zero() : claimed_shift[1];   // MODE 2: block totals of the unshifted values
  // element e of the tile: ((Gi * 2^A + r) * 2^C + x) * 2 + parity  <->  storage 2u + parity,
  // u = r << (k-1-A) | G << C | x   (r = brev_A(ml))
  auto storage_of = [&](uint32_t e) {
    const uint32_t par = e & 1u, x = (e >> 1) & ((1u << C) - 1u), r = (e >> (1 + C)) & ((1u << A) - 1u), gi = e >> (1 + C + A);
    const uint32_t G = gi ? (~g & gmask) : g;
    const uint32_t u = (r << (log_size - 1 - A)) | (G << C) | x;
    return 2u * u + par;
  };
  for (int i = 0; i < 16; ++i) {
    const uint32_t e = (uint32_t)i * TPB + threadIdx.x;
    T[e] = q_sub(last_tmp[storage_of(e)], shift);
  }
  __syncthreads();
  // 8 lanes per block of 2^A positions pairs; lane `part` owns m = part*8 .. part*8+7 (16 elements)
  const uint32_t blk = threadIdx.x >> 3, part = threadIdx.x & 7u;
  const uint32_t gi = blk >> C, x = blk & ((1u << C) - 1u);
  uint32_t slot[16];
  QM31 v[16];
  QM31 run = q_zero();
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const uint32_t ml = part * 8u + (uint32_t)j;
    const uint32_t r = __brev(ml) >> (32 - A);
    slot[2 * j] = (((gi << A) + r) << C | x) << 1;                                                  // storage 2u
    slot[2 * j + 1] = ((((1u - gi) << A) + ((1u << A) - 1u - r)) << C | ((1u << C) - 1u - x)) << 1 | 1u;  // storage ~(2u)
    run = q_add(run, T[slot[2 * j]]);
    v[2 * j] = run;
    run = q_add(run, T[slot[2 * j + 1]]);
    v[2 * j + 1] = run;
  }
  // inclusive scan of the 8 lanes' sums (xor butterfly inside the 8-lane group)
  QM31 pre = run, tot = run;
#pragma unroll
  for (int d = 1; d < 8; d <<= 1) {
    QM31 o;
    o.a = lmn_shfl_xor(tot.a, d);
    o.b = lmn_shfl_xor(tot.b, d);
    o.c = lmn_shfl_xor(tot.c, d);
    o.d = lmn_shfl_xor(tot.d, d);
    if (part & (uint32_t)d) pre = q_add(pre, o);
    tot = q_add(tot, o);
  }
  const uint32_t G = gi ? (~g & gmask) : g;
  const uint32_t mh = __brev((G << C) | x) >> (32 - (gbits + C));
  if (MODE != 1) {
    if (part == 0) blocksums[mh] = tot;
    return;
  }
  QM31 off = q_sub(pre, run);
  if (mh > 0) off = q_add(off, blocksums[mh - 1]);
#pragma unroll
  for (int j = 0; j < 16; ++j) T[slot[j]] = q_add(v[j], off);
  __syncthreads();
  for (int i = 0; i < 16; ++i) {
    const uint32_t e = (uint32_t)i * TPB + threadIdx.x;
    const QM31 t = T[e];
    uint32_t* o = out_cols + storage_of(e);
    o[0] = t.a;
    o[(uint64_t)n] = t.b;
    o[(uint64_t)2 * n] = t.c;
    o[(uint64_t)3 * n] = t.d;
  }
}

int logup_scan_num_blocks(int log_size) {
  if (log_size >= SCAN2_MIN_LOG) return 1 << (log_size - 1 - SCAN2_A);   // one total per block of 2^A position pairs
  return (int)cdiv(1ull << log_size, SCAN_PER_BLOCK);
}

void launch_logup_scan(const QM31* last_tmp, QM31* claimed_shift, int log_size, uint32_t* out_cols, QM31* blocksums,
                       lmn_stream_t s, bool derive_claim, uint32_t n_inv) {
  static const bool scattered = getenv("LMN_LOGUP_SCAN_V1") != nullptr;   // ablation: the round-1 kernel
  int nb = logup_scan_num_blocks(log_size);
  const uint32_t n = 1u << log_size;
  QM31* claim_out = derive_claim ? claimed_shift : nullptr;
  const dim3 sums_block(nb > 2048 ? SCAN_SUMS_THREADS : TPB);
  if (log_size >= SCAN2_MIN_LOG && !scattered) {
    const dim3 grid(1u << (log_size - 2 - SCAN2_A - SCAN2_C));
    const size_t smem = (size_t)SCAN2_ELEMS * sizeof(QM31);
#if !defined(LMN_EMU) && !defined(LMN_BATCH)
    allow_big_lds((const void*)k_logup_scan2<0>, (int)smem);
    allow_big_lds((const void*)k_logup_scan2<1>, (int)smem);
    allow_big_lds((const void*)k_logup_scan2<2>, (int)smem);
#endif
    if (derive_claim)
      LMN_LAUNCH(k_logup_scan2<2>, grid, dim3(TPB), smem, s, last_tmp, (const QM31*)claimed_shift, log_size, out_cols, blocksums);
    else
      LMN_LAUNCH(k_logup_scan2<0>, grid, dim3(TPB), smem, s, last_tmp, (const QM31*)claimed_shift, log_size, out_cols, blocksums);
    LMN_LAUNCH(k_scan_blocksums, dim3(1), sums_block, 0, s, blocksums, nb, claim_out, n_inv, 2u << SCAN2_A, n);
    LMN_LAUNCH(k_logup_scan2<1>, grid, dim3(TPB), smem, s, last_tmp, (const QM31*)claimed_shift, log_size, out_cols, blocksums);
    return;
  }
  if (scattered) nb = (int)cdiv(1ull << log_size, SCAN_PER_BLOCK);
  LMN_LAUNCH(k_logup_scan, dim3(nb), dim3(TPB), 0, s, last_tmp, (const QM31*)claimed_shift, log_size, out_cols, blocksums,
             derive_claim ? 2 : 0);
  LMN_LAUNCH(k_scan_blocksums, dim3(1), dim3(nb > 2048 ? SCAN_SUMS_THREADS : TPB), 0, s, blocksums, nb, claim_out, n_inv,
             (uint32_t)SCAN_PER_BLOCK, n);
  LMN_LAUNCH(k_logup_scan, dim3(nb), dim3(TPB), 0, s, last_tmp, (const QM31*)claimed_shift, log_size, out_cols, blocksums, 1);
}

}  // namespace lmn

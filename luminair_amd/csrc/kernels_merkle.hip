// gfx950 kernels, part 3: Blake2s Merkle trees in fused subtree launches, the device-resident channel of the FRI commit
// loop, the single-block FRI tail and the decommitment gather.
#include "kernels_common.h"

namespace lmn {

// =============================================================================================
// a4  Blake2s Merkle layer: one lane per node; a wave reads 64 consecutive rows of each column.
// =============================================================================================
LMN_KERNEL k_merkle_layer(const uint32_t* __restrict__ prev, const uint32_t* const* __restrict__ cols, int ncols,
                          uint32_t size, uint32_t* __restrict__ out) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= size) return;
  const int npre = prev ? 16 : 0;
  const int nwords = npre + ncols;
  const int nblocks = (nwords + 15) / 16;
  uint32_t h[8];
  b2_init(h);
  for (int b = 0; b < nblocks; ++b) {
    uint32_t m[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      int w = b * 16 + k;
      uint32_t v = 0u;
      if (w < npre)
        v = prev[(uint64_t)i * 16 + w];
      else if (w < nwords)
        v = cols[w - npre][i];
      m[k] = v;
    }
    bool last = b + 1 == nblocks;
    b2_compress(h, m, last ? (uint32_t)(4 * nwords) : (uint32_t)(64 * (b + 1)), last ? 0xffffffffu : 0u);
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) out[(uint64_t)i * 8 + k] = h[k];
}

void launch_merkle_layer(const uint32_t* prev, const uint32_t* const* cols, int ncols, uint32_t size, uint32_t* out,
                         lmn_stream_t s) {
  if (!prev && ncols == 0) throw LmnError(-100, "merkle layer with no input");
  LMN_LAUNCH(k_merkle_layer, dim3(cdiv(size, TPB)), dim3(TPB), 0, s, prev, cols, ncols, size, out);
}

// Fused Merkle subtree.  Every lane owns 2^sub consecutive start-level nodes and reduces them to one
// subtree root in registers (post-order, private LDS slots as the merge stack: all 64 lanes of a
// wave stay busy on every compression); the block's subtree roots then climb further levels
// through LDS.  All levels are written to HBM (decommitment needs them).
// Start-level columns are described as up to MERKLE_MAX_SEG runs of contiguous columns
// (column c of a run lives at base + c*size), so no per-column pointer loads are needed.
LMN_D const uint32_t* merkle_col_ptr(const MerkleSegs& sg, int c, uint64_t size) {
  // c is compile-time after unrolling; the comparisons are wave-uniform scalar work
  int n0 = sg.n[0], n1 = n0 + sg.n[1], n2 = n1 + sg.n[2];
  if (c < n0) return sg.base[0] + (uint64_t)c * size;
  if (c < n1) return sg.base[1] + (uint64_t)(c - n0) * size;
  if (c < n2) return sg.base[2] + (uint64_t)(c - n1) * size;
  return sg.base[3] + (uint64_t)(c - n2) * size;
}
// First 16 message words of start-level node i: the two child hashes when the level has a `prev`
// layer, else its first 16 columns (zero padded).  Split from the hashing so that callers can issue the
// loads of the next node before compressing the current one.
LMN_D void merkle_load_first(const uint32_t* __restrict__ prev, const MerkleSegs& sg, int ncols, uint32_t size,
                             uint32_t i, uint32_t m[16]) {
  if (prev) {
    const uint4* p4 = reinterpret_cast<const uint4*>(prev) + (uint64_t)i * 4;
    uint4 a = p4[0], b = p4[1], c = p4[2], d = p4[3];
    m[0] = a.x; m[1] = a.y; m[2] = a.z; m[3] = a.w;
    m[4] = b.x; m[5] = b.y; m[6] = b.z; m[7] = b.w;
    m[8] = c.x; m[9] = c.y; m[10] = c.z; m[11] = c.w;
    m[12] = d.x; m[13] = d.y; m[14] = d.z; m[15] = d.w;
  } else {
#pragma unroll
    for (int k = 0; k < 16; ++k) m[k] = k < ncols ? merkle_col_ptr(sg, k, size)[i] : 0u;
  }
}

// Hash of start-level node i given its first 16 message words (merkle_load_first).
LMN_D void merkle_hash_from(const uint32_t* __restrict__ prev, const MerkleSegs& sg, int ncols, uint32_t size,
                            uint32_t i, uint32_t m[16], uint32_t h[8]) {
  b2_init(h);
  const uint32_t total = (prev ? 64u : 0u) + 4u * (uint32_t)ncols;
  int c0 = prev ? 0 : 16;  // first column not yet consumed
  if (c0 >= ncols) {
    b2_compress(h, m, total, 0xffffffffu);
    return;
  }
  b2_compress(h, m, 64u, 0u);
  uint32_t done = 64u;
  while (c0 < ncols) {
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      int c = c0 + k;
      m[k] = c < ncols ? merkle_col_ptr(sg, c, size)[i] : 0u;
    }
    c0 += 16;
    bool last = c0 >= ncols;
    done += 64u;
    b2_compress(h, m, last ? total : done, last ? 0xffffffffu : 0u);
  }
}

LMN_D void merkle_hash_start(const uint32_t* __restrict__ prev, const MerkleSegs& sg, int ncols, uint32_t size,
                             uint32_t i, uint32_t h[8]) {
  uint32_t m[16];
  merkle_load_first(prev, sg, ncols, size, i, m);
  merkle_hash_from(prev, sg, ncols, size, i, m, h);
}

// MerkleFold::below: raw words of leaves 2i (m[0..7]) and 2i+1 (m[8..15]) of the level under the start level
LMN_D void merkle_load_below(const uint32_t* __restrict__ below, int below_ncols, uint32_t size, uint32_t i, uint32_t m[16]) {
  const uint64_t L = 2ull * size;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    if (k < below_ncols) {
      ld_ub_pair(below + (uint64_t)k * L, i, m[k], m[8 + k]);   // leaves 2i and 2i+1 of column k: one 8-byte access
    } else {
      m[k] = 0u;
      m[8 + k] = 0u;
    }
  }
}
// start node i = H(H(leaf 2i) || H(leaf 2i+1) || own columns) from the raw leaf words of merkle_load_below
LMN_D void merkle_hash_below(const MerkleSegs& sg, int ncols, uint32_t size, int below_ncols, uint32_t i, const uint32_t m[16],
                             uint32_t h[8]) {
  uint32_t ml[16], mr[16], hl[8], hr[8];
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    ml[k] = k < 8 ? m[k] : 0u;
    mr[k] = k < 8 ? m[8 + k] : 0u;
  }
  b2_compress_fresh_nz<8>(hl, ml, 4u * (uint32_t)below_ncols);
  b2_compress_fresh_nz<8>(hr, mr, 4u * (uint32_t)below_ncols);
  uint32_t m2[16];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    m2[k] = hl[k];
    m2[8 + k] = hr[k];
  }
  // merkle_hash_from only asks whether the node HAS children; their hashes are in m2
  merkle_hash_from(reinterpret_cast<const uint32_t*>(sg.base[0]), sg, ncols, size, i, m2, h);
}

LMN_D void store_hash(uint32_t* __restrict__ o, const uint32_t h[8]) {
  uint4* o4 = reinterpret_cast<uint4*>(o);
  o4[0] = make_uint4(h[0], h[1], h[2], h[3]);
  o4[1] = make_uint4(h[4], h[5], h[6], h[7]);
}

// Quad-cooperative Blake2s of one 64-byte final block (a Merkle parent): lane q of a quad owns column q
// of the 4x4 state (a, b, c, d = rows), so the four G functions of a half-round run on four lanes; the
// diagonal step rotates rows b, c, d by 1, 2, 3 lanes with DPP quad_perm.  Message words are fetched from
// LDS by per-lane sigma offsets.  ~400 issue slots instead of ~1000: single-hash latency 0.93 us vs 2.0 us
// on MI355X (round 2's single-hash microbenchmark, docs/HISTORY.md) - used where a level is too narrow to fill lanes anyway.
// Returns words q and 4+q of the digest.  Must be executed by all four lanes of the quad.
// WS = 0: the 16 message words are contiguous at msg; WS > 0: the two child hashes live in a word-major node array
// (word k of node j at base[k * WS + j]): message word w = msg[(w & 7) * WS + (w >> 3)] with msg = base + 2 * parent.
template <uint32_t WS = 0u>
LMN_D void b2_quad_parent(const uint32_t* msg, uint32_t q, uint32_t& o_lo, uint32_t& o_hi, uint32_t t0 = 64u) {
#define LMN_B2_QW(w) (WS ? msg[((w) & 7u) * WS + ((w) >> 3)] : msg[(w)])
#ifdef LMN_EMU
  // CPU emulation (tests only): a cross-lane rendezvous per DPP move would be a block-wide fiber switch;
  // every lane hashes the block alone and keeps its two words.  The DPP path is checked on the GPU.
  uint32_t h[8], m[16];
  for (uint32_t k = 0; k < 16; ++k) m[k] = LMN_B2_QW(k);
  b2_compress_fresh(h, m, t0);
  o_lo = h[q];
  o_hi = h[4 + q];
  return;
#endif
  const uint32_t iv_lo = q == 0 ? 0x6A09E667u : q == 1 ? 0xBB67AE85u : q == 2 ? 0x3C6EF372u : 0xA54FF53Au;
  const uint32_t iv_hi = q == 0 ? 0x510E527Fu : q == 1 ? 0x9B05688Cu : q == 2 ? 0x1F83D9ABu : 0x5BE0CD19u;
  const uint32_t h_lo = q == 0 ? (0x6A09E667u ^ 0x01010020u) : iv_lo;
  uint32_t a = h_lo, b = iv_hi, c = iv_lo, d = iv_hi ^ (q == 0 ? t0 : q == 2 ? 0xffffffffu : 0u);
  const uint32_t sh8 = 8u * q;
#define LMN_B2_QUAD_ROUND(...)                                            \
  {                                                                       \
    const uint64_t S = LMN_B2_SIGMA_PACK(__VA_ARGS__);                    \
    const uint32_t lo = (uint32_t)(S >> sh8), hi = (uint32_t)(S >> (32u + sh8)); \
    const uint32_t x0 = LMN_B2_QW(lo & 15u), y0 = LMN_B2_QW((lo >> 4) & 15u); \
    const uint32_t x1 = LMN_B2_QW(hi & 15u), y1 = LMN_B2_QW((hi >> 4) & 15u); \
    LMN_B2_G(a, b, c, d, x0, y0)                                          \
    b = lmn_quad_perm(b, 0x39);                                           \
    c = lmn_quad_perm(c, 0x4E);                                           \
    d = lmn_quad_perm(d, 0x93);                                           \
    LMN_B2_G(a, b, c, d, x1, y1)                                          \
    b = lmn_quad_perm(b, 0x93);                                           \
    c = lmn_quad_perm(c, 0x4E);                                           \
    d = lmn_quad_perm(d, 0x39);                                           \
  }
  LMN_B2_QUAD_ROUND(0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15)
  LMN_B2_QUAD_ROUND(14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3)
  LMN_B2_QUAD_ROUND(11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4)
  LMN_B2_QUAD_ROUND(7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8)
  LMN_B2_QUAD_ROUND(9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13)
  LMN_B2_QUAD_ROUND(2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9)
  LMN_B2_QUAD_ROUND(12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11)
  LMN_B2_QUAD_ROUND(13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10)
  LMN_B2_QUAD_ROUND(6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5)
  LMN_B2_QUAD_ROUND(10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0)
#undef LMN_B2_QUAD_ROUND
  o_lo = h_lo ^ a ^ c;
  o_hi = iv_hi ^ b ^ d;
#undef LMN_B2_QW
}

// The same for ANY block of a hash in progress: lane q of the quad holds words q and 4 + q of the chaining value
// (hl, hh: updated); t0 = bytes hashed so far including this block, f0 = the finalisation word (0xffffffff on the last block).
LMN_D void b2_quad_compress(const uint32_t* msg, uint32_t q, uint32_t& hl, uint32_t& hh, uint32_t t0, uint32_t f0) {
  const uint32_t iv_lo = q == 0 ? 0x6A09E667u : q == 1 ? 0xBB67AE85u : q == 2 ? 0x3C6EF372u : 0xA54FF53Au;
  const uint32_t iv_hi = q == 0 ? 0x510E527Fu : q == 1 ? 0x9B05688Cu : q == 2 ? 0x1F83D9ABu : 0x5BE0CD19u;
  uint32_t a = hl, b = hh, c = iv_lo, d = iv_hi ^ (q == 0 ? t0 : q == 2 ? f0 : 0u);
  const uint32_t sh8 = 8u * q;
#define LMN_B2_QUAD_ROUND(...)                                            \
  {                                                                       \
    const uint64_t S = LMN_B2_SIGMA_PACK(__VA_ARGS__);                    \
    const uint32_t lo = (uint32_t)(S >> sh8), hi = (uint32_t)(S >> (32u + sh8)); \
    const uint32_t x0 = msg[lo & 15u], y0 = msg[(lo >> 4) & 15u];         \
    const uint32_t x1 = msg[hi & 15u], y1 = msg[(hi >> 4) & 15u];         \
    LMN_B2_G(a, b, c, d, x0, y0)                                          \
    b = lmn_quad_perm(b, 0x39);                                           \
    c = lmn_quad_perm(c, 0x4E);                                           \
    d = lmn_quad_perm(d, 0x93);                                           \
    LMN_B2_G(a, b, c, d, x1, y1)                                          \
    b = lmn_quad_perm(b, 0x93);                                           \
    c = lmn_quad_perm(c, 0x4E);                                           \
    d = lmn_quad_perm(d, 0x39);                                           \
  }
  LMN_B2_QUAD_ROUND(0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15)
  LMN_B2_QUAD_ROUND(14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3)
  LMN_B2_QUAD_ROUND(11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4)
  LMN_B2_QUAD_ROUND(7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8)
  LMN_B2_QUAD_ROUND(9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13)
  LMN_B2_QUAD_ROUND(2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9)
  LMN_B2_QUAD_ROUND(12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11)
  LMN_B2_QUAD_ROUND(13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10)
  LMN_B2_QUAD_ROUND(6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5)
  LMN_B2_QUAD_ROUND(10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0)
#undef LMN_B2_QUAD_ROUND
  hl ^= a ^ c;
  hh ^= b ^ d;
}

// One level of an in-LDS Merkle climb.  The block's nodes live WORD-MAJOR in sh (word k of node j at sh[k * BLOCK + j]:
// lanes that walk nodes touch consecutive banks - the node-major form, 8 or 16 words per lane, put a whole wave on two
// banks): the children of this block's `n_par` parents are nodes 0 .. 2 * n_par - 1, parent j replaces node j and is
// written to out[(node0 + j)*8 ..].  Levels with at most 128 parents (two quad-waves per SIMD of the CU) use four lanes
// per hash, which is faster there; wider levels are throughput-bound inside the CU and keep one lane per hash.
// Block-uniform arguments; ends WITHOUT a barrier.
template <int BLOCK>
LMN_D void merkle_lds_level(uint32_t* sh, uint32_t* __restrict__ out, uint32_t node0, uint32_t n_par) {
  __syncthreads();
  if (n_par * 4u <= (uint32_t)BLOCK && n_par <= 128u) {
    const uint32_t g = threadIdx.x >> 2, q = threadIdx.x & 3u;
    const bool on = g < n_par;
    uint32_t o_lo = 0u, o_hi = 0u;
    const bool wave_on = ((threadIdx.x & ~63u) >> 2) < n_par;  // wave-uniform
    if (wave_on) b2_quad_parent<(uint32_t)BLOCK>(sh + 2u * g, q, o_lo, o_hi);
    __syncthreads();
    if (on) {
      sh[q * BLOCK + g] = o_lo;
      sh[(4u + q) * BLOCK + g] = o_hi;
      uint32_t* o = out + (uint64_t)(node0 + g) * 8;
      o[q] = o_lo;
      o[4u + q] = o_hi;
    }
  } else {
    const bool on = threadIdx.x < n_par;
    uint32_t cur[8];
    if (on) {
      uint32_t m[16];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        m[k] = sh[k * BLOCK + 2u * threadIdx.x];
        m[8 + k] = sh[k * BLOCK + 2u * threadIdx.x + 1u];
      }
      b2_compress_fresh(cur, m, 64u);
      store_hash(out + (uint64_t)(node0 + threadIdx.x) * 8, cur);
    }
    __syncthreads();
    if (on) {
#pragma unroll
      for (int k = 0; k < 8; ++k) sh[k * BLOCK + threadIdx.x] = cur[k];
    }
  }
}

// LDS climb shared by the Merkle kernels: `lvl_size` hashes of the whole level exist, this block's share
// sits in sh[idx*8..]; levels first..last are produced.
template <int BLOCK>
LMN_D void merkle_lds_climb(uint32_t* sh, const MerkleLevels& outs, int first, int last, uint32_t lvl_size) {
  for (int l = first; l <= last; ++l) {
    lvl_size >>= 1;
    const uint32_t active = (uint32_t)BLOCK >> (l - first + 1);
    const uint32_t node0 = blockIdx.x * active;
    const uint32_t n_par = node0 >= lvl_size ? 0u : (lvl_size - node0 < active ? lvl_size - node0 : active);
    merkle_lds_level<BLOCK>(sh, outs.p[l], node0, n_par);
  }
}

// MODE 1: leaf level of a single-size tree (no child layer, one contiguous run of <= 16 columns: one
// compression per leaf); MODE 2: pure inner level (children only); MODE 0: anything else.  The special modes
// drop the run selection and the multi-block loop from the hot loop.
// ZT: the caller keeps the message words this mode never loads at zero (set once, outside its leaf loop), so they are not
// rewritten for every leaf
template <int MODE, bool ZT = false>
LMN_D void merkle_load_mode(const uint32_t* __restrict__ prev, const MerkleSegs& sg, int ncols, uint32_t size,
                            uint32_t i, uint32_t m[16], const MerkleFold& fold = MerkleFold{}) {
  if (MODE == 3) {
    // leaf i of a FRI layer = fold of the pair (2i, 2i+1) of the previous layer (same arithmetic as k_fold); the layer's
    // coordinate columns are block-uniform bases, the pair is one 8-byte buffer access per coordinate (kernels_common.h)
    const uint64_t L = 2ull * size;
    QM31 a, b;
    ld_ub_pair(fold.src, i, a.a, b.a);
    ld_ub_pair(fold.src + L, i, a.b, b.b);
    ld_ub_pair(fold.src + 2 * L, i, a.c, b.c);
    ld_ub_pair(fold.src + 3 * L, i, a.d, b.d);
    const QM31 alpha = *fold.alpha;
    QM31 r = q_add(q_add(a, b), q_mul(alpha, q_mul_m(q_sub(a, b), ld_ub(fold.itw, i))));
    if (fold.src2) {   // block-uniform: a quotient column joins this layer (k_fold with accumulate = 1)
      QM31 c, d;
      ld_ub_pair(fold.src2, i, c.a, d.a);
      ld_ub_pair(fold.src2 + L, i, c.b, d.b);
      ld_ub_pair(fold.src2 + 2 * L, i, c.c, d.c);
      ld_ub_pair(fold.src2 + 3 * L, i, c.d, d.d);
      const QM31 rc = q_add(q_add(c, d), q_mul(alpha, q_mul_m(q_sub(c, d), ld_ub(fold.itw2, i))));
      r = q_add(q_mul(r, q_mul(alpha, alpha)), rc);
    }
    st_ub(fold.dst, i, r.a);
    st_ub(fold.dst + (uint64_t)size, i, r.b);
    st_ub(fold.dst + 2ull * size, i, r.c);
    st_ub(fold.dst + 3ull * size, i, r.d);
    m[0] = r.a;
    m[1] = r.b;
    m[2] = r.c;
    m[3] = r.d;
    if (!ZT) {
#pragma unroll
      for (int k = 4; k < 16; ++k) m[k] = 0u;
    }
  } else if (MODE == 1) {
    const uint32_t* __restrict__ base = sg.base[0];   // block-uniform column bases + the leaf as a 32-bit lane offset
#ifdef LMN_ABLATE
    // experiment build, mask 128: the 4-column leaves of the composition tree read a 16 KB stand-in that stays in L2 (garbage
    // hashes): the upper bound of what feeding those leaves from the last forward pass's LDS tile could save
    if (fold.below_ncols == -128) i &= 4095u;
#endif
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      if (k < ncols)
        m[k] = ld_col(base, k, size, i);
      else if (!ZT)
        m[k] = 0u;
    }
  } else if (MODE == 2) {
    const uint4* p4 = reinterpret_cast<const uint4*>(prev) + (uint64_t)i * 4;
    uint4 a = p4[0], b = p4[1], c = p4[2], d = p4[3];
    m[0] = a.x; m[1] = a.y; m[2] = a.z; m[3] = a.w;
    m[4] = b.x; m[5] = b.y; m[6] = b.z; m[7] = b.w;
    m[8] = c.x; m[9] = c.y; m[10] = c.z; m[11] = c.w;
    m[12] = d.x; m[13] = d.y; m[14] = d.z; m[15] = d.w;
  } else if (MODE == 4) {
    merkle_load_below(fold.below, fold.below_ncols, size, i, m);
  } else {
    merkle_load_first(prev, sg, ncols, size, i, m);
  }
}
// NZ (MODE 1): the leaf has at most NZ columns - message words NZ..15 are zero at compile time (b2_compress_fresh_nz)
template <int MODE, int NZ = 16>
LMN_D void merkle_hash_mode(const uint32_t* __restrict__ prev, const MerkleSegs& sg, int ncols, uint32_t size,
                            uint32_t i, uint32_t m[16], uint32_t h[8], const MerkleFold& fold = MerkleFold{}) {
  if (MODE == 4) {
    merkle_hash_below(sg, ncols, size, fold.below_ncols, i, m, h);
  } else if (MODE == 3) {
    b2_compress_fresh_nz<4>(h, m, 16u);
  } else if (MODE == 1) {
    b2_compress_fresh_nz<NZ>(h, m, 4u * (uint32_t)ncols);
  } else if (MODE == 2) {
    b2_compress_fresh(h, m, 64u);
  } else {
    merkle_hash_from(prev, sg, ncols, size, i, m, h);
  }
}

template <int MODE, int NZ = 16>
LMN_KERNEL k_merkle_fused(const uint32_t* __restrict__ prev, MerkleSegs sg, int ncols, uint32_t size,
                          MerkleLevels outs, int sub, int nfused, MerkleFold fold) {
  // Wave-cooperative subtree: in batch j lane l hashes start node W0 + 64*j + l (coalesced column
  // loads and hash stores).  Siblings sit in neighbouring lanes, so after every second batch the
  // lanes swap one hash with lane^1 and ALL 64 lanes compress one level-1 parent (even lanes for the
  // older batch, odd lanes for the newer one); after every fourth batch the same with lane^2, etc.
  // Each lane keeps its pending hash per level in private LDS slots (word 8 = node index).
  LMN_SHARED uint32_t stack[MERKLE_MAX_SUB * 8 * TPB];  // [level][word][thread]
  uint32_t* sh = stack;  // the climb buffer reuses the stack storage once the batch loop is over
  const uint32_t per = 1u << sub;
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t W0 = (t >> 6) * (64u << sub);
  uint32_t cur[8];
  uint32_t cur_idx = 0;
  // software pipeline: the next batch's loads are in flight while this batch is compressed (only ~2 waves share a SIMD
  // here, too few to hide HBM latency by occupancy alone).  Two message buffers alternate (the loop body handles an even and
  // an odd batch), so the prefetched words are hashed where they were loaded - no 16-register copy per leaf.
  uint32_t mA[16], mB[16];
  constexpr bool ZT = MODE == 1 || MODE == 3;   // the words beyond a leaf's columns stay zero for the whole kernel
  if (ZT) {
#pragma unroll
    for (int k = 0; k < 16; ++k) mA[k] = mB[k] = 0u;
  }
  merkle_load_mode<MODE, ZT>(prev, sg, ncols, size, W0 + lane, mA, fold);
  auto leaf = [&](uint32_t j, uint32_t (&mc)[16], uint32_t (&mn)[16]) {
    const uint32_t node = W0 + 64u * j + lane;
    cur_idx = node;
    if (j + 1 < per) merkle_load_mode<MODE, ZT>(prev, sg, ncols, size, node + 64u, mn, fold);
    merkle_hash_mode<MODE, NZ>(prev, sg, ncols, size, node, mc, cur, fold);
    if (outs.p[0]) store_hash(outs.p[0] + (uint64_t)node * 8, cur);
  };
  for (uint32_t j = 0; j < per; j += 2) {
    leaf(j, mA, mB);
    if (per == 1) break;                       // sub = 0: one leaf per lane, nothing to merge in registers
#pragma unroll
    for (int k = 0; k < 8; ++k) stack[k * TPB + threadIdx.x] = cur[k];   // level-0 slot: the even batch waits for its sibling
    leaf(j + 1, mB, mA);
    uint32_t jj = j + 1;
    int lvl = 0;
    while (jj & 1u) {
      const bool b = ((lane >> lvl) & 1u) != 0u;
      uint32_t m[16];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const uint32_t st = stack[(lvl * 8 + k) * TPB + threadIdx.x];
        const uint32_t recv = lmn_shfl_xor(b ? st : cur[k], 1 << lvl);
        m[k] = b ? recv : st;
        m[8 + k] = b ? cur[k] : recv;
      }
      // the pending (older) node of this lane sits exactly 64 nodes before the newer one at every level
      cur_idx = (b ? cur_idx : cur_idx - 64u) >> 1;
      b2_compress_fresh(cur, m, 64u);
      jj >>= 1;
      ++lvl;
      if (outs.p[lvl]) store_hash(outs.p[lvl] + (uint64_t)cur_idx * 8, cur);
    }
    if (lvl < sub) {
#pragma unroll
      for (int k = 0; k < 8; ++k) stack[(lvl * 8 + k) * TPB + threadIdx.x] = cur[k];
    }
  }
  __syncthreads();  // every lane is done with its stack slots before they are overwritten
  {
    const uint32_t local = cur_idx - blockIdx.x * TPB;  // this block owns level-`sub` nodes [b*TPB, (b+1)*TPB)
#pragma unroll
    for (int k = 0; k < 8; ++k) sh[k * TPB + local] = cur[k];   // word-major (merkle_lds_level)
  }
  merkle_lds_climb<TPB>(sh, outs, sub + 1, nfused, size >> sub);
}

LMN_D void chan_draw_words(DevChannel* ch, uint32_t out[8]) {
  uint32_t m[16];
#pragma unroll
  for (int k = 0; k < 8; ++k) m[k] = ch->digest[k];
  m[8] = ch->n_sent;
#pragma unroll
  for (int k = 9; k < 16; ++k) m[k] = 0u;
  b2_init(out);
  // KAT encoding: digest || u64 counter zero-padded to 32 bytes (64-byte message);
  // LMN_PV_DRAW_CTR_U32: digest || u32 counter || 0x00 (37-byte message)
  b2_compress(out, m, ch->variant == 0u ? 64u : 37u, 0xffffffffu);
  ch->n_sent += 1u;
}

// digest <- H(digest || root); alpha <- draw_felt(); executed by ONE lane
LMN_D void chan_mix_root_draw(DevChannel* ch, const uint32_t* root, QM31* out_alpha, uint32_t* root_copy) {
  uint32_t m[16], h[8];
  for (int k = 0; k < 8; ++k) {
    m[k] = ch->digest[k];
    m[8 + k] = root[k];
    root_copy[k] = root[k];
  }
  b2_compress_fresh(h, m, 64u);
  for (int k = 0; k < 8; ++k) ch->digest[k] = h[k];
  ch->n_sent = 0u;
  for (;;) {
    uint32_t w[8];
    chan_draw_words(ch, w);
    bool ok = true;
    for (int k = 0; k < 8; ++k) ok = ok && (w[k] < 2u * P31);
    if (!ok) continue;
    QM31 a;
    a.a = w[0] >= P31 ? w[0] - P31 : w[0];
    a.b = w[1] >= P31 ? w[1] - P31 : w[1];
    a.c = w[2] >= P31 ? w[2] - P31 : w[2];
    a.d = w[3] >= P31 ? w[3] - P31 : w[3];
    *out_alpha = a;
    break;
  }
}

// Block-cooperative form of chan_mix_root_draw for kernels that already hold the root in LDS: called by ALL
// threads of the block (block-uniform control flow); the hashing runs on the first quad with the
// quad-cooperative Blake2s (0.9 us per hash instead of 2 us).  `scratch`: 28 words of LDS that do not
// overlap the root's 8 words.  Ends with a barrier and returns the drawn alpha to every thread.
// `dg` (lanes 0..7: word `lane` of the channel's digest) and `variant` are loaded by the caller - at its start, so that
// the round trip to memory is hidden behind its hashing; `dg` is updated (a kernel that makes several steps never
// reads the channel back), and the channel in memory as well (for the next launch).
LMN_D QM31 chan_mix_root_draw_block(DevChannel* ch, uint32_t& dg, uint32_t variant, const uint32_t* root,
                                    uint32_t root_stride, uint32_t* scratch, QM31* out_alpha, uint32_t* root_copy) {
  uint32_t* msg = scratch;       // 16 words: digest || root, then digest || counter
  uint32_t* wbuf = scratch + 16;  // 8 words: drawn words
  const uint32_t tid = threadIdx.x, q = tid & 3u;
  const uint32_t t_draw = variant == 0u ? 64u : 37u;
  __syncthreads();
  if (tid < 8u) {
    msg[tid] = dg;
    const uint32_t r = root[tid * root_stride];   // word k of the root at root[k * root_stride] (word-major node array)
    msg[8u + tid] = r;
    root_copy[tid] = r;
  }
  __syncthreads();
  uint32_t lo = 0u, hi = 0u;
  if (tid < 64u) b2_quad_parent(msg, q, lo, hi);
  __syncthreads();
  if (tid < 4u) {
    msg[q] = lo;
    msg[4u + q] = hi;
    ch->digest[q] = lo;
    ch->digest[4u + q] = hi;
  }
  for (uint32_t n_sent = 0;; ++n_sent) {
    if (tid >= 8u && tid < 16u) msg[tid] = tid == 8u ? n_sent : 0u;
    __syncthreads();
    if (tid < 64u) b2_quad_parent(msg, q, lo, hi, t_draw);
    if (tid < 4u) {
      wbuf[q] = lo;
      wbuf[4u + q] = hi;
    }
    __syncthreads();
    bool ok = true;
#pragma unroll
    for (int k = 0; k < 8; ++k) ok = ok && (wbuf[k] < 2u * P31);
    if (ok) {  // block-uniform
      if (tid == 0u) {
        QM31 a;
        a.a = wbuf[0] >= P31 ? wbuf[0] - P31 : wbuf[0];
        a.b = wbuf[1] >= P31 ? wbuf[1] - P31 : wbuf[1];
        a.c = wbuf[2] >= P31 ? wbuf[2] - P31 : wbuf[2];
        a.d = wbuf[3] >= P31 ? wbuf[3] - P31 : wbuf[3];
        *out_alpha = a;
        ch->n_sent = n_sent + 1u;
        scratch[24] = a.a;
        scratch[25] = a.b;
        scratch[26] = a.c;
        scratch[27] = a.d;
      }
      break;
    }
    __syncthreads();  // everyone has read wbuf before the next draw overwrites it
  }
  if (tid < 8u) dg = msg[tid];   // the digest after the mix (the draws leave it alone)
  __syncthreads();
  return QM31{scratch[24], scratch[25], scratch[26], scratch[27]};
}

constexpr int CHAN_STEP_SCRATCH = 32 + 4 * CHAN_MAX_INST;   // LDS words a ChanStep works in (layout: chan_step_* below)
constexpr int CHAN_STEP_WORDS = (int)(sizeof(ChanStep) / 4);
static_assert(sizeof(ChanStep) % 8 == 0 && CHAN_STEP_WORDS <= 1024, "a ChanStep is fetched by one word per lane of a block");
LMN_D void chan_step_run(uint32_t* scr, DevChannel* ch, uint32_t dg, uint32_t variant, uint32_t root_word, const ChanStep& st);

// Small trees / tree tops: one node per lane, one block of up to 1024 lanes, up to 10 LDS levels.
// The launch that produces the root also makes the transcript step that consumes it when `ch` is given: mix_root and
// the draw of a folding alpha (step.kind 0, the FRI layers), or one of the commitment phases' steps (ChanStep, kernels.h).
constexpr int MERKLE_SMALL_BLOCK = 1024;
template <int MODE>
LMN_KERNEL k_merkle_small(const uint32_t* __restrict__ prev, MerkleSegs sg, int ncols, uint32_t size,
                          MerkleLevels outs, int nfused, DevChannel* ch, QM31* alpha_out, uint32_t* root_copy,
                          const ChanStep* __restrict__ step, int step_kind) {
  LMN_SERIAL_KERNEL();
  LMN_SHARED uint32_t sh[MERKLE_SMALL_BLOCK * 8];
  LMN_SHARED alignas(16) uint32_t sh_step[CHAN_STEP_WORDS];
  const uint32_t i = threadIdx.x;
  // when this launch produces the root, it also runs the device-resident Fiat-Shamir step: fetch the channel now
  const bool fs = ch != nullptr && (size >> nfused) == 1u;
  uint32_t dg = 0u, variant = 0u;
  if (fs && step_kind != 1) {   // (kind 1 starts the channel from step.start)
    if (i < 8u) dg = ch->digest[i];
    variant = ch->variant;
  }
  uint32_t cur[8];
  uint32_t m[16];
  if (i < size) merkle_load_mode<MODE>(prev, sg, ncols, size, i, m);
  // the step's plan lies in page-locked host memory: fetched behind the node loads (one round trip over the link, over
  // by the time the first compression is), kept in LDS
  uint32_t step_word = 0u;
  const bool step_lane = fs && step_kind != 0 && i < (uint32_t)CHAN_STEP_WORDS;
  if (step_lane) step_word = reinterpret_cast<const uint32_t*>(step)[i];
  if (i < size) {
    merkle_hash_mode<MODE>(prev, sg, ncols, size, i, m, cur);
    store_hash(outs.p[0] + (uint64_t)i * 8, cur);
#pragma unroll
    for (int k = 0; k < 8; ++k) sh[k * MERKLE_SMALL_BLOCK + i] = cur[k];   // word-major (merkle_lds_level)
  }
  if (step_lane) sh_step[i] = step_word;
  LMN_SERIAL_KERNEL();  // the leaf compression left the wave at its low phase priority
  merkle_lds_climb<MERKLE_SMALL_BLOCK>(sh, outs, 1, nfused, size);
  if (fs && step_kind == 0) chan_mix_root_draw_block(ch, dg, variant, sh, MERKLE_SMALL_BLOCK, sh + 16, alpha_out, root_copy);
  if (fs && step_kind != 0) {
    __syncthreads();   // the root's words (the climb ends without a barrier) and the plan
    const uint32_t root_word = i < 8u ? sh[i * MERKLE_SMALL_BLOCK] : 0u;
    // sh[16 .. 16 + CHAN_STEP_SCRATCH): word 0 of dead nodes
    chan_step_run(sh + 16, ch, dg, variant, root_word, *reinterpret_cast<const ChanStep*>(sh_step));
  }
}

void launch_merkle_fused(const uint32_t* prev, const MerkleSegs& sg, int ncols, uint32_t size,
                         const MerkleLevels& outs, int sub, int nfused, lmn_stream_t s, const MerkleFold* fold) {
  if (LMN_ABLATED(1u)) return;
  if (!prev && ncols == 0) throw LmnError(-100, "merkle level with no input");
  if (nfused > MERKLE_MAX_FUSED || sub > MERKLE_MAX_SUB || sub > nfused || nfused - sub > 8 ||
      size % ((uint32_t)TPB << sub) != 0)
    throw LmnError(-100, "merkle_fused: bad arguments");
  const dim3 g(cdiv(size >> sub, TPB)), b(TPB);
  // experiment knob (docs/SWITCHES.md): bytes of unused dynamic LDS per workgroup - caps the workgroups of these issue-bound
  // launches per CU (24 KB static + pad; 5 fit by registers) so that HBM-bound launches of other proofs find room next to them
  static const size_t lds_pad = getenv("LMN_MERKLE_LDS_PAD") ? (size_t)atol(getenv("LMN_MERKLE_LDS_PAD")) : 0;
  const MerkleFold none{};
  // a null p[l] (l < sub only: the levels a lane keeps in registers) is a level the caller does not want written
  for (int l = sub; l <= nfused; ++l)
    if (!outs.p[l]) throw LmnError(-100, "merkle_fused: only the per-lane levels may be left unwritten");
  if (fold && fold->below) {
    if (prev || fold->src || ncols < 1 || fold->below_ncols < 1 || fold->below_ncols > 8)
      throw LmnError(-100, "merkle_fused: a start level over its own leaf level has columns, no stored children and <= 8 leaf columns");
    LMN_LAUNCH(k_merkle_fused<4>, g, b, lds_pad, s, prev, sg, ncols, size, outs, sub, nfused, *fold);
  } else if (fold) {
    if (prev || ncols != 4) throw LmnError(-100, "merkle_fused: a folded level is a leaf level of 4 coordinate columns");
    LMN_LAUNCH(k_merkle_fused<3>, g, b, lds_pad, s, prev, sg, ncols, size, outs, sub, nfused, *fold);
  } else if (!prev && ncols <= 16 && sg.n[0] == ncols) {
    // the leaf's zero message words are compile-time zeros of the instantiation (blake2s.h b2_compress_fresh_nz)
    if (ncols <= 4) {
      MerkleFold leaf4 = none;
      if (LMN_ABLATED(128u)) leaf4.below_ncols = -128;
      LMN_LAUNCH((k_merkle_fused<1, 4>), g, b, lds_pad, s, prev, sg, ncols, size, outs, sub, nfused, leaf4);
    }
    else if (ncols <= 8)
      LMN_LAUNCH((k_merkle_fused<1, 8>), g, b, lds_pad, s, prev, sg, ncols, size, outs, sub, nfused, none);
    else if (ncols <= 12)
      LMN_LAUNCH((k_merkle_fused<1, 12>), g, b, lds_pad, s, prev, sg, ncols, size, outs, sub, nfused, none);
    else if (ncols <= 15)
      LMN_LAUNCH((k_merkle_fused<1, 15>), g, b, lds_pad, s, prev, sg, ncols, size, outs, sub, nfused, none);
    else
      LMN_LAUNCH((k_merkle_fused<1, 16>), g, b, lds_pad, s, prev, sg, ncols, size, outs, sub, nfused, none);
  } else if (prev && ncols == 0)
    LMN_LAUNCH(k_merkle_fused<2>, g, b, lds_pad, s, prev, sg, ncols, size, outs, sub, nfused, none);
  else
    LMN_LAUNCH(k_merkle_fused<0>, g, b, lds_pad, s, prev, sg, ncols, size, outs, sub, nfused, none);
}

void launch_merkle_small(const uint32_t* prev, const MerkleSegs& sg, int ncols, uint32_t size,
                         const MerkleLevels& outs, int nfused, DevChannel* ch, QM31* alpha_out, uint32_t* root_copy,
                         lmn_stream_t s, const ChanStep* step, int step_kind) {
  if (!prev && ncols == 0) throw LmnError(-100, "merkle level with no input");
  if (size > (uint32_t)MERKLE_SMALL_BLOCK || nfused > 10) throw LmnError(-100, "merkle_small: bad arguments");
  static_assert(16 + CHAN_STEP_SCRATCH <= MERKLE_SMALL_BLOCK, "the step's scratch lies inside word 0 of the node array");
  const dim3 g(1), b(MERKLE_SMALL_BLOCK);
  if (!step) step_kind = 0;
  if (step_kind < 0 || step_kind > 3) throw LmnError(-100, "merkle_small: bad transcript step");
  if (step_kind != 0 && (!ch || (size >> nfused) != 1u)) throw LmnError(-100, "merkle_small: a transcript step needs the root and a channel");
  if (!prev && ncols <= 16 && sg.n[0] == ncols)
    LMN_LAUNCH(k_merkle_small<1>, g, b, 0, s, prev, sg, ncols, size, outs, nfused, ch, alpha_out, root_copy, step, step_kind);
  else if (prev && ncols == 0)
    LMN_LAUNCH(k_merkle_small<2>, g, b, 0, s, prev, sg, ncols, size, outs, nfused, ch, alpha_out, root_copy, step, step_kind);
  else
    LMN_LAUNCH(k_merkle_small<0>, g, b, 0, s, prev, sg, ncols, size, outs, nfused, ch, alpha_out, root_copy, step, step_kind);
}

// =============================================================================================
// Device-resident Fiat-Shamir steps for the FRI commit loop (no host round trip per layer)
// =============================================================================================
LMN_KERNEL k_chan_mix_root_draw(DevChannel* ch, const uint32_t* __restrict__ root, QM31* out_alpha,
                                uint32_t* root_copy) {
  LMN_SERIAL_KERNEL();
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  chan_mix_root_draw(ch, root, out_alpha, root_copy);
}

void launch_chan_mix_root_draw(DevChannel* ch, const uint32_t* root, QM31* out_alpha, uint32_t* root_copy,
                               lmn_stream_t s) {
  LMN_LAUNCH(k_chan_mix_root_draw, dim3(1), dim3(64), 0, s, ch, root, out_alpha, root_copy);
}

// =============================================================================================
// Device-resident Fiat-Shamir of the commitment phases (kernels.h): the transcript steps between the big kernels of a
// proof - two to eight dependent compressions each - instead of a host round trip (30 - 45 us of idle GPU each).
// Block-cooperative: every thread of the block calls these with block-uniform arguments; the compressions run on the
// first quad (b2_quad_parent: 0.9 us per hash against 2 us on one lane).  They are the last thing the launch that
// produces the tree's root does (k_merkle_small), or a launch of their own where no such launch exists (k_chan_step).
// Scratch (LDS, CHAN_STEP_SCRATCH words): msg 16 | drawn words 8 | point 8 | claimed sums 4 x CHAN_MAX_INST.
// =============================================================================================
// digest (msg[0..8)) <- H(digest || msg[8..8+n)); the caller has written msg[8..16) (zero-padded).  No barrier at the end.
LMN_D void qchan_mix(uint32_t* msg, uint32_t n_words) {
  const uint32_t tid = threadIdx.x, q = tid & 3u;
  __syncthreads();
  uint32_t lo = 0u, hi = 0u;
  if (tid < 64u) b2_quad_parent(msg, q, lo, hi, 32u + 4u * n_words);
  __syncthreads();
  if (tid < 4u) {
    msg[q] = lo;
    msg[4u + q] = hi;
  }
}
// Channel::draw_base_felts: 8 M31 from one hash of digest || counter, redrawn while a word is >= 2P; every thread gets them
LMN_D void qchan_draw_felts8(uint32_t* msg, uint32_t* wbuf, uint32_t& n_sent, uint32_t t_draw, uint32_t f[8]) {
  const uint32_t tid = threadIdx.x, q = tid & 3u;
  for (;;) {
    if (tid >= 8u && tid < 16u) msg[tid] = tid == 8u ? n_sent : 0u;
    __syncthreads();
    uint32_t lo = 0u, hi = 0u;
    if (tid < 64u) b2_quad_parent(msg, q, lo, hi, t_draw);
    if (tid < 4u) {
      wbuf[q] = lo;
      wbuf[4u + q] = hi;
    }
    n_sent += 1u;
    __syncthreads();
    bool ok = true;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      f[k] = wbuf[k];
      ok = ok && (f[k] < 2u * P31);
    }
    __syncthreads();  // everyone has read wbuf before the next draw overwrites it
    if (ok) break;    // block-uniform
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) f[k] = f[k] >= P31 ? f[k] - P31 : f[k];
}
LMN_D void qchan_store(DevChannel* ch, const uint32_t* msg, uint32_t n_sent, uint32_t variant) {
  if (threadIdx.x < 8u) ch->digest[threadIdx.x] = msg[threadIdx.x];
  if (threadIdx.x == 0u) {
    ch->n_sent = n_sent;
    ch->variant = variant;
  }
}

// kind 1: the channel starts at step.start; mix_root(root 1); one draw_felts(2) per relation element set
LMN_D void chan_step_root_elems(uint32_t* scr, DevChannel* ch, uint32_t root_word, const ChanStep& st) {
  uint32_t *msg = scr, *wbuf = scr + 16;
  const uint32_t tid = threadIdx.x;
  DevReport* rep = st.rep;
  __syncthreads();
  if (tid < 8u) {
    msg[tid] = st.start.digest[tid];
    msg[8u + tid] = root_word;
    rep->roots[0][tid] = root_word;
  }
  if (tid == 0u) rep->bad = *st.bad_word;
  qchan_mix(msg, 8u);
  uint32_t n_sent = 0u;
  const uint32_t t_draw = st.start.variant == 0u ? 64u : 37u;
  for (int i = 0; i < st.sets.n; ++i) {
    uint32_t f[8];
    qchan_draw_felts8(msg, wbuf, n_sent, t_draw, f);
    const int e = st.sets.set[i];
    if (tid == 0u && e >= 0) {
      rep->elems.z[e] = QM31{f[0], f[1], f[2], f[3]};
      rep->elems.alpha[e] = QM31{f[4], f[5], f[6], f[7]};
    }
  }
  __syncthreads();
  qchan_store(ch, msg, n_sent, st.start.variant);
}

// kind 2: mix_felts([claimed_i]) per component, mix_root(root 2), draw_felt() = the composition randomness; then the
// signed coefficient of every kernel constraint slot.  dg: lanes 0..7 hold the channel's digest (loaded by the caller)
LMN_D void chan_step_claims_root_alpha(uint32_t* scr, DevChannel* ch, uint32_t dg, uint32_t variant, uint32_t root_word,
                                       const ChanStep& st) {
  uint32_t *msg = scr, *wbuf = scr + 16, *cl = scr + 32;
  const uint32_t tid = threadIdx.x;
  const ChanCoeffPlan& plan = st.coeff;
  DevReport* rep = st.rep;
  __syncthreads();
  if (tid < (uint32_t)plan.n_inst * 4u) {   // every claimed sum in one round trip to memory
    const uint32_t w = reinterpret_cast<const uint32_t*>(plan.claimed[tid >> 2])[tid & 3u];
    cl[tid] = w;
    reinterpret_cast<uint32_t*>(&rep->claimed[tid >> 2])[tid & 3u] = w;
  }
  if (tid < 8u) msg[tid] = dg;
  for (int i = 0; i < plan.n_inst; ++i) {
    __syncthreads();
    if (tid >= 8u && tid < 16u) msg[tid] = tid < 12u ? cl[4 * i + (int)tid - 8] : 0u;
    qchan_mix(msg, 4u);
  }
  if (tid < 8u) {
    msg[8u + tid] = root_word;
    rep->roots[1][tid] = root_word;
  }
  qchan_mix(msg, 8u);
  uint32_t n_sent = 0u, f[8];
  qchan_draw_felts8(msg, wbuf, n_sent, variant == 0u ? 64u : 37u, f);
  const QM31 alpha{f[0], f[1], f[2], f[3]};
  if (tid == 0u) rep->comp_alpha = alpha;
  qchan_store(ch, msg, n_sent, variant);
  // one lane per (component, kernel slot): alpha^e by square-and-multiply
  for (int idx = (int)tid; idx < plan.n_inst * 16; idx += (int)blockDim.x) {
    const int i = idx >> 4, k = idx & 15;
    QM31 c = q_zero();
    if (k < plan.n_kernel[i] && plan.proto_index[i][k] >= 0) {
      uint32_t e = (uint32_t)(plan.n_total - 1 - ((int)plan.k0[i] + (int)plan.proto_index[i][k]));
      QM31 b = alpha;
      c = q_one();
      while (e) {
        if (e & 1u) c = q_mul(c, b);
        b = q_mul(b, b);
        e >>= 1;
      }
      if ((plan.neg[i] >> k) & 1u) c = q_neg(c);
    }
    st.coeff_out[idx] = c;
  }
}

// kind 3: mix_root(root 3), t = draw_felt(), the OODS point, the sample points and their mappings; the finished report
// goes to the host's page-locked copy
LMN_D void chan_step_root_oods(uint32_t* scr, DevChannel* ch, uint32_t dg, uint32_t variant, uint32_t root_word,
                               const ChanStep& st) {
  uint32_t *msg = scr, *wbuf = scr + 16, *sh_pt = scr + 24;
  const uint32_t tid = threadIdx.x;
  const ChanOodsPlan& plan = st.oods;
  DevReport* rep = st.rep;
  // the job table's words are fetched now (one round trip over the link, behind the hashing below) and stored at the end
  const uint32_t copy_words = st.copy_words;
  const uint32_t* copy_src = st.copy_src;
  uint32_t* copy_dst = st.copy_dst;
  const uint32_t copy_word = tid < copy_words ? copy_src[tid] : 0u;
  __syncthreads();
  if (tid < 8u) {
    msg[tid] = dg;
    msg[8u + tid] = root_word;
    rep->roots[2][tid] = root_word;
  }
  qchan_mix(msg, 8u);
  uint32_t n_sent = 0u, f[8];
  qchan_draw_felts8(msg, wbuf, n_sent, variant == 0u ? 64u : 37u, f);
  qchan_store(ch, msg, n_sent, variant);
  if (tid == 0u) {
    const QM31 tt{f[0], f[1], f[2], f[3]};
    rep->t = tt;
    const QM31 t2 = q_sqr(tt);
    const QM31 tinv = q_inv(q_add_m(t2, 1u));
    const QM31 x = q_mul(q_sub(q_one(), t2), tinv), y = q_mul(q_add(tt, tt), tinv);
    sh_pt[0] = x.a; sh_pt[1] = x.b; sh_pt[2] = x.c; sh_pt[3] = x.d;
    sh_pt[4] = y.a; sh_pt[5] = y.b; sh_pt[6] = y.c; sh_pt[7] = y.d;
  }
  __syncthreads();
  const QM31 ox{sh_pt[0], sh_pt[1], sh_pt[2], sh_pt[3]}, oy{sh_pt[4], sh_pt[5], sh_pt[6], sh_pt[7]};
  for (int p = (int)tid; p < plan.n_points; p += (int)blockDim.x) {
    QM31 px = ox, py = oy;
    if (p > 0) {   // secure point + base point (host.h qpt_add_m)
      const uint32_t bx = plan.step_x[p], by = plan.step_y[p];
      px = q_sub(q_mul_m(ox, bx), q_mul_m(oy, by));
      py = q_add(q_mul_m(ox, by), q_mul_m(oy, bx));
    }
    QM31* mp = st.maps_out + (size_t)p * plan.n_maps;
    mp[0] = py;
    mp[1] = px;
    QM31 cur = px;
    for (int k = 2; k < plan.n_maps; ++k) {
      cur = q_sub_m(q_add(q_sqr(cur), q_sqr(cur)), 1u);
      mp[k] = cur;
    }
  }
  // the report is complete: the last of the device-resident steps writes the host's copy itself (page-locked memory;
  // this launch's own words of it are visible to the block since the barrier above)
  const uint32_t* rw = reinterpret_cast<const uint32_t*>(rep);
  for (uint32_t k = tid; k < (uint32_t)(sizeof(DevReport) / 4); k += blockDim.x) st.rep_host[k] = rw[k];
  if (tid < copy_words) copy_dst[tid] = copy_word;
  for (uint32_t k = tid + blockDim.x; k < copy_words; k += blockDim.x) copy_dst[k] = copy_src[k];
}

// =============================================================================================
// The step in front of the FRI quotient kernels (kernels.h QuotPrepPlan): mix_felts(sampled values), alpha = draw_felt(),
// the quotient tables of every LDE size.  One workgroup; block-uniform control flow.
// =============================================================================================
constexpr int QUOT_PREP_PLAN_WORDS = (int)(sizeof(QuotPrepPlan) / 4);
static_assert(sizeof(QuotPrepPlan) % 8 == 0, "the plan is fetched word by word");
LMN_KERNEL k_quot_prepare(DevChannel* ch, const QuotPrepPlan* __restrict__ plan_g) {
  LMN_SERIAL_KERNEL();
  LMN_SHARED alignas(16) uint32_t sh_plan[QUOT_PREP_PLAN_WORDS];
  LMN_SHARED uint32_t W[8 + 4 * QUOT_PREP_MAX_SAMPLES + 16];   // digest || values, zero-padded to whole blocks
  LMN_SHARED uint32_t scr[32];                                   // msg 16 | drawn words 8 | alpha 4
  LMN_SHARED QM31 sa[QUOT_MAX_ENTRIES], sb[QUOT_MAX_ENTRIES];
  const uint32_t tid = threadIdx.x, q = tid & 3u;
  for (uint32_t k = tid; k < (uint32_t)QUOT_PREP_PLAN_WORDS; k += blockDim.x) sh_plan[k] = reinterpret_cast<const uint32_t*>(plan_g)[k];
  const uint32_t dgw = tid < 8u ? ch->digest[tid] : 0u;
  const uint32_t variant = ch->variant;
  __syncthreads();
  const QuotPrepPlan& plan = *reinterpret_cast<const QuotPrepPlan*>(sh_plan);
  const uint32_t n_val_words = 4u * (uint32_t)plan.n_samples, total_words = 8u + n_val_words;
  const uint32_t n_blocks = (total_words + 15u) / 16u;
  if (tid < 8u) W[tid] = dgw;
  for (uint32_t k = tid; k < n_blocks * 16u - 8u; k += blockDim.x) {
    uint32_t w = 0u;
    if (k < n_val_words) {
      w = reinterpret_cast<const uint32_t*>(plan.vals)[k];
      reinterpret_cast<uint32_t*>(plan.vals_host)[k] = w;   // the host's copy of the sampled values (page-locked memory)
    }
    W[8u + k] = w;
  }
  __syncthreads();
  // ---- Channel::mix_felts: digest <- Blake2s(digest || values)
  uint32_t* msg = scr;
#ifdef LMN_EMU
  if (tid == 0u) {
    uint32_t h[8];
    b2_init(h);
    for (uint32_t b = 0; b < n_blocks; ++b) {
      const bool last = b + 1u == n_blocks;
      b2_compress(h, W + 16u * b, last ? 4u * total_words : 64u * (b + 1u), last ? 0xffffffffu : 0u);
    }
    for (int k = 0; k < 8; ++k) msg[k] = h[k];
  }
#else
  if (tid < 64u) {
    uint32_t hl = q == 0 ? (0x6A09E667u ^ 0x01010020u) : q == 1 ? 0xBB67AE85u : q == 2 ? 0x3C6EF372u : 0xA54FF53Au;
    uint32_t hh = q == 0 ? 0x510E527Fu : q == 1 ? 0x9B05688Cu : q == 2 ? 0x1F83D9ABu : 0x5BE0CD19u;
    for (uint32_t b = 0; b < n_blocks; ++b) {
      const bool last = b + 1u == n_blocks;
      b2_quad_compress(W + 16u * b, q, hl, hh, last ? 4u * total_words : 64u * (b + 1u), last ? 0xffffffffu : 0u);
    }
    if (tid < 4u) {
      msg[q] = hl;
      msg[4u + q] = hh;
    }
  }
#endif
  __syncthreads();
  // ---- alpha = draw_felt(); the channel goes on into the FRI commit loop
  uint32_t n_sent = 0u, f[8];
  qchan_draw_felts8(msg, scr + 16, n_sent, variant == 0u ? 64u : 37u, f);
  const QM31 alpha{f[0], f[1], f[2], f[3]};
  qchan_store(ch, msg, n_sent, variant);
  if (tid == 0u) plan.vals_host[plan.n_samples] = alpha;
  // ---- one lane per (column, sample): alpha^(k+1) times the line through (p.y, v) and (conj p.y, conj v)
  for (uint32_t e = tid; e < (uint32_t)plan.n_entries; e += blockDim.x) {
    const QuotPrepEntry en = plan.entries[e];
    const QuotPrepSize& sz = plan.size[en.size];
    const uint32_t local = e - sz.first_entry;
    const uint32_t k = local - (uint32_t)sz.batch_start[en.batch];
    const QM31 power = q_pow(alpha, (uint64_t)k + 1u);
    const uint32_t* vw = W + 8u + 4u * en.sample;
    const QM31 val{vw[0], vw[1], vw[2], vw[3]};
    const QM31 py = plan.maps[(size_t)sz.point[en.batch] * (size_t)plan.n_maps];
    const QM31 la = q_sub(q_conj(val), val);
    const QM31 lc = q_sub(q_conj(py), py);
    const QM31 lbb = q_sub(q_mul(val, lc), q_mul(la, py));
    sz.entries_out[local] = QuotEntry{en.col, q_mul(power, lc)};
    sa[e] = q_mul(power, la);
    sb[e] = q_mul(power, lbb);
  }
  __syncthreads();
  // ---- one lane per (size, batch): the batch's sums, its power of alpha, its point
  if (tid < (uint32_t)plan.n_sizes * (uint32_t)QUOT_MAX_BATCH) {
    const QuotPrepSize& sz = plan.size[tid / QUOT_MAX_BATCH];
    const int b = (int)(tid % QUOT_MAX_BATCH);
    if (b < sz.n_batch) {
      QM31 A = q_zero(), B = q_zero();
      for (int i = sz.batch_start[b]; i < sz.batch_start[b + 1]; ++i) {
        A = q_add(A, sa[sz.first_entry + (uint32_t)i]);
        B = q_add(B, sb[sz.first_entry + (uint32_t)i]);
      }
      const QM31* mp = plan.maps + (size_t)sz.point[b] * (size_t)plan.n_maps;
      const QM31 py = mp[0], px = mp[1];
      QuotDev* d = sz.dev_out;
      d->A[b] = A;
      d->B[b] = B;
      d->batch_coeff[b] = q_pow(alpha, (uint64_t)(sz.batch_start[b + 1] - sz.batch_start[b]));
      d->prx[b] = CM31{px.a, px.b};
      d->pix[b] = CM31{px.c, px.d};
      d->pry[b] = CM31{py.a, py.b};
      d->piy[b] = CM31{py.c, py.d};
    }
  }
  for (uint32_t k = tid; k < plan.copy_words; k += blockDim.x) plan.copy_dst[k] = plan.copy_src[k];
}
void launch_quot_prepare(DevChannel* ch, const QuotPrepPlan* plan, lmn_stream_t s) {
  if (!ch || !plan) throw LmnError(-100, "quot_prepare: no channel / plan");
  LMN_LAUNCH(k_quot_prepare, dim3(1), dim3(TPB), 0, s, ch, plan);
}

// root_word: lanes 0..7 hold the root's words; dg / variant: the channel as the caller fetched it at its start (kind 1
// starts from step.start instead)
LMN_D void chan_step_run(uint32_t* scr, DevChannel* ch, uint32_t dg, uint32_t variant, uint32_t root_word, const ChanStep& st) {
  if (st.kind == 1)
    chan_step_root_elems(scr, ch, root_word, st);
  else if (st.kind == 2)
    chan_step_claims_root_alpha(scr, ch, dg, variant, root_word, st);
  else if (st.kind == 3)
    chan_step_root_oods(scr, ch, dg, variant, root_word, st);
}

// a step as a launch of its own (trees whose root is not produced by k_merkle_small)
LMN_KERNEL k_chan_step(DevChannel* ch, const ChanStep* __restrict__ step, int step_kind, const uint32_t* __restrict__ root) {
  LMN_SERIAL_KERNEL();
  LMN_SHARED uint32_t scr[CHAN_STEP_SCRATCH];
  LMN_SHARED alignas(16) uint32_t sh_step[CHAN_STEP_WORDS];
  for (uint32_t k = threadIdx.x; k < (uint32_t)CHAN_STEP_WORDS; k += blockDim.x) sh_step[k] = reinterpret_cast<const uint32_t*>(step)[k];
  uint32_t dg = 0u, variant = 0u, root_word = 0u;
  if (threadIdx.x < 8u) root_word = root[threadIdx.x];
  if (step_kind != 1) {
    if (threadIdx.x < 8u) dg = ch->digest[threadIdx.x];
    variant = ch->variant;
  }
  __syncthreads();
  chan_step_run(scr, ch, dg, variant, root_word, *reinterpret_cast<const ChanStep*>(sh_step));
}
void check_chan_step(const ChanStep& st) {
  if (st.kind < 1 || st.kind > 3) throw LmnError(-100, "chan_step: bad kind");
  if (st.kind == 1 && (st.sets.n < 1 || st.sets.n > CHAN_N_ELEMS)) throw LmnError(-100, "chan_step: bad draw count");
  if (st.kind == 2 && (st.coeff.n_inst < 1 || st.coeff.n_inst > CHAN_MAX_INST)) throw LmnError(-100, "chan_step: bad component count");
  if (st.kind == 3 && (st.oods.n_points < 1 || st.oods.n_points > CHAN_MAX_POINTS || st.oods.n_maps < 2))
    throw LmnError(-100, "chan_step: bad sample plan");
}
// `step`: device-visible (page-locked host memory is: read once)
void launch_chan_step(DevChannel* ch, const ChanStep* step, int step_kind, const uint32_t* root, lmn_stream_t s) {
  if (!step || step_kind < 1 || step_kind > 3) throw LmnError(-100, "chan_step: bad kind");
  LMN_LAUNCH(k_chan_step, dim3(1), dim3(TPB), 0, s, ch, step, step_kind, root);
}

// =============================================================================================
// FRI tail: all layers of size <= 1024 in ONE single-block launch: per layer Merkle-commit the
// line evaluation (LDS tree), mix the root into the device-resident channel, draw the folding
// alpha, fold.  Every layer's evaluations and tree levels still go to HBM for decommitment.
// =============================================================================================
LMN_KERNEL k_fri_tail(DevChannel* ch, const FriTailLayer* __restrict__ layers, int n_layers, int first_log,
                      QM31* alphas_out, uint32_t* roots_out, FriTailIo pre) {
  LMN_SERIAL_KERNEL();
  LMN_SHARED uint32_t sh[MERKLE_SMALL_BLOCK * 8];
  LMN_SHARED uint32_t shv[MERKLE_SMALL_BLOCK * 4];   // the layer's values, coordinate-major: the fold reads its pair here
  const uint32_t i = threadIdx.x;
  // the channel is read once and the layers' values once: a lane keeps value i of the current layer (its fold output is
  // the next layer's value i), so that no step of the chain waits for memory
  uint32_t dg = i < 8u ? ch->digest[i] : 0u;
  const uint32_t variant = ch->variant;
  QM31 mine = q_zero();
  {
    const uint32_t size0 = 1u << first_log;
    if (pre.src) {   // the first layer is the fold of the layer before it (same arithmetic as k_fold / MerkleFold)
      if (i < size0) {
        const uint64_t L = 2ull * size0;
        const uint32_t* __restrict__ sp = pre.src + 2ull * i;
        const QM31 a{sp[0], sp[L], sp[2 * L], sp[3 * L]};
        const QM31 b{sp[1], sp[L + 1], sp[2 * L + 1], sp[3 * L + 1]};
        mine = q_add(q_add(a, b), q_mul(*pre.alpha, q_mul_m(q_sub(a, b), pre.itw[i])));
        uint32_t* v0 = const_cast<uint32_t*>(layers[0].vals);
        v0[i] = mine.a;
        v0[size0 + i] = mine.b;
        v0[2 * size0 + i] = mine.c;
        v0[3 * size0 + i] = mine.d;
      }
    } else {
      const uint32_t* v0 = layers[0].vals;
      if (i < size0) mine = QM31{v0[i], v0[size0 + i], v0[2 * size0 + i], v0[3 * size0 + i]};
    }
  }
  for (int li = 0; li < n_layers; ++li) {
    const FriTailLayer& ly = layers[li];   // (read through the scalar cache where needed: a copy would be indexed in scratch memory)
    const int L = first_log - li;
    const uint32_t size = 1u << L;
    uint32_t cur[8];
    if (i < size) {
      uint32_t m[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) m[k] = 0u;
      m[0] = mine.a;
      m[1] = mine.b;
      m[2] = mine.c;
      m[3] = mine.d;
      shv[i] = mine.a;
      shv[MERKLE_SMALL_BLOCK + i] = mine.b;
      shv[2 * MERKLE_SMALL_BLOCK + i] = mine.c;
      shv[3 * MERKLE_SMALL_BLOCK + i] = mine.d;
      b2_compress_fresh_nz<4>(cur, m, 16u);
      store_hash(ly.merkle[L] + (uint64_t)i * 8, cur);
#pragma unroll
      for (int k = 0; k < 8; ++k) sh[k * MERKLE_SMALL_BLOCK + i] = cur[k];   // word-major (merkle_lds_level)
    }
    for (int l = L - 1; l >= 0; --l) merkle_lds_level<MERKLE_SMALL_BLOCK>(sh, ly.merkle[l], 0u, 1u << l);
    const QM31 alpha = chan_mix_root_draw_block(ch, dg, variant, sh, MERKLE_SMALL_BLOCK, sh + 16, &alphas_out[li],
                                                roots_out + li * 8);
    const uint32_t n = size >> 1;
    if (i < n) {
      const QM31 a{shv[2 * i], shv[MERKLE_SMALL_BLOCK + 2 * i], shv[2 * MERKLE_SMALL_BLOCK + 2 * i],
                   shv[3 * MERKLE_SMALL_BLOCK + 2 * i]};
      const QM31 b{shv[2 * i + 1], shv[MERKLE_SMALL_BLOCK + 2 * i + 1], shv[2 * MERKLE_SMALL_BLOCK + 2 * i + 1],
                   shv[3 * MERKLE_SMALL_BLOCK + 2 * i + 1]};
      QM31 f0 = q_add(a, b);
      QM31 f1 = q_mul_m(q_sub(a, b), ly.itw[i]);
      QM31 r = q_add(f0, q_mul(alpha, f1));
      ly.next[i] = r.a;
      ly.next[n + i] = r.b;
      ly.next[2 * n + i] = r.c;
      ly.next[3 * n + i] = r.d;
      mine = r;
    }
    __syncthreads();
  }
  if (pre.out_mirror)   // (this launch's own words of the block are visible to the block since the barrier above)
    for (uint32_t k = i; k < pre.out_words; k += blockDim.x) pre.out_mirror[k] = pre.out_block[k];
}

void launch_fri_tail(DevChannel* ch, const FriTailLayer* layers, int n_layers, int first_log, QM31* alphas_out,
                     uint32_t* roots_out, lmn_stream_t s, const FriTailIo* pre) {
  if (first_log > 10 || n_layers < 1 || n_layers > first_log) throw LmnError(-100, "fri_tail: bad arguments");
  const FriTailIo p = pre ? *pre : FriTailIo{nullptr, nullptr, nullptr, nullptr, nullptr, 0u};
  LMN_LAUNCH(k_fri_tail, dim3(1), dim3(MERKLE_SMALL_BLOCK), 0, s, ch, layers, n_layers, first_log, alphas_out,
             roots_out, p);
}

// =============================================================================================
// gather
// =============================================================================================
LMN_KERNEL k_gather(const uint32_t* __restrict__ arena, const GatherEntry* __restrict__ entries, uint32_t n,
                    const MerkleRecompute* __restrict__ jobs, uint32_t n_jobs, uint32_t* __restrict__ out) {
  LMN_SERIAL_KERNEL();
  uint32_t e = blockIdx.x;
  if (e < n) {
    GatherEntry g = entries[e];
    for (uint32_t k = threadIdx.x; k < g.len; k += blockDim.x) out[g.dst_off + k] = arena[g.src_off + k];
    return;
  }
  e -= n;
  if (e >= n_jobs) return;
  // A tree node the fused launch kept in registers only: lane q of a quad hashes start node (node << depth) + q, the
  // quad reduces the 2^depth hashes pairwise (every quad of the block does the same; block-uniform control flow).
  const MerkleRecompute j = jobs[e];
  const uint32_t q = threadIdx.x & 3u;
  uint32_t h[8];
  const uint32_t i0 = (j.node << j.depth) + (q & ((1u << j.depth) - 1u));
  if (j.below) {
    uint32_t mb[16];
    merkle_load_below(j.below, j.below_ncols, j.size, i0, mb);
    merkle_hash_below(j.sg, j.ncols, j.size, j.below_ncols, i0, mb, h);
  } else {
    merkle_hash_start(j.prev, j.sg, j.ncols, j.size, i0, h);
  }
  for (int s = 0; s < j.depth; ++s) {
    const bool hi = ((q >> s) & 1u) != 0u;
    uint32_t m[16];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const uint32_t other = lmn_shfl_xor(h[k], 1 << s);
      m[k] = hi ? other : h[k];
      m[8 + k] = hi ? h[k] : other;
    }
    b2_compress_fresh(h, m, 64u);
  }
  if (threadIdx.x == 0) {
#pragma unroll
    for (int k = 0; k < 8; ++k) out[j.dst_off + k] = h[k];
  }
}

void launch_gather(const uint32_t* arena, const GatherEntry* entries, uint32_t n_entries, const MerkleRecompute* jobs,
                   uint32_t n_jobs, uint32_t* out, lmn_stream_t s) {
  if (n_entries + n_jobs == 0) return;
  LMN_LAUNCH(k_gather, dim3(n_entries + n_jobs), dim3(64), 0, s, arena, entries, n_entries, jobs, n_jobs, out);
}

}  // namespace lmn

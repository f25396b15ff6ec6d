// Host-side protocol pieces of the prover: circle-group index math, Fiat-Shamir channel, proof
// container + bincode writer.  These replace the stwo types `prove()` touches on the host
// (/root/reference/crates/prover/src/prover.rs:36-46,177,186,296,312).
#pragma once
#include <array>
#include <map>
#include <vector>

#include "../../include/luminair_hip.h"
#include "blake2s.h"
#include "field.h"

namespace lmn {

// ------------------------------------------------------------------------------------ circle group
struct Pt {
  uint32_t x, y;
};
constexpr Pt CIRCLE_GEN{2u, 1268011823u};
constexpr int LOG_ORDER = 31;

inline Pt pt_add(Pt a, Pt b) {
  return {m_sub(m_mul(a.x, b.x), m_mul(a.y, b.y)), m_add(m_mul(a.x, b.y), m_mul(a.y, b.x))};
}
inline Pt pt_double(Pt a) { return {m_sub(m_dbl(m_sqr(a.x)), 1u), m_dbl(m_mul(a.x, a.y))}; }
inline Pt pt_of_index(uint32_t k) {  // k taken mod 2^31
  k &= 0x7fffffffu;
  Pt res{1u, 0u}, cur = CIRCLE_GEN;
  while (k) {
    if (k & 1u) res = pt_add(res, cur);
    cur = pt_double(cur);
    k >>= 1;
  }
  return res;
}
inline uint32_t subgroup_gen_index(int log_size) { return 1u << (LOG_ORDER - log_size); }
inline uint32_t bit_reverse(uint32_t i, int bits) {
  uint32_t r = 0;
  for (int k = 0; k < bits; ++k) r |= ((i >> k) & 1u) << (bits - 1 - k);
  return r;
}

// point of CanonicCoset(log).circle_domain() stored at index s (bit-reversed order)
inline Pt domain_point(int log, uint32_t s) {
  uint32_t idx = bit_reverse(s, log);
  uint32_t half = 1u << (log - 1);
  uint32_t init = 1u << (30 - log);
  uint32_t step = log >= 2 ? (1u << (32 - log)) : 0u;
  if (idx < half) return pt_of_index(init + idx * step);
  Pt p = pt_of_index(init + (idx - half) * step);
  return {p.x, m_neg(p.y)};
}
// x-coordinate of LineDomain(Coset::half_odds(log)) at bit-reversed index i (FRI line layers)
inline uint32_t line_domain_x(int log, uint32_t i) {
  uint32_t init = 1u << (31 - (log + 2)), step = log >= 1 ? (1u << (31 - log)) : 0u;
  return pt_of_index(init + bit_reverse(i, log) * step).x;
}

struct QPt {
  QM31 x, y;
};
inline QPt qpt_add_m(QPt a, Pt b) {  // secure point + base point
  return {q_sub(q_mul_m(a.x, b.x), q_mul_m(a.y, b.y)), q_add(q_mul_m(a.x, b.y), q_mul_m(a.y, b.x))};
}

// ------------------------------------------------------------------------------------ channel
inline Hash32 b2_hash_bytes(const uint8_t* data, size_t n) {
  uint32_t h[8];
  b2_init(h);
  size_t nblocks = n == 0 ? 1 : (n + 63) / 64;
  for (size_t b = 0; b < nblocks; ++b) {
    uint8_t blk[64] = {0};
    size_t take = (b + 1 == nblocks) ? n - 64 * b : 64;
    memcpy(blk, data + 64 * b, take);
    uint32_t m[16];
    for (int i = 0; i < 16; ++i)
      m[i] = (uint32_t)blk[4 * i] | ((uint32_t)blk[4 * i + 1] << 8) | ((uint32_t)blk[4 * i + 2] << 16) |
             ((uint32_t)blk[4 * i + 3] << 24);
    bool last = b + 1 == nblocks;
    b2_compress(h, m, last ? (uint32_t)n : (uint32_t)(64 * (b + 1)), last ? 0xffffffffu : 0u);
  }
  Hash32 r;
  for (int i = 0; i < 8; ++i) r.w[i] = h[i];
  return r;
}

// Blake2sChannel.  `flags` = lmn_config.protocol_variant (LMN_PV_* bits, luminair_hip.h): all
// transcript bits clear = the KAT-pinned encodings; each bit switches one encoding to what stwo at the pinned rev is
// believed to use (un-vendored: unpinned).
class Channel {
 public:
  explicit Channel(uint32_t flags) : variant_(flags) { memset(digest_.w, 0, sizeof digest_.w); }
  uint32_t flags() const { return variant_; }
  const Hash32& digest() const { return digest_; }
  void set_digest(const Hash32& d) { update(d); }

  void mix_root(const Hash32& root) {
    uint32_t w[16];
    memcpy(w, digest_.w, 32);
    memcpy(w + 8, root.w, 32);
    update(b2_hash_words(w, 16));
  }
  void mix_felts(const std::vector<QM31>& felts) {
    std::vector<uint32_t> w(8 + 4 * felts.size());
    memcpy(w.data(), digest_.w, 32);
    for (size_t i = 0; i < felts.size(); ++i) {
      w[8 + 4 * i] = felts[i].a;
      w[9 + 4 * i] = felts[i].b;
      w[10 + 4 * i] = felts[i].c;
      w[11 + 4 * i] = felts[i].d;
    }
    update(b2_hash_words(w.data(), w.size()));
  }
  void mix_u64(uint64_t v) {
    uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
    if (!(variant_ & LMN_PV_MIX_U64_HASHED)) {
      uint32_t h[8], m[16] = {0};
      memcpy(h, digest_.w, 32);
      m[0] = lo;
      m[1] = hi;
      b2_compress(h, m, 0u, 0u);  // bare compression function, no finalisation flag
      Hash32 r;
      memcpy(r.w, h, 32);
      update(r);
    } else {
      uint32_t w[10];
      memcpy(w, digest_.w, 32);
      w[8] = lo;
      w[9] = hi;
      update(b2_hash_words(w, 10));
    }
  }
  Hash32 draw_random_words() {
    Hash32 r;
    if (!(variant_ & LMN_PV_DRAW_CTR_U32)) {
      uint32_t w[16] = {0};
      memcpy(w, digest_.w, 32);
      w[8] = n_sent_;  // u64 counter zero-padded to 32 bytes
      r = b2_hash_words(w, 16);
    } else {
      uint8_t buf[37];
      memcpy(buf, digest_.w, 32);
      buf[32] = (uint8_t)n_sent_;
      buf[33] = (uint8_t)(n_sent_ >> 8);
      buf[34] = (uint8_t)(n_sent_ >> 16);
      buf[35] = (uint8_t)(n_sent_ >> 24);
      buf[36] = 0;
      r = b2_hash_bytes(buf, 37);
    }
    ++n_sent_;
    return r;
  }
  std::array<uint32_t, 8> draw_base_felts() {
    for (;;) {
      Hash32 r = draw_random_words();
      bool ok = true;
      for (int i = 0; i < 8; ++i) ok = ok && r.w[i] < 2u * P31;
      if (!ok) continue;
      std::array<uint32_t, 8> f;
      for (int i = 0; i < 8; ++i) f[i] = r.w[i] >= P31 ? r.w[i] - P31 : r.w[i];
      return f;
    }
  }
  QM31 draw_felt() {
    auto f = draw_base_felts();
    return QM31{f[0], f[1], f[2], f[3]};
  }
  std::vector<QM31> draw_felts(size_t n) {
    std::vector<QM31> out;
    std::vector<uint32_t> pool;
    size_t pos = 0;
    while (out.size() < n) {
      if (pool.size() - pos < 4) {
        auto f = draw_base_felts();
        pool.insert(pool.end(), f.begin(), f.end());
      }
      out.push_back(QM31{pool[pos], pool[pos + 1], pool[pos + 2], pool[pos + 3]});
      pos += 4;
    }
    return out;
  }
  uint32_t trailing_zeros() const {
    for (int w = 0; w < 4; ++w)
      if (digest_.w[w]) return 32 * w + (uint32_t)__builtin_ctz(digest_.w[w]);
    return 128;
  }
  // Proof of work.  KAT form: the digest after mix_u64(nonce) must end in >= pow_bits zero bits.  POW_PREFIXED form
  // (stwo `Channel::verify_pow_nonce` at the pinned rev, from memory): blake2s(blake2s(0x12345678 LE || 12 zero bytes ||
  // digest || pow_bits LE) || nonce LE) must; the nonce is mixed afterwards either way.
  Hash32 pow_prefixed_digest(uint32_t pow_bits) const {
    uint32_t w[13] = {0x12345678u, 0u, 0u, 0u};
    memcpy(w + 4, digest_.w, 32);
    w[12] = pow_bits;
    return b2_hash_words(w, 13);
  }
  static uint32_t trailing_zeros_of(const Hash32& d) {
    for (int w = 0; w < 4; ++w)
      if (d.w[w]) return 32 * w + (uint32_t)__builtin_ctz(d.w[w]);
    return 128;
  }
  bool verify_pow_nonce(uint32_t pow_bits, uint64_t nonce) const {
    if (variant_ & LMN_PV_POW_PREFIXED) return pow_check_prefixed(pow_prefixed_digest(pow_bits), pow_bits, nonce);
    Channel c = *this;
    c.mix_u64(nonce);
    return c.trailing_zeros() >= pow_bits;
  }
  uint64_t grind(uint32_t pow_bits) const {
    if (variant_ & LMN_PV_POW_PREFIXED) {
      const Hash32 pre = pow_prefixed_digest(pow_bits);
      for (uint64_t nonce = 0;; ++nonce)
        if (pow_check_prefixed(pre, pow_bits, nonce)) return nonce;
    }
    for (uint64_t nonce = 0;; ++nonce) {
      Channel c = *this;
      c.mix_u64(nonce);
      if (c.trailing_zeros() >= pow_bits) return nonce;
    }
  }

 private:
  static bool pow_check_prefixed(const Hash32& pre, uint32_t pow_bits, uint64_t nonce) {
    uint32_t w[10];
    memcpy(w, pre.w, 32);
    w[8] = (uint32_t)nonce;
    w[9] = (uint32_t)(nonce >> 32);
    return trailing_zeros_of(b2_hash_words(w, 10)) >= pow_bits;
  }
  void update(const Hash32& d) {
    digest_ = d;
    n_sent_ = 0;
  }
  Hash32 digest_;
  uint32_t n_sent_ = 0;
  uint32_t variant_;
};

// ------------------------------------------------------------------------------------ proof container
struct Decommitment {
  std::vector<Hash32> hash_witness;
  std::vector<uint32_t> column_witness;
};
struct FriLayerProof {
  std::vector<QM31> fri_witness;
  Decommitment decommitment;
  Hash32 commitment;
};
struct Proof {
  std::vector<int> claim;                         // -1 = None, else log_size, per slot
  std::vector<std::pair<bool, QM31>> interaction_claim;
  uint32_t pow_bits, log_blowup, log_last_layer;
  uint64_t n_queries;
  std::vector<Hash32> commitments;
  std::vector<std::vector<std::vector<QM31>>> sampled_values;
  std::vector<Decommitment> decommitments;
  std::vector<std::vector<uint32_t>> queried_values;
  uint64_t proof_of_work;
  FriLayerProof first_layer;
  std::vector<FriLayerProof> inner_layers;
  std::vector<QM31> last_layer_coeffs;
  uint32_t last_layer_log_size;
};

// bincode 1.3: little-endian, u64 lengths, Option = 1 tag byte
// (the host is little-endian like the format: words and arrays of words are appended as they lie in memory - this runs
// behind the proof's last wait, i.e. on its latency)
static_assert(__BYTE_ORDER__ == __ORDER_LITTLE_ENDIAN__, "BinWriter appends host words as little-endian bytes");
static_assert(sizeof(QM31) == 16 && sizeof(Hash32) == 32, "QM31 / Hash32 arrays are appended as they lie in memory");
class BinWriter {
 public:
  std::vector<uint8_t> buf;
  void raw(const void* p, size_t n) {
    const uint8_t* b = static_cast<const uint8_t*>(p);
    buf.insert(buf.end(), b, b + n);
  }
  void u8(uint8_t v) { buf.push_back(v); }
  void u32(uint32_t v) { raw(&v, 4); }
  void u64(uint64_t v) { raw(&v, 8); }
  void q(const QM31& f) { raw(&f, 16); }
  void hash(const Hash32& h) { raw(h.w, 32); }
  void qs(const std::vector<QM31>& v) {
    u64(v.size());
    raw(v.data(), v.size() * 16);
  }
  void words(const std::vector<uint32_t>& v) {
    u64(v.size());
    raw(v.data(), v.size() * 4);
  }
  void decommit(const Decommitment& d) {
    u64(d.hash_witness.size());
    raw(d.hash_witness.data(), d.hash_witness.size() * 32);
    words(d.column_witness);
  }
  void layer(const FriLayerProof& l) {
    qs(l.fri_witness);
    decommit(l.decommitment);
    hash(l.commitment);
  }
  static size_t bytes_of(const Decommitment& d) { return 16 + d.hash_witness.size() * 32 + d.column_witness.size() * 4; }
  static size_t bytes_of(const FriLayerProof& l) { return 8 + l.fri_witness.size() * 16 + bytes_of(l.decommitment) + 32; }
};

inline std::vector<uint8_t> proof_to_bincode(const Proof& p) {
  BinWriter w;
  {
    size_t n = 256 + 17 * (p.claim.size() + p.interaction_claim.size()) + 32 * p.commitments.size() + 16 * p.last_layer_coeffs.size();
    for (auto& t : p.sampled_values) {
      n += 8;
      for (auto& col : t) n += 8 + 16 * col.size();
    }
    for (auto& d : p.decommitments) n += BinWriter::bytes_of(d);
    for (auto& t : p.queried_values) n += 8 + 4 * t.size();
    n += BinWriter::bytes_of(p.first_layer);
    for (auto& l : p.inner_layers) n += BinWriter::bytes_of(l);
    w.buf.reserve(n);
  }
  for (int c : p.claim) {
    if (c < 0) {
      w.u8(0);
    } else {
      w.u8(1);
      w.u32((uint32_t)c);
    }
  }
  for (auto& c : p.interaction_claim) {
    if (!c.first) {
      w.u8(0);
    } else {
      w.u8(1);
      w.q(c.second);
    }
  }
  w.u32(p.pow_bits);
  w.u32(p.log_blowup);
  w.u32(p.log_last_layer);
  w.u64(p.n_queries);
  w.u64(p.commitments.size());
  for (auto& c : p.commitments) w.hash(c);
  w.u64(p.sampled_values.size());
  for (auto& t : p.sampled_values) {
    w.u64(t.size());
    for (auto& col : t) w.qs(col);
  }
  w.u64(p.decommitments.size());
  for (auto& d : p.decommitments) w.decommit(d);
  w.u64(p.queried_values.size());
  for (auto& t : p.queried_values) w.words(t);
  w.u64(p.proof_of_work);
  w.layer(p.first_layer);
  w.u64(p.inner_layers.size());
  for (auto& l : p.inner_layers) w.layer(l);
  w.qs(p.last_layer_coeffs);
  w.u32(p.last_layer_log_size);
  return std::move(w.buf);
}

}  // namespace lmn

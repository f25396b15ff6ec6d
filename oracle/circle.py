"""Circle group over M31, cosets, canonic domains, bit-reversal (oracle; test infrastructure only).

Restates stwo `core/circle.rs`, `core/poly/circle/{canonic,domain}.rs`, `core/poly/line.rs`,
`core/utils.rs` @0790eba4 (absent from /root/reference) per SURVEY.md Appendix A.2:
  generator G = (2, 1268011823) of order 2^31; index k <-> k*G;
  CanonicCoset(n).circle_domain() = half coset {(2^(30-n) + j*2^(32-n))*G : j < 2^(n-1)}
  followed by its conjugates; columns are stored bit-reversed.
The reference call sites that fix these domains: `crates/prover/src/prover.rs:38-42`
(`CanonicCoset::new(max_log_size + log_blowup + 2).circle_domain().half_coset`).
"""
from __future__ import annotations

import numpy as np

from .field import P, U64, QM31, m_add, m_mul, m_sub, m_neg, m_inv

GEN = (2, 1268011823)
LOG_ORDER = 31
ORDER = 1 << LOG_ORDER


# ----------------------------------------------------------------------------- scalar points
def p_add(a, b):
    return ((a[0] * b[0] - a[1] * b[1]) % P, (a[0] * b[1] + a[1] * b[0]) % P)


def p_double(a):
    return ((2 * a[0] * a[0] - 1) % P, (2 * a[0] * a[1]) % P)


def p_conj(a):
    return (a[0], (-a[1]) % P)


def p_neg(a):
    return p_conj(a)  # group inverse on the circle is conjugation


def point_of_index(k: int):
    k %= ORDER
    res, cur = (1, 0), GEN
    while k:
        if k & 1:
            res = p_add(res, cur)
        cur = p_double(cur)
        k >>= 1
    return res


def subgroup_gen_index(log_size: int) -> int:
    return 1 << (LOG_ORDER - log_size)


# secure-field points (for the OODS point)
def qp_add(a, b):
    return (a[0] * b[0] - a[1] * b[1], a[0] * b[1] + a[1] * b[0])


def qp_from_m(pt):
    return (QM31(pt[0]), QM31(pt[1]))


# ----------------------------------------------------------------------------- cosets / domains
class Coset:
    """{ (initial + j*step) * G : j < 2^log_size } by point index."""

    def __init__(self, initial_index: int, log_size: int):
        self.initial_index = initial_index % ORDER
        self.log_size = log_size
        self.step_index = subgroup_gen_index(log_size)

    @staticmethod
    def half_odds(log_size: int) -> "Coset":
        return Coset(subgroup_gen_index(log_size + 2), log_size)

    @staticmethod
    def odds(log_size: int) -> "Coset":
        return Coset(subgroup_gen_index(log_size + 1), log_size)

    def size(self):
        return 1 << self.log_size

    def index_at(self, j: int) -> int:
        return (self.initial_index + j * self.step_index) % ORDER

    def at(self, j: int):
        return point_of_index(self.index_at(j))

    def double(self) -> "Coset":
        assert self.log_size > 0
        c = Coset(2 * self.initial_index, self.log_size - 1)
        return c

    def points(self):
        """All points in natural order as two uint64 arrays (x, y)."""
        xs = np.array([0], dtype=U64)
        ys = np.array([0], dtype=U64)
        x0, y0 = point_of_index(self.initial_index)
        xs[0], ys[0] = x0, y0
        for k in range(self.log_size):
            sx, sy = point_of_index(self.step_index << k)
            nx = m_sub(m_mul(xs, sx), m_mul(ys, sy))
            ny = m_add(m_mul(xs, sy), m_mul(ys, sx))
            xs = np.concatenate([xs, nx])
            ys = np.concatenate([ys, ny])
        return xs, ys


class CircleDomain:
    def __init__(self, half_coset: Coset):
        self.half_coset = half_coset
        self.log_size = half_coset.log_size + 1

    def size(self):
        return 1 << self.log_size

    def index_at(self, i: int) -> int:
        h = self.half_coset.size()
        if i < h:
            return self.half_coset.index_at(i)
        return (-self.half_coset.index_at(i - h)) % ORDER

    def at(self, i: int):
        return point_of_index(self.index_at(i))

    def points(self):
        """Natural (domain-index) order."""
        xs, ys = self.half_coset.points()
        return np.concatenate([xs, xs]), np.concatenate([ys, m_neg(ys)])

    def points_bitrev(self):
        """Storage order: entry s is at(bit_reverse(s))."""
        xs, ys = self.points()
        idx = bit_reverse_indices(self.log_size)
        return xs[idx], ys[idx]


class CanonicCoset:
    def __init__(self, log_size: int):
        assert log_size > 0
        self.log_size = log_size

    def coset(self) -> Coset:
        return Coset.odds(self.log_size)

    def half_coset(self) -> Coset:
        return Coset.half_odds(self.log_size - 1)

    def circle_domain(self) -> CircleDomain:
        return CircleDomain(self.half_coset())

    def step_index(self) -> int:
        return subgroup_gen_index(self.log_size)


class LineDomain:
    def __init__(self, coset: Coset):
        self.coset = coset
        self.log_size = coset.log_size

    def at(self, i: int) -> int:
        return self.coset.at(i)[0]

    def double(self) -> "LineDomain":
        return LineDomain(self.coset.double())

    def xs_bitrev(self):
        xs, _ = self.coset.points()
        return xs[bit_reverse_indices(self.log_size)]


# ----------------------------------------------------------------------------- index helpers
def bit_reverse_index(i: int, log_size: int) -> int:
    if log_size == 0:
        return i
    return int(format(i, "0%db" % log_size)[::-1], 2)


_BR_CACHE = {}


def bit_reverse_indices(log_size: int) -> np.ndarray:
    r = _BR_CACHE.get(log_size)
    if r is None:
        r = np.zeros(1, dtype=np.int64)
        for _ in range(log_size):
            r = np.concatenate([2 * r, 2 * r + 1])
        _BR_CACHE[log_size] = r
    return r


def coset_index_to_circle_domain_index(i: int, log_size: int) -> int:
    """Appendix A.2: coset (row-successor) order -> circle-domain index."""
    if i & 1 == 0:
        return i // 2
    return ((2 << log_size) - i) // 2


def coset_order_storage_indices(log_size: int) -> np.ndarray:
    """storage index (bit-reversed circle-domain order) of coset-order position i, for all i.

    Coset order = the order in which `CanonicCoset(n).coset()` (step subgroup_gen(n)) walks the
    trace rows; position i+1 is the 'next row' of position i (mask offset +1).
    """
    n = 1 << log_size
    i = np.arange(n, dtype=np.int64)
    # circle-domain index: even i -> i/2 ; odd i -> n - (i+1)/2
    cd = np.where(i % 2 == 0, i // 2, n - (i + 1) // 2)
    br = bit_reverse_indices(log_size)
    return br[cd]


# vanishing polynomial of a canonic coset of log size n evaluated at x-coordinates
def coset_vanishing_x(x, log_size: int):
    """Z(p) = pi^(n-1)(p.x) for CanonicCoset(n) (Appendix A.2).  Works on ints/arrays."""
    for _ in range(log_size - 1):
        x = (2 * x * x - 1) % P if not isinstance(x, np.ndarray) else m_sub(m_mul(2, m_mul(x, x)), 1)
    return x

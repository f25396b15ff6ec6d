"""M31 / CM31 / QM31 arithmetic (oracle; test infrastructure only).

Follows stwo `core/fields/{m31,cm31,qm31}.rs` @0790eba4 (not in /root/reference; call sites
`crates/air/src/components/mod.rs:41,168` use SECURE_EXTENSION_DEGREE = 4) as restated in
SURVEY.md Appendix A.1:  P = 2^31-1;  CM31 = M31[i]/(i^2+1);  QM31 = CM31[u]/(u^2-(2+i));
a QM31 (a,b,c,d) = (a+bi) + (c+di)u, serialised as 4 x u32 LE.

Two flavours:
  * scalar: class QM31 over Python ints (transcript values, constants);
  * vector: numpy uint64 arrays; a QM31 vector has shape (..., 4).
"""
from __future__ import annotations

import numpy as np

P = (1 << 31) - 1
U64 = np.uint64
_P = U64(P)


# ----------------------------------------------------------------------------- scalar M31
def m_inv(a: int) -> int:
    a %= P
    assert a != 0, "inverse of zero"
    return pow(a, P - 2, P)


class QM31:
    """Scalar secure-field element (a + bi) + (c + di)u with Python ints."""

    __slots__ = ("v",)

    def __init__(self, a=0, b=0, c=0, d=0):
        self.v = (a % P, b % P, c % P, d % P)

    @staticmethod
    def from_m31(a: int) -> "QM31":
        return QM31(a, 0, 0, 0)

    @staticmethod
    def from_partial_evals(e):
        """stwo `SecureField::from_partial_evals`: e0 + e1*i + e2*u + e3*iu (Appendix A.1)."""
        return e[0] + e[1] * QM31(0, 1, 0, 0) + e[2] * QM31(0, 0, 1, 0) + e[3] * QM31(0, 0, 0, 1)

    def __iter__(self):
        return iter(self.v)

    def __eq__(self, o):
        o = _q(o)
        return self.v == o.v

    def __hash__(self):
        return hash(self.v)

    def __repr__(self):
        return "QM31(%d, %d, %d, %d)" % self.v

    def __add__(self, o):
        o = _q(o)
        return QM31(*[x + y for x, y in zip(self.v, o.v)])

    __radd__ = __add__

    def __sub__(self, o):
        o = _q(o)
        return QM31(*[x - y for x, y in zip(self.v, o.v)])

    def __rsub__(self, o):
        return _q(o) - self

    def __neg__(self):
        return QM31(*[-x for x in self.v])

    def __mul__(self, o):
        o = _q(o)
        a, b, c, d = self.v
        e, f, g, h = o.v
        # (A + Bu)(C + Du) = (AC + R*BD) + (AD + BC)u ; R = 2 + i
        ac = (a * e - b * f, a * f + b * e)
        bd = (c * g - d * h, c * h + d * g)
        rbd = (2 * bd[0] - bd[1], bd[0] + 2 * bd[1])
        ad = (a * g - b * h, a * h + b * g)
        bc = (c * e - d * f, c * f + d * e)
        return QM31(ac[0] + rbd[0], ac[1] + rbd[1], ad[0] + bc[0], ad[1] + bc[1])

    __rmul__ = __mul__

    def square(self):
        return self * self

    def double(self):
        return self + self

    def conj(self):
        """`complex_conjugate`: negates the u-part (Appendix A.8)."""
        a, b, c, d = self.v
        return QM31(a, b, -c, -d)

    def inverse(self):
        a, b, c, d = self.v
        # denom = A^2 - R*B^2 in CM31
        a2 = (a * a - b * b, 2 * a * b)
        b2 = (c * c - d * d, 2 * c * d)
        rb2 = (2 * b2[0] - b2[1], b2[0] + 2 * b2[1])
        den = ((a2[0] - rb2[0]) % P, (a2[1] - rb2[1]) % P)
        n = m_inv(den[0] * den[0] + den[1] * den[1])
        di = (den[0] * n % P, -den[1] * n % P)  # 1/den
        # (A - Bu) * di
        return QM31(a * di[0] - b * di[1], a * di[1] + b * di[0],
                    -(c * di[0] - d * di[1]), -(c * di[1] + d * di[0]))

    def __truediv__(self, o):
        return self * _q(o).inverse()

    def __pow__(self, e: int):
        r, b = QM31(1), self
        while e:
            if e & 1:
                r = r * b
            b = b * b
            e >>= 1
        return r

    def is_zero(self):
        return self.v == (0, 0, 0, 0)

    def to_bytes(self) -> bytes:
        return b"".join(int(x).to_bytes(4, "little") for x in self.v)

    def np(self):
        return np.array(self.v, dtype=U64)


def _q(o) -> QM31:
    if isinstance(o, QM31):
        return o
    return QM31(int(o), 0, 0, 0)


ONE = QM31(1)
ZERO = QM31(0)


# ----------------------------------------------------------------------------- vector M31
def vm(a):
    return np.asarray(a, dtype=U64)


def m_add(a, b):
    return (vm(a) + vm(b)) % _P


def m_sub(a, b):
    return (vm(a) + _P - vm(b)) % _P


def m_neg(a):
    return (_P - vm(a)) % _P


def m_mul(a, b):
    return (vm(a) * vm(b)) % _P


def m_pow(a, e: int):
    a = vm(a)
    r = np.ones_like(a)
    b = a.copy()
    while e:
        if e & 1:
            r = m_mul(r, b)
        b = m_mul(b, b)
        e >>= 1
    return r


def m_inv_vec(a):
    return m_pow(a, P - 2)


# ----------------------------------------------------------------------------- vector CM31 (..., 2)
def c_mul(a, b):
    a0, a1, b0, b1 = a[..., 0], a[..., 1], b[..., 0], b[..., 1]
    re = m_sub(m_mul(a0, b0), m_mul(a1, b1))
    im = m_add(m_mul(a0, b1), m_mul(a1, b0))
    return np.stack([re, im], axis=-1)


def c_inv(a):
    a0, a1 = a[..., 0], a[..., 1]
    n = m_inv_vec(m_add(m_mul(a0, a0), m_mul(a1, a1)))
    return np.stack([m_mul(a0, n), m_mul(m_neg(a1), n)], axis=-1)


def _mul_r(x):
    """(2+i) * x for CM31 x."""
    x0, x1 = x[..., 0], x[..., 1]
    return np.stack([m_sub(m_add(x0, x0), x1), m_add(x0, m_add(x1, x1))], axis=-1)


# ----------------------------------------------------------------------------- vector QM31 (..., 4)
def q_from_m(a):
    a = vm(a)
    z = np.zeros_like(a)
    return np.stack([a, z, z, z], axis=-1)


def q_const(q: QM31, shape=()):
    return np.broadcast_to(q.np(), tuple(shape) + (4,)).copy()


def q_add(a, b):
    return (vm(a) + vm(b)) % _P


def q_sub(a, b):
    return (vm(a) + _P - vm(b)) % _P


def q_neg(a):
    return (_P - vm(a)) % _P


def q_mul(a, b):
    a, b = vm(a), vm(b)
    A, B, C, D = a[..., 0:2], a[..., 2:4], b[..., 0:2], b[..., 2:4]
    lo = q2_add(c_mul(A, C), _mul_r(c_mul(B, D)))
    hi = q2_add(c_mul(A, D), c_mul(B, C))
    return np.concatenate([lo, hi], axis=-1)


def q2_add(a, b):
    return (a + b) % _P


def q_mul_m(a, m):
    """QM31 vector * M31 vector (broadcast over the coordinate axis)."""
    return (vm(a) * vm(m)[..., None]) % _P


def q_mul_c(a, c):
    """QM31 * CM31 (`mul_cm31`)."""
    a = vm(a)
    return np.concatenate([c_mul(a[..., 0:2], c), c_mul(a[..., 2:4], c)], axis=-1)


def q_inv(a):
    a = vm(a)
    A, B = a[..., 0:2], a[..., 2:4]
    den = (c_mul(A, A) + _P - _mul_r(c_mul(B, B))) % _P
    di = c_inv(den)
    return np.concatenate([c_mul(A, di), (_P - c_mul(B, di)) % _P], axis=-1)


def q_to_scalar(a) -> QM31:
    return QM31(*[int(x) for x in a])

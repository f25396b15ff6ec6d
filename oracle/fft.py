"""Circle FFT: interpolate / evaluate / eval_at_point (oracle; test infrastructure only).

Contract (restating stwo `prover/backend/cpu/circle.rs` and
`core/poly/circle/*` @0790eba4; reached from `crates/air/src/utils.rs:112-128`
`tree_builder.extend_evals` and `crates/prover/src/prover.rs:56-59,179,298` `.commit`):
  * evaluations are stored bit-reversed over CanonicCoset(n).circle_domain();
  * coefficient j (bits j0..j_{n-1}) multiplies  y^j0 * x^j1 * pi(x)^j2 * pi^2(x)^j3 ...,
    pi(x) = 2x^2-1;
  * LDE = zero-extend the coefficients and evaluate on the next canonic domain.
The butterfly schedule is an implementation choice; only the contract above is observable.
"""
from __future__ import annotations

import numpy as np

from .field import (P, U64, QM31, m_add, m_sub, m_mul, m_inv, m_inv_vec, q_add, q_mul_m, q_mul, q_from_m,
                    q_to_scalar)
from .circle import CanonicCoset, CircleDomain, Coset, LineDomain, bit_reverse_indices

_TW_CACHE = {}


def domain_twiddles(log_size: int):
    """List over layers i=0..n-1 of twiddle arrays (layer i has 2^(n-1-i) entries).

    layer 0: y of the point at storage index 2h;  layer i>=1: pi^(i-1)(x at storage index h<<(i+1)).
    """
    tw = _TW_CACHE.get(log_size)
    if tw is not None:
        return tw
    dom = CanonicCoset(log_size).circle_domain()
    xs, ys = dom.points_bitrev()
    tws = [ys[0::2].copy()]
    cur = xs[0::4].copy() if log_size >= 2 else None
    for i in range(1, log_size):
        tws.append(cur)
        if i + 1 < log_size:
            nxt = cur[0::2]
            cur = m_sub(m_mul(2, m_mul(nxt, nxt)), 1)
    itws = [m_inv_vec(t) for t in tws]
    _TW_CACHE[log_size] = (tws, itws)
    return _TW_CACHE[log_size]


def interpolate(values: np.ndarray) -> np.ndarray:
    """Bit-reversed evaluations on CanonicCoset(n).circle_domain() -> coefficients.

    `values` has shape (..., 2^n).
    """
    values = np.asarray(values, dtype=U64)
    n = values.shape[-1].bit_length() - 1
    assert 1 << n == values.shape[-1] and n >= 1
    _, itws = domain_twiddles(n)
    lead = values.shape[:-1]
    a = values
    for i in range(n):
        a = a.reshape(lead + (1 << (n - 1 - i), 2, 1 << i))
        v0, v1 = a[..., 0, :], a[..., 1, :]
        t = itws[i][:, None]
        a = np.stack([m_add(v0, v1), m_mul(m_sub(v0, v1), t)], axis=-2)
    a = a.reshape(lead + (1 << n,))
    return m_mul(a, m_inv(pow(2, n, P)))


def evaluate(coeffs: np.ndarray, log_size: int) -> np.ndarray:
    """Coefficients (..., 2^m), m <= log_size -> bit-reversed evaluations on the canonic domain."""
    coeffs = np.asarray(coeffs, dtype=U64)
    m = coeffs.shape[-1]
    lead = coeffs.shape[:-1]
    size = 1 << log_size
    assert m <= size
    if m < size:
        coeffs = np.concatenate([coeffs, np.zeros(lead + (size - m,), dtype=U64)], axis=-1)
    tws, _ = domain_twiddles(log_size)
    a = coeffs
    n = log_size
    for i in range(n - 1, -1, -1):
        a = a.reshape(lead + (1 << (n - 1 - i), 2, 1 << i))
        v0, v1 = a[..., 0, :], a[..., 1, :]
        t = m_mul(v1, tws[i][:, None])
        a = np.stack([m_add(v0, t), m_sub(v0, t)], axis=-2)
    return a.reshape(lead + (size,))


def point_mappings(x: QM31, y: QM31, n: int):
    maps = [y, x]
    cur = x
    for _ in range(n - 2):
        cur = cur * cur * 2 - 1
        maps.append(cur)
    return maps[:n]


def eval_at_point(coeffs: np.ndarray, point) -> QM31:
    """Evaluate one coefficient vector (2^n,) at a secure-field circle point (x, y)."""
    coeffs = np.asarray(coeffs, dtype=U64)
    n = coeffs.shape[-1].bit_length() - 1
    maps = point_mappings(point[0], point[1], n)
    acc = q_from_m(coeffs)  # (2^n, 4)
    for k in range(n - 1, -1, -1):
        half = acc.shape[0] // 2
        lo, hi = acc[:half], acc[half:]
        acc = q_add(lo, q_mul(hi, maps[k].np()[None, :]))
    return q_to_scalar(acc[0])


# ----------------------------------------------------------------------------- line polys (FRI last layer)
def line_interpolate(values, domain: LineDomain):
    """Secure-field evaluations (bit-reversed over `domain`) -> ordered line-poly coefficients.

    Restates stwo `LineEvaluation::interpolate` + `into_ordered_coefficients` (core/poly/line.rs);
    only used for the (tiny) last FRI layer.  values: list[QM31].
    """
    n = len(values)
    log_n = n.bit_length() - 1
    vals = list(values)
    # inverse FFT over the line: layer sizes n, n/2, ...
    dom = domain
    size = n
    # work in bit-reversed order: pairs (2i, 2i+1) are (x, -x)
    layers = []
    chunks = [vals]
    while size > 1:
        xs = dom.xs_bitrev()
        new_chunks = []
        for ch in chunks:
            f0, f1 = [], []
            for i in range(size // 2):
                a, b = ch[2 * i], ch[2 * i + 1]
                xinv = m_inv(int(xs[2 * i]))
                f0.append(a + b)
                f1.append((a - b) * xinv)
            new_chunks.append(f0)
            new_chunks.append(f1)
        chunks = new_chunks
        dom = dom.double()
        size //= 2
    # chunks now hold 2^log_n singletons in "bit-reversed coefficient" order: chunk index bits
    # (msb first) = choices at successive layers (first layer = lowest-degree split).
    ninv = m_inv(n % P)
    coeffs_br = [c[0] * ninv for c in chunks]
    # ordered coefficients: coefficient index j has bit k = choice at layer k (layer 0 = x^1 term)
    out = [None] * n
    for idx, c in enumerate(coeffs_br):
        j = 0
        for k in range(log_n):
            bit = (idx >> (log_n - 1 - k)) & 1
            j |= bit << k
        out[j] = c
    return out

"""Blake2s-256 (unkeyed) + its raw compression function (oracle; test infrastructure only).

The reference path hashes through the `blake2` 0.10.6 crate (`Cargo.lock:198-199`) and stwo's
`core/vcs/blake2_hash.rs` / `core/channel/blake2s.rs` (un-vendored).  RFC 7693 is the published
algorithm; `hashlib.blake2s` is used for whole-message hashes and the explicit compression
function below for stwo's KAT-era `mix_u64`, which calls the bare
compression function on the channel digest.
"""
from __future__ import annotations

import hashlib

import numpy as np

IV = [0x6A09E667, 0xBB67AE85, 0x3C6EF372, 0xA54FF53A, 0x510E527F, 0x9B05688C, 0x1F83D9AB, 0x5BE0CD19]
SIGMA = [
    [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15],
    [14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3],
    [11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4],
    [7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8],
    [9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13],
    [2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9],
    [12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11],
    [13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10],
    [6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5],
    [10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0],
]
M32 = 0xFFFFFFFF


def _rotr(x, n):
    return ((x >> n) | (x << (32 - n))) & M32


def compress(h, m, t0=0, t1=0, f0=0, f1=0):
    """RFC 7693 compression F on 8-word state h and 16-word message m (Python ints)."""
    v = list(h) + list(IV)
    v[12] ^= t0
    v[13] ^= t1
    v[14] ^= f0
    v[15] ^= f1

    def G(a, b, c, d, x, y):
        v[a] = (v[a] + v[b] + x) & M32
        v[d] = _rotr(v[d] ^ v[a], 16)
        v[c] = (v[c] + v[d]) & M32
        v[b] = _rotr(v[b] ^ v[c], 12)
        v[a] = (v[a] + v[b] + y) & M32
        v[d] = _rotr(v[d] ^ v[a], 8)
        v[c] = (v[c] + v[d]) & M32
        v[b] = _rotr(v[b] ^ v[c], 7)

    for r in range(10):
        s = SIGMA[r]
        G(0, 4, 8, 12, m[s[0]], m[s[1]])
        G(1, 5, 9, 13, m[s[2]], m[s[3]])
        G(2, 6, 10, 14, m[s[4]], m[s[5]])
        G(3, 7, 11, 15, m[s[6]], m[s[7]])
        G(0, 5, 10, 15, m[s[8]], m[s[9]])
        G(1, 6, 11, 12, m[s[10]], m[s[11]])
        G(2, 7, 8, 13, m[s[12]], m[s[13]])
        G(3, 4, 9, 14, m[s[14]], m[s[15]])
    return [h[i] ^ v[i] ^ v[i + 8] for i in range(8)]


def blake2s(data: bytes) -> bytes:
    return hashlib.blake2s(data, digest_size=32).digest()


def blake2s_ref(data: bytes) -> bytes:
    """Pure restatement via `compress` (cross-checked against hashlib in the tests)."""
    h = list(IV)
    h[0] ^= 0x01010020
    n = len(data)
    blocks = [data[i:i + 64] for i in range(0, n, 64)] or [b""]
    t = 0
    for bi, blk in enumerate(blocks):
        last = bi == len(blocks) - 1
        t += len(blk)
        blk = blk + b"\0" * (64 - len(blk))
        m = [int.from_bytes(blk[4 * i:4 * i + 4], "little") for i in range(16)]
        h = compress(h, m, t & M32, t >> 32, M32 if last else 0, 0)
    return b"".join(x.to_bytes(4, "little") for x in h)


# ----------------------------------------------------------------------------- vectorised (Merkle layers)
def _vrotr(x, n):
    return (x >> np.uint32(n)) | (x << np.uint32(32 - n))


def compress_vec(h, m, t0, f0):
    """Vectorised F: h is a list of 8 uint32 arrays, m a list of 16 uint32 arrays."""
    v = [x.copy() for x in h] + [np.full_like(h[0], iv) for iv in IV]
    v[12] ^= np.uint32(t0)
    v[14] ^= np.uint32(f0)

    def G(a, b, c, d, x, y):
        v[a] = v[a] + v[b] + x
        v[d] = _vrotr(v[d] ^ v[a], 16)
        v[c] = v[c] + v[d]
        v[b] = _vrotr(v[b] ^ v[c], 12)
        v[a] = v[a] + v[b] + y
        v[d] = _vrotr(v[d] ^ v[a], 8)
        v[c] = v[c] + v[d]
        v[b] = _vrotr(v[b] ^ v[c], 7)

    for r in range(10):
        s = SIGMA[r]
        G(0, 4, 8, 12, m[s[0]], m[s[1]])
        G(1, 5, 9, 13, m[s[2]], m[s[3]])
        G(2, 6, 10, 14, m[s[4]], m[s[5]])
        G(3, 7, 11, 15, m[s[6]], m[s[7]])
        G(0, 5, 10, 15, m[s[8]], m[s[9]])
        G(1, 6, 11, 12, m[s[10]], m[s[11]])
        G(2, 7, 8, 13, m[s[12]], m[s[13]])
        G(3, 4, 9, 14, m[s[14]], m[s[15]])
    return [h[i] ^ v[i] ^ v[i + 8] for i in range(8)]


def blake2s_words_vec(words: np.ndarray) -> np.ndarray:
    """Hash each row of a (n, w) uint32 array (w words = 4w message bytes) -> (n, 8) uint32."""
    words = np.ascontiguousarray(words, dtype=np.uint32)
    n, w = words.shape
    nbytes = 4 * w
    nblocks = max(1, (w + 15) // 16)
    h = [np.full(n, iv, dtype=np.uint32) for iv in IV]
    h[0] ^= np.uint32(0x01010020)
    with np.errstate(over="ignore"):
        for b in range(nblocks):
            m = []
            for i in range(16):
                k = 16 * b + i
                m.append(words[:, k].copy() if k < w else np.zeros(n, dtype=np.uint32))
            last = b == nblocks - 1
            t = nbytes if last else 64 * (b + 1)
            h = compress_vec(h, m, t, M32 if last else 0)
    return np.stack(h, axis=1)

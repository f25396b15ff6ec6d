"""Blake2s Merkle commitment over mixed-size M31 columns (oracle; test infrastructure only).

Restates stwo `prover/vcs/prover.rs` + `core/vcs/{blake2_merkle,verifier}.rs` @0790eba4
(un-vendored; reached from `crates/prover/src/prover.rs:59,179,298` `tree_builder.commit`)
per SURVEY.md Appendix A.4:
  node hash = blake2s([left || right] || this layer's column values as u32 LE, tree order);
  tallest columns form the leaves, shorter columns are injected at the layer of their own size;
  empty tree root = blake2s("").
"""
from __future__ import annotations

import numpy as np

from .blake2s import blake2s, blake2s_words_vec


class MerkleTree:
    """layers[k] is a (2^k, 8) uint32 array of node hashes; layers[0] is the root."""

    def __init__(self, columns, hasher=None, layer_fn=None):
        """`layer_fn(prev_or_None, cols, size) -> (size, 8) uint32` may replace the numpy layer hashing."""
        hasher = hasher or blake2s_words_vec
        # stable sort by length descending (stwo: sorted_by_key(Reverse(len)))
        self.columns = [np.ascontiguousarray(c, dtype=np.uint32) for c in columns]
        order = sorted(range(len(self.columns)), key=lambda i: -len(self.columns[i]))
        self.sorted_columns = [self.columns[i] for i in order]
        if not self.columns:
            self.layers = [np.frombuffer(blake2s(b""), dtype="<u4").reshape(1, 8).copy()]
            return
        max_log = len(self.sorted_columns[0]).bit_length() - 1
        layers = []
        prev = None
        pos = 0
        for log_size in range(max_log, -1, -1):
            cols = []
            while pos < len(self.sorted_columns) and len(self.sorted_columns[pos]) == 1 << log_size:
                cols.append(self.sorted_columns[pos])
                pos += 1
            if layer_fn is not None:
                prev = layer_fn(prev, cols, 1 << log_size)
                layers.append(prev)
                continue
            parts = []
            if prev is not None:
                parts.append(prev.reshape(1 << log_size, 16))
            if cols:
                parts.append(np.stack(cols, axis=1))
            words = np.concatenate(parts, axis=1) if parts else np.zeros((1 << log_size, 0), np.uint32)
            prev = hasher(words)
            layers.append(prev)
        layers.reverse()
        self.layers = layers

    def root(self) -> bytes:
        return self.layers[0][0].astype("<u4").tobytes()

    def hash_at(self, layer_log: int, idx: int) -> bytes:
        return self.layers[layer_log][idx].astype("<u4").tobytes()

    def decommit(self, queries_per_log_size):
        """-> (queried_values: list[int], hash_witness: list[bytes], column_witness: list[int]).

        `queries_per_log_size`: {log_size: sorted positions}.  Order of outputs = layer by layer
        from the leaves up, node-ascending, this layer's columns in (size-sorted) tree order.
        """
        queried_values, hash_witness, column_witness = [], [], []
        pos = 0
        last_layer_queries = []
        n_layers = len(self.layers)
        for layer_log in range(n_layers - 1, -1, -1):
            layer_cols = []
            while pos < len(self.sorted_columns) and len(self.sorted_columns[pos]) == 1 << layer_log:
                layer_cols.append(self.sorted_columns[pos])
                pos += 1
            have_prev = layer_log + 1 < n_layers
            prev_q = list(last_layer_queries)
            col_q = list(queries_per_log_size.get(layer_log, []))
            pi = ci = 0
            total = []
            while pi < len(prev_q) or ci < len(col_q):
                cand = []
                if pi < len(prev_q):
                    cand.append(prev_q[pi] // 2)
                if ci < len(col_q):
                    cand.append(col_q[ci])
                node = min(cand)
                # consume all prev-layer queries that are children of `node`
                if have_prev:
                    if pi < len(prev_q) and prev_q[pi] == 2 * node:
                        pi += 1
                    else:
                        hash_witness.append(self.hash_at(layer_log + 1, 2 * node))
                    if pi < len(prev_q) and prev_q[pi] == 2 * node + 1:
                        pi += 1
                    else:
                        hash_witness.append(self.hash_at(layer_log + 1, 2 * node + 1))
                vals = [int(c[node]) for c in layer_cols]
                if ci < len(col_q) and col_q[ci] == node:
                    ci += 1
                    queried_values.extend(vals)
                else:
                    column_witness.extend(vals)
                total.append(node)
            last_layer_queries = total
        return queried_values, hash_witness, column_witness


def verify_decommitment(root: bytes, column_log_sizes, queries_per_log_size, queried_values,
                        hash_witness, column_witness) -> bool:
    """Restates stwo `MerkleVerifier::verify` (core/vcs/verifier.rs)."""
    sizes = sorted(column_log_sizes, reverse=True)
    if not sizes:
        return root == blake2s(b"")
    max_log = sizes[0]
    n_cols_by_log = {}
    for s in sizes:
        n_cols_by_log[s] = n_cols_by_log.get(s, 0) + 1
    qv, hw, cw = iter(queried_values), iter(hash_witness), iter(column_witness)
    last = []  # list of (index, hash)
    try:
        for layer_log in range(max_log, -1, -1):
            n_cols = n_cols_by_log.get(layer_log, 0)
            prev = list(last)
            col_q = list(queries_per_log_size.get(layer_log, []))
            pi = ci = 0
            total = []
            while pi < len(prev) or ci < len(col_q):
                cand = []
                if pi < len(prev):
                    cand.append(prev[pi][0] // 2)
                if ci < len(col_q):
                    cand.append(col_q[ci])
                node = min(cand)
                data = b""
                if layer_log < max_log:
                    if pi < len(prev) and prev[pi][0] == 2 * node:
                        left = prev[pi][1]
                        pi += 1
                    else:
                        left = next(hw)
                    if pi < len(prev) and prev[pi][0] == 2 * node + 1:
                        right = prev[pi][1]
                        pi += 1
                    else:
                        right = next(hw)
                    data = left + right
                if ci < len(col_q) and col_q[ci] == node:
                    ci += 1
                    vals = [next(qv) for _ in range(n_cols)]
                else:
                    vals = [next(cw) for _ in range(n_cols)]
                data += b"".join(int(v).to_bytes(4, "little") for v in vals)
                total.append((node, blake2s(data)))
            last = total
    except StopIteration:
        return False
    if any(True for _ in qv) or any(True for _ in hw) or any(True for _ in cw):
        return False
    return len(last) == 1 and last[0][1] == root

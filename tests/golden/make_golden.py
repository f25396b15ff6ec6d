#!/usr/bin/env python3
"""Regenerates tests/golden/proof_digests.json: SHA-256 of the oracle's proof bytes for seeded
synthetic pies (the reference holds no such vectors — SURVEY.md §4 — so these pin the oracle against
regressions; the KAT files in kat_simple/ are the reference's own data)."""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from luminair_amd import synthetic as syn            # noqa: E402
from oracle.channel import ProtocolVariant           # noqa: E402
from oracle.proof import to_bincode                  # noqa: E402
from oracle.prover import prove                      # noqa: E402

CASES = {
    "simple_example/kat": (lambda: syn.simple_example(), "KAT"),
    "add_only_100_seed1/kat": (lambda: syn.config2_add_only(100, 1), "KAT"),
    "add_only_4096_seed6/kat": (lambda: syn.config2_add_only(4096, 6), "KAT"),
    "chain_300_seed3/kat": (lambda: syn.chain_graph(300, 3), "KAT"),
    "config3_14_13_13_seed8/kat": (lambda: syn.config3_mixed(14, 13, 13, 8), "KAT"),
    "linear_layer_20x7_max_seed2/kat": (lambda: syn.linear_layer(20, 7, 2, True), "KAT"),
    "graph_faithful_100_seed3/pinned": (lambda: syn.config2_graph_faithful(100, 3), "PINNED"),
    "less_than_100_seed3/pinned": (lambda: syn.less_than_graph(100, 3), "PINNED"),
    "sqrt_rem_50_seed4/pinned": (lambda: syn.sqrt_rem_graph(50, 4), "PINNED"),
    # generators returning (tables, luts): LUT components with their preprocessed columns
    "activations_40_seed3/pinned": (lambda: syn.activation_graph(40, 3), "PINNED"),
    "config5_3x4x5_seed8/kat": (lambda: syn.config5_linear_layers(3, 4, 5, 8), "KAT"),
}


def tables_and_luts(name):
    out = CASES[name][0]()
    return out if isinstance(out, tuple) else (out, None)


def digest(name):
    variant = CASES[name][1]
    tabs, luts = tables_and_luts(name)
    proof = prove([(k, r.astype(np.uint64)) for k, r in tabs], variant=ProtocolVariant[variant], luts=luts)
    b = to_bincode(proof)
    return {"sha256": hashlib.sha256(b).hexdigest(), "len": len(b)}


if __name__ == "__main__":
    out = {name: digest(name) for name in CASES}
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "proof_digests.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print(json.dumps(out, indent=1, sort_keys=True))

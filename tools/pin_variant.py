#!/usr/bin/env python3
"""Which protocol was this proof made with?  Exhaustive search over the LMN_PV_* protocol flags
(include/luminair_hip.h) through the product's host-only verifier - no GPU needed for the search.

    python tools/pin_variant.py proof.bin                       # search; prints the accepting flag combination(s)
    python tools/pin_variant.py proof.bin --tables DIR          # + prove the same tables here, first diverging field
    python tools/pin_variant.py proof.bin --digests             # + the channel digest after every transcript step

`proof.bin` = `LuminairProof::to_bincode()` bytes of the reference (crates/prover/src/lib.rs:25-32); `--json` reads
the serde-JSON form instead.  DIR holds what INTEGRATION.md's "Pinning the protocol of a build" test dumps:
`table_<kind>.bin` (AoS rows, little-endian u32, `Column::index()` order) per trace table of the pie and, for LUT
graphs, `lut_<sin|exp2|log2>_<0|1>.bin`.  With --tables the proofs are made by the library given with --library
(default: the HIP library - needs a GPU).

Exit status: 0 = exactly one combination accepts (over the flags the proof depends on), 1 = several, 2 = none.
Logic: luminair_amd/pinning.py.
"""
import argparse
import glob
import json
import os
import re
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from luminair_amd import backend as B, pinning  # noqa: E402


def load_tables(d, lib):
    tables = []
    for path in sorted(glob.glob(os.path.join(d, "table_*.bin")), key=lambda p: int(re.findall(r"table_(\d+)\.bin", p)[0])):
        kind = int(re.findall(r"table_(\d+)\.bin", path)[0])
        ncols = int(lib.lib.lmn_kind_columns(kind))
        if ncols == 0:
            raise SystemExit("unsupported table kind %d" % kind)
        rows = np.fromfile(path, dtype="<u4")
        if rows.size == 0 or rows.size % ncols:
            raise SystemExit("%s: %d words is not a multiple of the %d columns of kind %d" % (path, rows.size, ncols, kind))
        tables.append((kind, np.ascontiguousarray(rows.reshape(-1, ncols)), rows.size // ncols))
    luts = {}
    for name in ("sin", "exp2", "log2"):
        p0, p1 = (os.path.join(d, "lut_%s_%d.bin" % (name, i)) for i in (0, 1))
        if os.path.exists(p0) and os.path.exists(p1):
            luts[name] = (np.fromfile(p0, dtype="<u4"), np.fromfile(p1, dtype="<u4"))
    return tables, (luts or None)


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("proof")
    ap.add_argument("--json", action="store_true", help="the proof file is LuminairProof's serde-JSON form")
    ap.add_argument("--tables", help="directory with table_<kind>.bin (+ lut_<name>_<0|1>.bin) dumps of the pie")
    ap.add_argument("--library", help="library to load (default: luminair_amd/csrc/libluminair_hip.so)")
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--pow-bits", type=int), ap.add_argument("--n-queries", type=int)
    ap.add_argument("--log-last-layer", type=int)
    ap.add_argument("--digests", action="store_true", help="print the channel digest after every transcript step")
    ap.add_argument("--report", help="write the full result as JSON here")
    a = ap.parse_args()

    lib = B.Library(a.library) if a.library else B.default_library()
    proof = open(a.proof, "rb").read()
    if a.json:
        from luminair_amd.pie import LuminairProof
        proof = LuminairProof.from_json(proof.decode()).to_bincode()
    cfg = lib.default_config()
    for k, v in (("pow_bits", a.pow_bits), ("n_queries", a.n_queries), ("log_last_layer", a.log_last_layer)):
        if v is not None:
            setattr(cfg, k, v)
    res = pinning.search(lib, proof, cfg, keep_steps=a.digests)
    print(pinning.format_report(res))
    candidates = res.accepted or [t.flags for t in sorted(res.trials, key=lambda t: (-t.score, t.flags))[:4]]
    if a.digests:
        for f in candidates[:2]:
            t = next(t for t in res.trials if t.flags == f)
            print("channel digests of the replay under 0x%04x %s:" % (f, pinning.flag_names(f) or ["KAT"]))
            for name, idx, dg in t.steps:
                print("  %-20s %3d  %s" % (name, idx, dg))
    out = {"accepted": res.accepted, "unique": res.unique, "kinds": res.kinds,
           "determined": {B.PV_NAMES[b]: v for b, v in res.determined.items()},
           "undetermined": [B.PV_NAMES[b] for b in res.undetermined],
           "irrelevant": [B.PV_NAMES[b] for b in res.irrelevant_bits], "diagnosis": res.diagnosis(),
           "trials": [{"flags": t.flags, "names": pinning.flag_names(t.flags), "rc": t.rc,
                       "passed": pinning.check_names(t.passed), "failed": pinning.check_names(t.failed),
                       "message": t.message} for t in res.trials]}
    if a.tables:
        tables, luts = load_tables(a.tables, lib)
        out["divergence"] = {}
        seen = set()
        for f in candidates:
            # the constraint-form / transcript flags of components absent from the proof do not matter: one run per distinct value
            if f in seen:
                continue
            seen.add(f)
            c = lib.default_config()
            for k in ("pow_bits", "n_queries", "log_last_layer"):
                setattr(c, k, getattr(cfg, k))
            c.protocol_variant = f
            ctx = B.Context(a.device, c, lib)
            try:
                ours = ctx.prove_tables(tables, luts)
                where = pinning.first_divergence(ours, proof, bool(f & B.PV_CLAIM17))
                msg = "byte-identical to the given proof" if where is None else "first difference: " + where
            except B.LuminairBackendError as e:
                msg = "prover refused the tables: %s" % e
            finally:
                ctx.close()
            print("prove under 0x%04x %s: %s" % (f, pinning.flag_names(f) or ["KAT"], msg))
            out["divergence"]["0x%04x" % f] = msg
    if a.report:
        with open(a.report, "w") as fh:
            json.dump(out, fh, indent=1)
    sys.exit(0 if res.unique else (1 if res.accepted else 2))


if __name__ == "__main__":
    main()

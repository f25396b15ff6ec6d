"""The lock-step batch library's HOST code on the CPU (round 6): `csrc/batch.cpp` - member fibers on hand-made stacks,
wait-free rendezvous, per-launch argument tables, the batched copy list, bounce slots - and `batch.h` compiled as they are
into the emulation runtime (`tests/emu/build_emu.sh batch`: the dozen HIP calls they make are served from host memory,
the trampoline runs through the emulated launch with blockIdx.z = the member).  What the GPU tests of tests/test_batch.py
check on the device is checked here against the emulation build's `lmn_prove`: byte-identical proofs for several operators,
a bad pie failing alone, mixed shapes refused, batches that grow, more members than worker threads.  tests/test_sanitizers.py
runs the same scenarios under ASan + UBSan and TSan."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
from luminair_amd import backend, synthetic as syn          # noqa: E402
from luminair_amd.batch import BatchProver                   # noqa: E402

EMU = os.path.join(ROOT, "tests", "emu", "libluminair_emu.so")
EMU_BATCH = os.path.join(ROOT, "tests", "emu", "libluminair_emu_batch.so")


def _build(kind=None):
    so = EMU_BATCH if not kind else EMU_BATCH.replace(".so", "_%s.so" % kind)
    csrc = os.path.join(ROOT, "luminair_amd", "csrc")
    srcs = [os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith((".hip", ".cpp", ".h"))]
    srcs += [os.path.join(ROOT, "tests", "emu", f) for f in ("emu_runtime.cpp", "build_emu.sh")]
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        r = subprocess.run([os.path.join(ROOT, "tests", "emu", "build_emu.sh"), "batch"] + ([kind] if kind else []),
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-3000:]
    if not os.path.exists(EMU) or any(os.path.getmtime(s) > os.path.getmtime(EMU) for s in srcs):
        subprocess.run([os.path.join(ROOT, "tests", "emu", "build_emu.sh")], check=True, capture_output=True)
    return so


def _solo(variant):
    lib = backend.Library(EMU)
    cfg = lib.default_config()
    cfg.protocol_variant = variant
    return backend.Context(0, cfg, lib)


def _pie(tabs):
    return [(k, r, len(r)) for k, r in tabs]


def _raw_batch(bp, pies, luts=None):
    """lmn_batch_prove with the per-pie return codes (BatchProver raises on the first)"""
    n = len(pies)
    arrs = (C.POINTER(backend.LmnTable) * n)()
    keep = []
    for i, t in enumerate(pies):
        arr, nt, st, k = backend.Context._marshal_tables(None, t, luts)
        keep.append(k)
        arrs[i] = C.cast(arr, C.POINTER(backend.LmnTable))
    proofs, lens, rcs = (C.POINTER(C.c_uint8) * n)(), (C.c_size_t * n)(), (C.c_int * n)()
    rc = bp.lib.lib.lmn_batch_prove(bp.handle, n, arrs, nt, C.byref(st), proofs, lens, rcs)
    out = []
    for i in range(n):
        out.append(C.string_at(proofs[i], lens[i]) if proofs[i] else None)
        if proofs[i]:
            bp.lib.lib.lmn_free(proofs[i])
    return rc, list(rcs), out


def scenario_operators(batch_so):
    solo = _solo(backend.VARIANT_PINNED)
    bp = BatchProver(0, 3, protocol_variant=backend.VARIANT_PINNED, library_path=batch_so)
    try:
        act, luts = syn.activation_graph(40, 1, names=("sin",))
        for name, mk, lt in (("add + inputs", lambda s: syn.config2_graph_faithful(64, s), None),
                             ("sqrt + rem", lambda s: syn.sqrt_rem_graph(48, s), None),
                             ("less_than + range-check LUT", lambda s: syn.less_than_graph(40, s), None),
                             ("sin + LUT tree 0", lambda s: syn.activation_graph(40, s, names=("sin",))[0], luts)):
            pies = [_pie(mk(50 + i)) for i in range(3)]
            assert bp.prove_batch(pies, lt) == [solo.prove_tables(p, lt) for p in pies], name
        c = bp.counters()
        assert c["host_waits"] < c["launches"] and c["direct_copies"] == 0, c
    finally:
        bp.close()
        solo.close()


def scenario_kat_and_fewer_threads_than_members(batch_so, kat_bytes, members=5):
    os.environ["LMN_BATCH_THREADS"] = "2"          # five members on two worker threads: three fibers on one of them
    try:
        bk = BatchProver(0, members, library_path=batch_so)
    finally:
        del os.environ["LMN_BATCH_THREADS"]
    try:
        kat = _pie(syn.simple_example())
        assert bk.prove_batch([kat] * members) == [kat_bytes] * members
        assert bk.prove_batch([kat] * 2) == [kat_bytes] * 2    # a batch smaller than the number of slots
    finally:
        bk.close()


def scenario_failures(batch_so, rows=64, repeats=2):
    solo = _solo(backend.VARIANT_PINNED)
    bp = BatchProver(0, 4, protocol_variant=backend.VARIANT_PINNED, library_path=batch_so)
    try:
        good = [_pie(syn.config2_graph_faithful(rows, 7 + i)) for i in range(4)]
        want = [solo.prove_tables(p) for p in good]
        other = _pie(syn.config2_graph_faithful(200, 3))
        try:
            bp.prove_batch(good[:3] + [other])
            raise AssertionError("pies of different shapes were accepted")
        except backend.LuminairBackendError as e:
            assert e.code == backend.ERR_INVALID_ARGUMENT
        # pie 2 violates its constraints: the other three proofs are still produced and correct
        bad = [(k, r.copy(), n) for k, r, n in good[2]]
        bad[0][1][5, 11] = (int(bad[0][1][5, 11]) + 1) % ((1 << 31) - 1)
        rc, rcs, out = _raw_batch(bp, good[:2] + [bad] + good[3:])
        assert rc == backend.ERR_CONSTRAINTS and rcs == [0, 0, backend.ERR_CONSTRAINTS, 0], (rc, rcs)
        assert [out[i] for i in (0, 1, 3)] == [want[i] for i in (0, 1, 3)] and out[2] is None
        # pie 1 holds a non-canonical word: rejected alone; the slot proves correctly in the next batches
        bad = [(k, r.copy(), n) for k, r, n in good[1]]
        bad[0][1][7, 9] = (1 << 31) - 1
        rc, rcs, out = _raw_batch(bp, [good[0], bad, good[2], good[3]])
        assert rc == backend.ERR_INVALID_ARGUMENT and rcs == [0, backend.ERR_INVALID_ARGUMENT, 0, 0], (rc, rcs)
        assert [out[i] for i in (0, 2, 3)] == [want[i] for i in (0, 2, 3)]
        for _ in range(repeats):
            assert bp.prove_batch(good) == want
    finally:
        bp.close()
        solo.close()


def scenario_growth(batch_so):
    """more slots, then a larger shape, than the batches before: contexts that prove a shape for the first time build their
    twiddle tables outside lock-step (Context::prepare_for)"""
    solo = _solo(backend.VARIANT_PINNED)
    bp = BatchProver(0, 4, protocol_variant=backend.VARIANT_PINNED, library_path=batch_so)
    try:
        small = [_pie(syn.config2_graph_faithful(40, 20 + i)) for i in range(4)]
        big = [_pie(syn.config2_graph_faithful(600, 30 + i)) for i in range(4)]
        ws, wb = [solo.prove_tables(p) for p in small], [solo.prove_tables(p) for p in big]
        assert bp.prove_batch(small[:2]) == ws[:2]
        assert bp.prove_batch(small) == ws
        assert bp.prove_batch(big[:3]) == wb[:3]
        assert bp.prove_batch(big) == wb
        assert bp.prove_batch(small) == ws
    finally:
        bp.close()
        solo.close()


def scenario_two_groups(batch_so, rows=64, slots=3, n_pies=8):
    """two batch groups driven at once (`BatchPool`): the groups share the library's registries (page-locked ranges, twiddle
    tables) and nothing else"""
    from luminair_amd.batch import BatchPool
    solo = _solo(backend.VARIANT_PINNED)
    pool = BatchPool(0, groups=2, slots=slots, protocol_variant=backend.VARIANT_PINNED, library_path=batch_so)
    try:
        pies = [_pie(syn.config2_graph_faithful(rows, 70 + i)) for i in range(n_pies)]
        want = [solo.prove_tables(p) for p in pies]
        assert pool.prove_many(pies) == want
        assert pool.prove_many(pies[:slots + 1]) == want[:slots + 1]
    finally:
        pool.close()
        solo.close()


def run_all(batch_so, small=False):
    kat = open(os.path.join(ROOT, "tests", "golden", "kat_simple", "proof"), "rb").read()
    if small:     # the thread sanitizer tracks every fiber switch of every emulated lane (~30 s per 16-row proof): three
        # members on two worker threads, a smaller batch, a refused mix, a bad pie failing alone, the slot proving again
        scenario_kat_and_fewer_threads_than_members(batch_so, kat, members=3)
        scenario_failures(batch_so, 12, repeats=1)
        scenario_two_groups(batch_so, rows=12, slots=2, n_pies=4)
        return
    scenario_kat_and_fewer_threads_than_members(batch_so, kat)
    scenario_failures(batch_so)
    scenario_growth(batch_so)
    scenario_operators(batch_so)
    scenario_two_groups(batch_so)


def test_emu_batch_kat_and_more_members_than_worker_threads(kat_bytes):
    scenario_kat_and_fewer_threads_than_members(_build(), kat_bytes)


def test_emu_batch_bad_pies_fail_alone_and_mixed_shapes_are_refused():
    scenario_failures(_build())


def test_emu_two_batch_groups_at_once():
    scenario_two_groups(_build())


def test_emu_batch_grows_across_calls():
    scenario_growth(_build())


def test_emu_batched_proofs_equal_lmn_prove_for_several_operators():
    scenario_operators(_build())


if __name__ == "__main__":
    # `python tests/test_batch_emu.py <batch library>`: every scenario without pytest (the thread sanitizer's interpreter)
    run_all(sys.argv[1], small=len(sys.argv) > 2 and sys.argv[2] == "small")
    print("emulated batches ok")

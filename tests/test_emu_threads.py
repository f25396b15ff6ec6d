"""Host code shared by contexts that are driven from several host threads: the per-device twiddle registry (contexts that
prove larger shapes replace the shared set while others still use the old one), the proofs-in-flight counter, the
per-context locks, `lmn_prove_submit / _wait` next to a plain `lmn_prove`.  Runs on the emulation build (the "device" is
one emulated launch at a time, the HOST code of the threads runs concurrently) - under the thread sanitizer in
tests/test_sanitizers.py; here it checks that every thread gets the bytes a sequential run gives."""
import os
import threading

import numpy as np

from luminair_amd import backend, synthetic as syn


def _pies(small=False):
    if small:   # the thread-sanitizer run: every emulated lane is a fiber switch the sanitizer tracks
        return [syn.config2_add_only(40, 1), syn.chain_graph(60, 3), syn.config2_add_only(200, 5), syn.simple_example()]
    return [syn.config2_add_only(100, 1), syn.chain_graph(300, 3), syn.config2_add_only(1 << 11, 5), syn.chain_graph(40, 9),
            syn.chain_graph(1 << 12, 7), syn.simple_example()]


def test_emu_contexts_in_concurrent_threads(root):
    run_concurrent_contexts(os.path.join(root, "tests", "emu", "libluminair_emu.so"))


def run_concurrent_contexts(so_path, small=False):
    lib = backend.Library(so_path)
    pies = [[(k, r, len(r)) for k, r in tabs] for tabs in _pies(small)]
    seq = backend.Context(0, None, lib)
    want = [seq.prove_tables(p) for p in pies]
    seq.close()
    got = [[None] * len(pies) for _ in range(4)]
    errors = []

    def worker(w):
        try:
            ctx = backend.Context(0, None, lib)
            # every thread walks the shapes in its own order: the shared twiddle set grows under the others' feet
            order = list(range(len(pies)))
            order = order[w:] + order[:w]
            for i in order:
                got[w][i] = ctx.prove_tables(pies[i])
            ctx.close()
        except BaseException as e:  # noqa: BLE001
            errors.append((w, repr(e)))

    def submitter():
        try:
            ctx = backend.Context(0, None, lib)
            for i in (1, len(pies) - 1):
                ctx.prove_submit(pies[i])
                assert ctx.prove_wait() == want[i]
            ctx.close()
        except BaseException as e:  # noqa: BLE001
            errors.append(("submit", repr(e)))

    threads = [threading.Thread(target=worker, args=(w,)) for w in range(4)] + [threading.Thread(target=submitter)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for w in range(4):
        assert got[w] == want, w


if __name__ == "__main__":
    # `python tests/test_emu_threads.py <library>`: the same scenario without pytest (tests/test_sanitizers.py runs it in an
    # interpreter that has the thread sanitizer's runtime preloaded - and nothing else that brings threads of its own)
    import sys
    run_concurrent_contexts(sys.argv[1], small=True)
    # error paths and reuse from a second thread
    lib_ = backend.Library(sys.argv[1])
    ctx_ = backend.Context(0, None, lib_)
    bad = syn.config2_add_only(64, 9)[0][1].copy()
    bad[3, 11] ^= 1

    def failing():
        try:
            ctx_.prove_tables([(0, bad, len(bad))])
            raise SystemExit("a violated constraint went unnoticed")
        except backend.LuminairBackendError as e:
            assert e.code == backend.ERR_CONSTRAINTS
    th = threading.Thread(target=failing)
    th.start()
    th.join()
    ctx_.prove_tables([(k, r, len(r)) for k, r in syn.simple_example()])
    ctx_.close()
    print("concurrent contexts ok")

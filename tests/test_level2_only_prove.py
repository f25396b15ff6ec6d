"""The level-2 surface ALONE reproduces proofs (VERDICT r2 missing #2): `oracle.prover.prove`'s host logic drives
`lmn_col_*` / `lmn_tree_*` for every per-row pass (tests/level2_prover.py) - `lmn_prove` is never called - and the
bytes equal the reference's known-answer proof / the oracle's own proofs.  Here through the test-only emulation build;
tests/test_gpu_parity.py runs the same check on the MI355X."""
import os
import subprocess

import numpy as np
import pytest

from luminair_amd import backend, synthetic as syn
from oracle.channel import ProtocolVariant
from oracle.proof import to_bincode
from oracle.prover import prove as oracle_prove

from level2_prover import prove_with_level2_only


@pytest.fixture(scope="module")
def emu_ctx(root):
    so = os.path.join(root, "tests", "emu", "libluminair_emu.so")
    if not os.path.exists(so):
        subprocess.run([os.path.join(root, "tests", "emu", "build_emu.sh")], check=True, capture_output=True)
    return backend.Context(0, None, backend.Library(so))


def test_level2_only_reproduces_the_reference_kat(emu_ctx, kat_bytes):
    got, calls = prove_with_level2_only(emu_ctx, syn.simple_example())
    assert got == kat_bytes
    # every stage went through the handle ops: 2 components -> 2 logup + 2 composition calls, 3 non-empty trace trees
    # + composition tree + FRI layers committed on the device
    assert calls["logup"] == 2 and calls["composition"] == 2 and calls["commit"] >= 4 + 4
    assert calls["accumulate_quotients"] >= 2 and calls["fold_line"] >= 4 and calls["eval_at_point"] >= 31 + 24 + 4


@pytest.mark.parametrize("name,tabs,variant,luts", [
    ("chain (Add+Mul+Recip, ragged)", syn.chain_graph(100, 3), ProtocolVariant.KAT, None),
    ("mixed sizes", [(0, syn.chain_graph(64, 4)[0][1]), (1, syn.chain_graph(500, 5)[1][1])], ProtocolVariant.KAT, None),
    ("graph-faithful Add + Inputs (PINNED)", syn.config2_graph_faithful(40, 3), ProtocolVariant.PINNED, None),
    ("LessThan + range-check LUT (PINNED)", syn.less_than_graph(30, 5), ProtocolVariant.PINNED, None),
    ("linear layer + max", syn.linear_layer(20, 7, 2, True), ProtocolVariant.KAT, None),
])
def test_level2_only_equals_oracle(emu_ctx, name, tabs, variant, luts):
    want = to_bincode(oracle_prove([(k, r.astype(np.uint64)) for k, r in tabs], variant=variant, luts=luts))
    got, _ = prove_with_level2_only(emu_ctx, tabs, variant, luts)
    assert got == want, name


def test_level2_only_with_a_lut_component(emu_ctx):
    tabs, luts = syn.activation_graph(30, 4, names=("sin",), ranges={"sin": (-300, 200)})
    want = to_bincode(oracle_prove([(k, r.astype(np.uint64)) for k, r in tabs], variant=ProtocolVariant.PINNED, luts=luts))
    got, calls = prove_with_level2_only(emu_ctx, tabs, ProtocolVariant.PINNED, luts)
    assert got == want and calls["logup"] == 3


def test_logup_and_composition_argument_validation(emu_ctx):
    E = backend.LuminairBackendError
    main = emu_ctx.col_from_cpu(np.zeros((15, 16), np.uint32))
    wrong = emu_ctx.col_from_cpu(np.zeros((14, 16), np.uint32))
    small = emu_ctx.col_from_cpu(np.zeros((15, 8), np.uint32))
    elems = {0: ((1, 2, 3, 4), (5, 6, 7, 8))}
    inter, claimed = emu_ctx.col_logup(0, main, None, elems)
    assert inter.ncols == 12 and inter.log_size == 4 and claimed == (0, 0, 0, 0)    # all multiplicities are zero
    assert emu_ctx.lib.lib.lmn_kind_constraints(0) == 9 and emu_ctx.lib.lib.lmn_kind_relations(0) == 3
    acc = emu_ctx.col_zeros(4, 5)
    main_lde, inter_lde = emu_ctx.col_zeros(15, 5), emu_ctx.col_zeros(12, 5)
    ok_coeffs = [(1, 0, 0, 0)] * 9
    emu_ctx.col_composition(0, main_lde, inter_lde, None, elems, (0, 0, 0, 0), ok_coeffs, acc)
    P = (1 << 31) - 1
    for fn in (lambda: emu_ctx.col_logup(0, wrong, None, elems), lambda: emu_ctx.col_logup(99, main, None, elems),
               lambda: emu_ctx.col_logup(0, small, None, elems),
               lambda: emu_ctx.col_logup(10, emu_ctx.col_zeros(1, 8), None, elems),       # lookup kind without its LUT
               lambda: emu_ctx.col_logup(0, main, None, {0: ((P, 0, 0, 0), (1, 0, 0, 0))}),
               lambda: emu_ctx.col_composition(0, main_lde, inter_lde, None, elems, (0, 0, 0, 0), ok_coeffs[:8], acc),
               lambda: emu_ctx.col_composition(1, main_lde, inter_lde, None, elems, (0, 0, 0, 0), ok_coeffs, acc),
               lambda: emu_ctx.col_composition(0, main_lde, inter_lde, None, elems, (0, 0, 0, 0), ok_coeffs, emu_ctx.col_zeros(4, 6)),
               lambda: main.view(14, 2), lambda: main.view(0, 0)):
        with pytest.raises(E) as e:
            fn()
        assert e.value.code == backend.ERR_INVALID_ARGUMENT
    v = main.view(3, 2)
    assert v.ncols == 2 and v.log_size == 4 and v.device_ptr == main.device_ptr + 3 * 16 * 4
    v.free()
    assert main.to_cpu().shape == (15, 16)          # freeing a view frees nothing

"""The reference's producer edge cases re-expressed for `DeviceGraph` (VERDICT r2 missing #3): the 15 expansion /
broadcast scenarios of /root/reference/crates/graph/src/tests/expansions.rs:64-368 (+ its two edge cases), the
binary-op shape matrix of crates/graph/src/tests/mod.rs:64-190 (3x4, 32x32, 17x13, scalar / row / column broadcast,
for Add and Mul: tests/ops.rs:23-24), the unary shape set (tests/mod.rs:50-62) and the reduce / LessThan / Contiguous
cases of tests/ops.rs:27-256.

The reference checks `gen_trace -> prove -> verify` and the output against a float CPU run; here: gen_trace ON THE
DEVICE -> every table equals the numpy mirror's (tests/host_graph.py) -> `lmn_prove` -> `lmn_verify` (whose
logup check is sum of claimed sums == 0, i.e. every expansion-adjusted multiplicity is right) -> output equals the
mirror's fixed-point result.  Used by tests/test_producer_scenarios.py (emulation build, CPU) and
tests/test_gpu_parity.py (MI355X)."""
from __future__ import annotations

import numpy as np

from luminair_amd import backend
from luminair_amd.graph import DeviceGraph

S = 4096


def rnd(rng, shape, nonzero=False):
    """`random_vec_rng` (tests/mod.rs:200-214): U(-0.5, 0.5) as Fixed<12>; nonzero: at least 0.001"""
    v = rng.integers(-2048, 2048, size=shape)
    if nonzero:
        v = rng.integers(5, 2048, size=shape)
    return v.astype(np.int64)


def _ex(t, axis, size):
    return DeviceGraph.luminal_expand(t, axis, size)


# ---- crates/graph/src/tests/expansions.rs
def single_dimension_expansion(g, rng):
    a, b = g.input(rnd(rng, (2, 3))), g.input(rnd(rng, (2, 1)))
    return [g.mul(a, _ex(b, 1, 3))]


def multiple_dimension_expansion(g, rng):
    a, b = g.input(rnd(rng, (3, 4, 2))), g.input(rnd(rng, (1, 1, 2)))
    return [g.add(a, _ex(_ex(b, 0, 3), 1, 4))]


def scalar_broadcasting(g, rng):
    a, s = g.input(rnd(rng, (3, 4))), g.input(np.array([[int(2.5 * S)]]))
    return [g.mul(a, g.expand_to(s, (3, 4)))]


def chained_expansions(g, rng):
    a, b = g.input(rnd(rng, (2, 3))), g.input(rnd(rng, (1, 3)))
    inter = g.add(a, _ex(b, 0, 2))
    c = g.input(rnd(rng, (2, 1)))
    return [g.mul(inter, _ex(c, 1, 3))]


def multiple_consumers_different_expansions(g, rng):
    base = g.input(rnd(rng, (2, 2)))
    a = g.input(rnd(rng, (2, 2, 3)))
    r1 = g.mul(_ex(base, 2, 3), a)
    b = g.input(rnd(rng, (2, 2, 4)))
    r2 = g.add(_ex(base, 2, 4), b)
    return [g.add(g.sum_reduce(r1, 2), g.sum_reduce(r2, 2))]


def mixed_real_fake_dimensions(g, rng):
    a, b = g.input(rnd(rng, (3, 2, 4))), g.input(rnd(rng, (3, 1, 4)))
    return [g.mul(a, _ex(b, 1, 2))]


def row_vector_broadcasting(g, rng):
    m, r = g.input(rnd(rng, (4, 5))), g.input(rnd(rng, (1, 5)))
    return [g.add(m, _ex(r, 0, 4))]


def column_vector_broadcasting(g, rng):
    m, c = g.input(rnd(rng, (4, 5))), g.input(rnd(rng, (4, 1)))
    return [g.mul(m, _ex(c, 1, 5))]


def complex_expansion_chain(g, rng):
    a, b, c = g.input(rnd(rng, (2, 3))), g.input(rnd(rng, (1, 3))), g.input(rnd(rng, (2, 1)))
    d = g.input(np.array([[int(1.5 * S)]]))
    s1 = g.add(a, _ex(b, 0, 2))
    s2 = g.mul(s1, _ex(c, 1, 3))
    s3 = g.add(s2, g.expand_to(d, (2, 3)))
    e = g.input(rnd(rng, (2, 3, 4)))
    return [g.mul(_ex(s3, 2, 4), e)]


def nested_operations_with_expansions(g, rng):
    x, y, z = g.input(rnd(rng, (3, 2))), g.input(rnd(rng, (1, 2))), g.input(rnd(rng, (3, 1)))
    left = g.add(x, _ex(y, 0, 3))
    right = g.add(x, _ex(z, 1, 2))
    return [g.mul(left, right)]


def reduction_after_expansion(g, rng):
    base, w = g.input(rnd(rng, (2, 3))), g.input(rnd(rng, (1, 3)))
    return [g.sum_reduce(g.mul(base, _ex(w, 0, 2)), 1)]


def large_expansion_factors(g, rng):
    small, large = g.input(np.array([[int(3.14 * S)]])), g.input(rnd(rng, (8, 16)))
    return [g.add(large, g.expand_to(small, (8, 16)))]


def expansion_with_unary_operations(g, rng):
    g.set_lut("sin", -2048, 2047)
    base = g.input(rnd(rng, (2, 2)))
    other = g.input(rnd(rng, (2, 2, 3)))
    return [g.mul(_ex(g.sin(base), 2, 3), other)]


def zero_expansion_edge_case(g, rng):
    a, b = g.input(rnd(rng, (1, 4))), g.input(rnd(rng, (3, 4)))
    return [g.add(_ex(a, 0, 3), b)]


def identity_expansion_edge_case(g, rng):
    a = g.input(rnd(rng, (3, 3)))
    other = g.input(rnd(rng, (3, 3, 1)))
    return [g.add(_ex(a, 2, 1), other)]


def comprehensive_integration(g, rng):
    g.set_lut("sin", -3 * S, 3 * S)
    i1, i2, i3 = g.input(rnd(rng, (2, 3))), g.input(rnd(rng, (1, 3))), g.input(rnd(rng, (2, 1)))
    bias = g.input(np.array([[int(0.1 * S)]]))
    m1 = g.add(i1, _ex(i2, 0, 2))
    m2 = g.mul(m1, _ex(i3, 1, 3))
    m3 = g.add(m2, g.expand_to(bias, (2, 3)))
    filt = g.input(rnd(rng, (2, 3, 4)))
    filtered = g.mul(_ex(g.sin(m3), 2, 4), filt)
    red = g.sum_reduce(filtered, 2)
    fb = g.input(np.array([[int(-0.05 * S)]]))
    return [g.add(red, g.expand_to(fb, (2, 3)))]


EXPANSIONS = [single_dimension_expansion, multiple_dimension_expansion, scalar_broadcasting, chained_expansions,
              multiple_consumers_different_expansions, mixed_real_fake_dimensions, row_vector_broadcasting,
              column_vector_broadcasting, complex_expansion_chain, nested_operations_with_expansions,
              reduction_after_expansion, large_expansion_factors, expansion_with_unary_operations,
              zero_expansion_edge_case, identity_expansion_edge_case, comprehensive_integration]


# ---- crates/graph/src/tests/mod.rs:64-190 (binary_test!) and :50-62 (unary_test!), tests/ops.rs
BINARY_SHAPES = [((3, 4), (3, 4)), ((32, 32), (32, 32)), ((17, 13), (17, 13)), ((1, 1), (5, 5)), ((1, 4), (3, 4)),
                 ((3, 1), (3, 4))]
UNARY_SHAPES = [(3, 4), (1, 1), (1, 8), (8, 1)]


def binary_case(op, sa, sb):
    def build(g, rng):
        a, b = g.input(rnd(rng, sa)), g.input(rnd(rng, sb))

        def bc(t, st, so):        # the macro's broadcasting rule (tests/mod.rs:80-112)
            if st == so:
                return t
            if st == (1, 1):
                return g.expand_to(t, so)
            if st[0] == 1:
                return _ex(t, 0, so[0])
            if st[1] == 1:
                return _ex(t, 1, so[1])
            return t
        return [getattr(g, op)(bc(a, sa, sb), bc(b, sb, sa))]
    build.__name__ = "%s_%dx%d_%dx%d" % (op, sa[0], sa[1], sb[0], sb[1])
    return build


def unary_case(op, shape):
    def build(g, rng):
        if op in ("sin", "exp2"):
            g.set_lut(op, 0, 2047)
        return [getattr(g, op)(g.input(rnd(rng, shape, nonzero=True)))]
    build.__name__ = "%s_%dx%d" % (op, shape[0], shape[1])
    return build


def reduce_case(op):
    def build(g, rng):        # tests/ops.rs:27-66: three reductions of one (1, 4, 100) tensor, three outputs
        a = g.input(rnd(rng, (1, 4, 100)))
        f = g.sum_reduce if op == "sum" else g.max_reduce
        return [f(a, 1), f(a, 0), f(a, 2)]
    build.__name__ = "%s_reduce_1x4x100" % op
    return build


def less_than_case(shape):
    def build(g, rng):
        return [g.less_than(g.input(rnd(rng, shape)), g.input(rnd(rng, shape)))]
    build.__name__ = "less_than_%dx%d" % shape
    return build


def contiguous_slice(g, rng):   # tests/ops.rs:224-256: a.slice((.., 0..1)).contiguous()
    a = g.input(rnd(rng, (2, 2)))
    return [g.contiguous(g.slice(a, (None, (0, 1))))]


def contiguous_permuted(g, rng):
    a = g.input(rnd(rng, (3, 5)))
    b = g.contiguous(g.permute(a, (1, 0)))          # consumed again: the yields of the materialised tensor must balance
    return [g.add(b, g.input(rnd(rng, (5, 3))))]


def contiguous_expanded(g, rng):
    a = g.input(rnd(rng, (4,)))
    b = g.contiguous(g.expand(a, 0, 3))
    return [g.mul(b, g.input(rnd(rng, (3, 4))))]


def lut_over_several_ranges(g, rng):
    """Two Sin nodes on disjoint input ranges -> one LUT over the coalesced ranges.  The ranges are the ones of the
    reference's own unit test of `LookupLayout::find_index` (crates/air/src/preprocessed.rs:581-634)."""
    g.set_lut_ranges("sin", [(-100, -50), (0, 10), (200, 210)])
    x = g.input(np.concatenate([rng.integers(-100, -49, size=20), rng.integers(0, 11, size=12)]).reshape(4, 8))
    y = g.input(np.concatenate([rng.integers(200, 211, size=9), rng.integers(-100, -49, size=7)]).reshape(4, 4))
    return [g.sin(x), g.sin(y)]


OPS = ([binary_case(op, sa, sb) for op in ("add", "mul") for sa, sb in BINARY_SHAPES]
       + [unary_case(op, s) for op in ("sin", "sqrt", "exp2", "recip") for s in UNARY_SHAPES]
       + [reduce_case("sum"), reduce_case("max")]
       + [less_than_case(s) for s in ((4, 4), (17, 3), (3, 4))]
       + [contiguous_slice, contiguous_permuted, contiguous_expanded, lut_over_several_ranges])


def run_scenario(lib, build, seed, device=0):
    """gen_trace on the device -> tables == numpy mirror -> lmn_prove -> lmn_verify -> outputs == mirror."""
    from host_graph import host_tables
    cfg = lib.default_config()
    cfg.protocol_variant = backend.VARIANT_PINNED
    ctx = backend.Context(device, cfg, lib)
    try:
        g = DeviceGraph(ctx)
        outs = build(g, np.random.default_rng(seed))
        for o in outs:
            g.output(o)
        tables, luts, bufs = g.gen_trace()
        want, vals = host_tables(g)
        assert [k for k, _, _ in tables] == sorted(want), (build.__name__, [k for k, _, _ in tables], sorted(want))
        for k, buf, n in tables:
            got = ctx.download(buf.view(0, n * want[k].shape[1] * 4)).reshape(n, -1)
            assert got.shape == want[k].shape and np.array_equal(got, want[k]), "%s: table of kind %d differs" % (build.__name__, k)
        for o in outs:
            assert np.array_equal(g.read(o).reshape(-1), vals[o.node_id]), build.__name__
        proof = ctx.prove_tables(tables, luts)
        lib.verify(proof, backend.VARIANT_PINNED)
        for b in bufs:
            b.free()
        return len(proof)
    finally:
        ctx.close()

"""Shared level-2 op checks (run against the emulation build on CPU and the HIP library on GPU)."""
import numpy as np


def check_quotient_fold_grind_ops(ctx, log):
    """accumulate_quotients / fold_line / fold_circle_into_line / grind against the oracle's restatement."""
    from oracle.channel import Blake2sChannel
    from oracle.circle import LineDomain, Coset
    from oracle.field import P, QM31
    from oracle.prover import accumulate_quotients, fold_circle_into_line, fold_line
    rng = np.random.default_rng(11)
    q = lambda: QM31(*[int(v) for v in rng.integers(0, P, size=4)])
    L = 1 << log
    cols = [rng.integers(0, P, size=L, dtype=np.uint64) for _ in range(5)]
    pts = [(q(), q()), (q(), q())]
    # columns 0..3 sampled at point 0; column 4 at point 1 then point 0 (mask [-1, 0] order)
    samples = [[(pts[0], q())] for _ in range(4)] + [[(pts[1], q()), (pts[0], q())]]
    alpha = q()
    want = accumulate_quotients(log, cols, samples, alpha)            # (L, 4)
    flat = [(c, pts.index(pt), val.v) for c, ss in enumerate(samples) for (pt, val) in ss]
    got = ctx.accumulate_quotients(cols, flat, [p[0].v + p[1].v for p in pts], alpha.v)
    assert np.array_equal(got.T.astype(np.uint64), want)
    sec = rng.integers(0, P, size=(L, 4), dtype=np.uint64)
    dom = LineDomain(Coset.half_odds(log))
    assert np.array_equal(ctx.fold_line(sec.T, alpha.v).T.astype(np.uint64), fold_line(sec, alpha, dom))
    dst = rng.integers(0, P, size=(L // 2, 4), dtype=np.uint64)
    assert np.array_equal(ctx.fold_circle_into_line(dst.T, sec.T, alpha.v).T.astype(np.uint64),
                          fold_circle_into_line(dst, sec, alpha, log))
    for variant in (0, 0x1f, 0x4, 0x10, 0x0c):   # KAT, PINNED, hashed mix_u64 alone, prefixed proof of work alone, ...
        from oracle.channel import ProtocolVariant
        ch = Blake2sChannel(ProtocolVariant(variant))
        ch.mix_u64(12345 + variant)
        nonce = ctx.lib.grind(ch.digest, 9, variant)
        c2 = Blake2sChannel(ProtocolVariant(variant))
        c2.digest = ch.digest
        assert nonce == c2.grind(9)
        assert c2.verify_pow_nonce(9, nonce) and (nonce == 0 or not c2.verify_pow_nonce(9, nonce - 1))
        if not variant & 0x10:   # KAT form of the proof of work: the digest after the mix shows the zeros
            c2.mix_u64(nonce)
            assert c2.trailing_zeros() >= 9


def check_device_trace_generation(ctx, n, seed=21):
    """Device-side `process_trace` (lmn_trace_elementwise) for the chain c = a*b; d = c + w; e = recip(d):
    rows equal the host generator's rows (luminair_amd.synthetic restates prim.rs), intermediate tensors stay
    in device memory, and the proof from device-resident rows equals the proof from host rows."""
    from luminair_amd import synthetic as syn
    rng = np.random.default_rng(seed)
    a = rng.integers(1, 2048, size=n).astype(np.int32)
    b = rng.integers(1, 2048, size=n).astype(np.int32)
    w = rng.integers(8, 2048, size=n).astype(np.int32)
    want = syn.chain_graph(n, seed)            # same draws, KAT-era multiplicities
    da, db, dw = ctx.upload(a), ctx.upload(b), ctx.upload(w)
    mul_rows, c = ctx.trace_elementwise(1, da, db, n, node_id=3, input_ids=(6, 7), num_consumers=1, input_mults=(0, 0))
    add_rows, d = ctx.trace_elementwise(0, c, dw, n, node_id=4, input_ids=(3, 8), num_consumers=1, input_mults=(-1, 0))
    rec_rows, e = ctx.trace_elementwise(2, d, None, n, node_id=5, input_ids=(4,), num_consumers=0, is_final_output=True,
                                        input_mults=(-1,))
    got = {0: ctx.download(add_rows).reshape(n, 15), 1: ctx.download(mul_rows).reshape(n, 16),
           2: ctx.download(rec_rows).reshape(n, 13)}
    for kind, rows in want:
        assert np.array_equal(got[kind], rows), kind
    cc = (a.astype(np.int64) * b) >> 12
    assert np.array_equal(ctx.download(e, np.int32), (4096 * 4096) // (cc + w))
    host = ctx.prove_tables([(k, r, len(r)) for k, r in want])
    dev = ctx.prove_tables([(0, add_rows, n), (1, mul_rows, n), (2, rec_rows, n)])
    assert dev == host
    # two nodes of one kind share a table: the second node's rows are appended at row_offset
    rows2, _ = ctx.trace_elementwise(0, da, db, n, node_id=9, input_ids=(6, 7), num_consumers=2, input_mults=(0, 0),
                                     rows=ctx.alloc(2 * n * 15 * 4), row_offset=0)
    ctx.trace_elementwise(0, db, dw, n, node_id=10, input_ids=(7, 8), num_consumers=0, is_final_output=True,
                          input_mults=(0, 0), rows=rows2, row_offset=n)
    both = ctx.download(rows2).reshape(2 * n, 15)
    assert np.array_equal(both[:n], syn.add_rows(a, b, node=9, lhs_id=6, rhs_id=7, mults=(0, 0, 2)))
    assert np.array_equal(both[n:], syn.add_rows(b, w, node=10, lhs_id=7, rhs_id=8, mults=(0, 0, 0)))
    for buf in (da, db, dw, mul_rows, c, add_rows, d, rec_rows, e, rows2):
        buf.free()


def random_pie(seed, scale=1):
    """A pie of randomly chosen components with random (ragged) row counts; multiplicities are arbitrary
    (prove does not need the logup sums to cancel), values satisfy every component's AIR."""
    rng = np.random.default_rng(1000 + seed)
    from luminair_amd import synthetic as syn
    n = lambda: int(rng.integers(1, 700)) * scale
    pool = {
        0: lambda: syn.add_rows(rng.integers(-2048, 2048, size=(k := n())), rng.integers(-2048, 2048, size=k),
                                mults=(-1, 0, 2)),
        1: lambda: syn.mul_rows(rng.integers(0, 2048, size=(k := n())), rng.integers(0, 2048, size=k), node=5,
                                mults=(0, -1, 1)),
        2: lambda: syn.recip_rows(rng.integers(4, 2048, size=n()), node=7, mults=(-1, 1)),
        5: lambda: syn.sum_reduce_rows(rng.integers(-100, 100, size=(int(rng.integers(1, 20)), int(rng.integers(1, 30)))),
                                       node=8),
        6: lambda: syn.max_reduce_rows(rng.integers(-100, 100, size=(int(rng.integers(1, 20)), int(rng.integers(1, 30)))),
                                       node=9),
        7: lambda: syn.sqrt_rows(rng.integers(1, 1 << 20, size=n()), node=10),
        8: lambda: syn.rem_rows(rng.integers(1, 1 << 16, size=(k := n())), rng.integers(1, 4096, size=k), node=11),
        15: lambda: syn.inputs_rows(rng.integers(-2048, 2048, size=n()), 12, 3),
        16: lambda: syn.contiguous_rows(rng.integers(-2048, 2048, size=n()), node=13),
    }
    kinds = sorted(rng.choice(sorted(pool), size=int(rng.integers(2, 6)), replace=False).tolist())
    tabs = [(k, pool[k]()) for k in kinds]
    luts = None
    if seed % 2 == 0:     # add a LUT op with its lookup table (and, every fourth seed, LessThan + range check)
        name = ("sin", "exp2", "log2")[seed // 2 % 3]
        lo, hi = {"sin": (-3000, 3000), "exp2": (-5000, 100), "log2": (1, 9000)}[name]
        x = rng.integers(lo, hi + 1, size=n())
        rows, counts = syn.unary_lut_rows(name, x, lo, node=14, input_id=20, mults=(0, 1))
        lut = syn.make_lut(name, lo, hi)
        kind, lk = syn._LUT_KINDS[name]
        tabs += [(kind, rows), (lk, syn.lut_lookup_rows(counts, len(lut[0])))]
        luts = {name: lut}
    if seed % 4 == 0:
        lt, counts = syn.less_than_rows(rng.integers(-4096, 4096, size=(k := n())), rng.integers(-4096, 4096, size=k),
                                        node=15)
        tabs += [(13, lt), (14, syn.range_check_lookup_rows(counts))]
    return sorted(tabs, key=lambda t: t[0]), luts


def check_device_linear_layer(ctx, n_out=20, dim=7, seed=2):
    """Mul -> SumReduce -> Add on the device (the lowering of a linear layer): rows equal synthetic.linear_layer's,
    the proof from device-resident tables equals the proof from host tables."""
    from luminair_amd import synthetic as syn
    want = dict(syn.linear_layer(n_out, dim, seed))
    rng = np.random.default_rng(seed)
    x = rng.integers(0, 2048, size=(n_out, dim)).astype(np.int32)
    w = rng.integers(0, 2048, size=(n_out, dim)).astype(np.int32)
    b = rng.integers(-2048, 2048, size=n_out).astype(np.int32)
    dx, dw, db = ctx.upload(x.reshape(-1)), ctx.upload(w.reshape(-1)), ctx.upload(b)
    n = n_out * dim
    mul_rows, prod = ctx.trace_elementwise(1, dx, dw, n, node_id=3, input_ids=(7, 8), num_consumers=1, input_mults=(0, 0))
    sum_rows, y1 = ctx.trace_sum_reduce(prod, n_out, dim, 1, node_id=4, input_id=3, num_consumers=1)
    add_rows, y = ctx.trace_elementwise(0, y1, db, n_out, node_id=5, input_ids=(4, 9), num_consumers=0,
                                        is_final_output=True, input_mults=(-1, 0))
    assert np.array_equal(ctx.download(mul_rows).reshape(n, 16), want[1])
    assert np.array_equal(ctx.download(sum_rows).reshape(n, 14), want[5])
    assert np.array_equal(ctx.download(add_rows).reshape(n_out, 15), want[0])
    assert ctx.prove_tables([(0, add_rows, n_out), (1, mul_rows, n), (5, sum_rows, n)]) == \
        ctx.prove_tables([(k, want[k], len(want[k])) for k in (0, 1, 5)])
    # a middle axis: shape (front, dim, back) = (3, 5, 4)
    t = rng.integers(-500, 500, size=(3, 5, 4)).astype(np.int32)
    dt = ctx.upload(t.reshape(-1))
    rows, out = ctx.trace_sum_reduce(dt, 3, 5, 4, node_id=11, input_id=10, num_consumers=2)
    groups = t.transpose(0, 2, 1).reshape(12, 5)          # (i, j) groups, k along the row
    assert np.array_equal(ctx.download(rows).reshape(60, 14),
                          syn.sum_reduce_rows(groups, node=11, input_id=10, input_mult=-1, out_mult=2))
    assert np.array_equal(ctx.download(out, np.int32), groups.sum(axis=1))
    for buf in (dx, dw, db, mul_rows, prod, sum_rows, y1, add_rows, y, dt, rows, out):
        buf.free()
    # reduction groups longer than a workgroup (the running value crosses block boundaries through the carry-in),
    # groups that end exactly on a block boundary, and several short groups inside one block - sum and max
    for front, dim2, back in ((2, 600, 3), (3, 256, 1), (5, 100, 2), (1, 1500, 1)):
        t = rng.integers(-2000, 2000, size=(front, dim2, back)).astype(np.int32)
        dt = ctx.upload(t.reshape(-1))
        groups = t.transpose(0, 2, 1).reshape(front * back, dim2)
        for maximum in (False, True):
            rows, out = ctx.trace_sum_reduce(dt, front, dim2, back, node_id=21, input_id=20, num_consumers=1, maximum=maximum)
            gen = syn.max_reduce_rows if maximum else syn.sum_reduce_rows
            want_rows = gen(groups, node=21, input_id=20, input_mult=-1, out_mult=1)
            assert np.array_equal(ctx.download(rows).reshape(front * back * dim2, 15 if maximum else 14), want_rows)
            assert np.array_equal(ctx.download(out, np.int32), groups.max(axis=1) if maximum else groups.sum(axis=1))
            rows.free()
            out.free()
        dt.free()


def check_device_graph(lib, device=0):
    """`DeviceGraph` (host mirror of gen_trace over the device-side process_trace kernels): the simple example
    c = a*b; d = c + w; e = c*d and a linear layer y = sum(x*w, axis=1) + b, HEAD multiplicities.  The tables
    equal the host generators' rows, and the proofs pass the verifier (the logup sums of a whole graph cancel)."""
    from luminair_amd import synthetic as syn
    from luminair_amd.graph import DeviceGraph
    cfg = lib.default_config()
    cfg.protocol_variant = backend_mod().VARIANT_PINNED
    ctx = backend_mod().Context(device, cfg, lib)
    S = 4096
    a = np.array([[1, 2], [3, 4]]) * S
    b = np.array([[10, 20], [30, 40]]) * S
    w = np.array([[-1, -1], [-1, -1]]) * S
    g = DeviceGraph(ctx)
    ta, tb, tw = g.input(a), g.input(b), g.input(w)
    tc = g.mul(ta, tb)
    td = g.add(tc, tw)
    te = g.output(g.mul(tc, td))
    tables, _, bufs = g.gen_trace()
    c = (a * b) >> 12
    d = c + w
    assert np.array_equal(g.read(te), (c * d) >> 12)
    got = {k: ctx.download(buf).reshape(n, -1) for k, buf, n in tables}
    assert sorted(got) == [0, 1, 15]
    f = lambda x: x.reshape(-1)
    assert np.array_equal(got[0], syn.add_rows(f(c), f(w), node=4, lhs_id=3, rhs_id=2, mults=(-1, -1, 1)))
    assert np.array_equal(got[1], np.concatenate([
        syn.mul_rows(f(a), f(b), node=3, lhs_id=0, rhs_id=1, mults=(-1, -1, 2)),
        syn.mul_rows(f(c), f(d), node=5, lhs_id=3, rhs_id=4, mults=(-1, -1, 0))]))
    assert np.array_equal(got[15], np.concatenate([syn.inputs_rows(f(a), 0, 1), syn.inputs_rows(f(b), 1, 1),
                                                   syn.inputs_rows(f(w), 2, 1)]))
    proof = ctx.prove_tables(tables)
    lib.verify(proof, backend_mod().VARIANT_PINNED)
    assert proof == ctx.prove_tables([(k, got[k], len(got[k])) for k in sorted(got)])
    for buf in bufs:
        buf.free()
    # linear layer: y = sum(x * w, axis=1) + bias, then 1/y on positive data
    rng = np.random.default_rng(3)
    x = rng.integers(1, 2048, size=(12, 9))
    wt = rng.integers(1, 2048, size=(12, 9))
    bias = rng.integers(1, 2048, size=12)
    g = DeviceGraph(ctx)
    tx, twt, tbias = g.input(x), g.input(wt), g.input(bias)
    ty = g.add(g.sum_reduce(g.mul(tx, twt), axis=1), tbias)
    tz = g.output(g.recip(ty))
    tables, _, bufs = g.gen_trace()
    y = ((x * wt) >> 12).sum(axis=1) + bias
    assert np.array_equal(g.read(tz), (S * S) // y)
    assert [k for k, _, _ in tables] == [0, 1, 2, 5, 15]
    proof = ctx.prove_tables(tables)
    lib.verify(proof, backend_mod().VARIANT_PINNED)
    for buf in bufs:
        buf.free()
    # the black-scholes MLP shape (BASELINE config 4) entirely on the device: Linear = expand + Mul + SumReduce + Add,
    # tanh(v) = 2 * recip(1 + exp2(v * (-2/ln 2))) - 1 with the Exp2 LUT; constants are one-element inputs expanded
    # by their consumers; the proof verifying means every multiplicity (expansion-adjusted, LUT) is right
    rng = np.random.default_rng(5)
    g = DeviceGraph(ctx)
    g.set_lut("exp2", -S, S)          # 8 193 LUT rows -> a 2^14-row Exp2Lookup table
    c_scale = g.constant(int(round(-2.0 / np.log(2.0) * S)))
    c_one, c_two, c_neg1 = g.constant(S), g.constant(2 * S), g.constant(-S)

    def linear(h, n_in, n_out):
        w = rng.integers(-128, 128, size=(n_out, n_in))
        b = rng.integers(-64, 64, size=n_out)
        prod = g.mul(g.expand(h, 0, n_out), g.input(w))
        return g.add(g.sum_reduce(prod, axis=1), g.input(b)), w, b

    def tanh(v, n):
        t = g.mul(v, g.broadcast_to(c_scale, (n,)))
        e = g.exp2(t)
        s1 = g.add(e, g.broadcast_to(c_one, (n,)))
        r = g.recip(s1)
        u = g.mul(r, g.broadcast_to(c_two, (n,)))
        return g.add(u, g.broadcast_to(c_neg1, (n,)))

    x0 = rng.integers(-2048, 2048, size=2)
    h, ref, params = g.input(x0), x0.astype(np.int64), []
    for n_in, n_out, act in ((2, 8, True), (8, 8, True), (8, 1, False)):
        y, w, b = linear(h, n_in, n_out)
        ref = ((ref[None, :] * w) >> 12).sum(axis=1) + b
        if act:
            h = tanh(y, n_out)
            tt = (ref * int(round(-2.0 / np.log(2.0) * S))) >> 12
            e = np.rint(np.exp2(tt / S) * S).astype(np.int64)
            ref = ((((S * S) // (e + S)) * 2 * S) >> 12) - S
        else:
            h = y
    g.output(h)
    tables, luts, bufs = g.gen_trace()
    assert np.array_equal(g.read(h), ref)
    assert [k for k, _, _ in tables] == [0, 1, 2, 5, 9, 10, 15]
    proof = ctx.prove_tables(tables, luts)
    lib.verify(proof, backend_mod().VARIANT_PINNED)
    assert proof == ctx.prove_tables(tables, luts)
    for buf in bufs:
        buf.free()
    ctx.close()


def backend_mod():
    from luminair_amd import backend
    return backend


def device_mlp(ctx, widths=(2, 64, 64, 1), seed=42, lut_half_range=8 * 4096):
    """BASELINE config 4's shape (2 -> 64 -> 64 -> 1 MLP with tanh, examples/black-schole-nn/src/main.rs:61-103)
    built and executed with DeviceGraph.  Returns (graph, output tensor, numpy reference of the forward pass)."""
    from luminair_amd.graph import DeviceGraph
    S = 4096
    rng = np.random.default_rng(seed)
    g = DeviceGraph(ctx)
    g.set_lut("exp2", -lut_half_range, lut_half_range)
    cs = int(round(-2.0 / np.log(2.0) * S))
    c_scale, c_one, c_two, c_neg1 = g.constant(cs), g.constant(S), g.constant(2 * S), g.constant(-S)
    x0 = rng.integers(-2048, 2048, size=widths[0])
    h, ref = g.input(x0), x0.astype(np.int64)
    for li, (n_in, n_out) in enumerate(zip(widths[:-1], widths[1:])):
        w = rng.integers(-600, 600, size=(n_out, n_in))
        b = rng.integers(-512, 512, size=n_out)
        y = g.add(g.sum_reduce(g.mul(g.expand(h, 0, n_out), g.input(w)), axis=1), g.input(b))
        ref = ((ref[None, :] * w) >> 12).sum(axis=1) + b
        if li + 2 < len(widths):
            t = g.mul(y, g.broadcast_to(c_scale, (n_out,)))
            u = g.mul(g.recip(g.add(g.exp2(t), g.broadcast_to(c_one, (n_out,)))), g.broadcast_to(c_two, (n_out,)))
            h = g.add(u, g.broadcast_to(c_neg1, (n_out,)))
            e = np.rint(np.exp2(((ref * cs) >> 12) / S) * S).astype(np.int64)
            ref = ((((S * S) // (e + S)) * 2 * S) >> 12) - S
        else:
            h = y
    g.output(h)
    return g, h, ref


def check_device_remaining_ops(lib, device=0):
    """Sqrt, Rem, LessThan (+ RangeCheckLookup table), MaxReduce and Contiguous `process_trace` on the device: rows
    equal the host generators', and a graph using all of them proves and verifies."""
    from luminair_amd import synthetic as syn
    from luminair_amd.graph import DeviceGraph
    B = backend_mod()
    cfg = lib.default_config()
    cfg.protocol_variant = B.VARIANT_PINNED
    ctx = B.Context(device, cfg, lib)
    rng = np.random.default_rng(9)
    n = 77
    a = rng.integers(1, 1 << 20, size=n).astype(np.int32)
    m = rng.integers(1, 4096, size=n).astype(np.int32)
    da, dm = ctx.upload(a), ctx.upload(m)
    rows, out = ctx.trace_elementwise(7, da, None, n, node_id=2, input_ids=(0,), num_consumers=1, input_mults=(-1,))
    want = syn.sqrt_rows(a, node=2, input_id=0, mults=(-1, 1))
    assert np.array_equal(ctx.download(rows).reshape(n, 13), want)
    s_out = ctx.download(out, np.int32)
    assert np.array_equal(s_out, want[:, 8].astype(np.int32))
    rows2, out2 = ctx.trace_elementwise(8, out, dm, n, node_id=3, input_ids=(2, 1), num_consumers=0, is_final_output=True)
    assert np.array_equal(ctx.download(rows2).reshape(n, 16), syn.rem_rows(s_out, m, node=3, lhs_id=2, rhs_id=1,
                                                                            mults=(-1, -1, 0)))
    rows3, _ = ctx.trace_elementwise(16, da, None, n, node_id=4, input_ids=(0,), num_consumers=3, input_mults=(-1,))
    assert np.array_equal(ctx.download(rows3).reshape(n, 11), syn.contiguous_rows(a, node=4, input_id=0, input_mult=-1,
                                                                                  out_mult=3))
    x = rng.integers(-4096, 4096, size=n).astype(np.int32)
    y = rng.integers(-4096, 4096, size=n).astype(np.int32)
    y[:5] = x[:5]                                   # equal operands: the borrow / diff = P case
    dx, dy = ctx.upload(x), ctx.upload(y)
    rc = ctx.upload(np.zeros(256, dtype=np.uint32))
    rows4, out4 = ctx.trace_less_than(dx, dy, n, node_id=5, input_ids=(6, 7), num_consumers=0, range_check_mult=rc,
                                      is_final_output=True)
    want_rows, want_counts = syn.less_than_rows(x, y, node=5, lhs_id=6, rhs_id=7, mults=(-1, -1, 0))
    assert np.array_equal(ctx.download(rows4).reshape(n, 22), want_rows)
    assert np.array_equal(ctx.download(rc).astype(np.int64), want_counts)
    t = rng.integers(-500, 500, size=(6, 9)).astype(np.int32)
    dt = ctx.upload(t.reshape(-1))
    rows5, out5 = ctx.trace_sum_reduce(dt, 6, 9, 1, node_id=8, input_id=9, num_consumers=2, maximum=True)
    assert np.array_equal(ctx.download(rows5).reshape(54, 15), syn.max_reduce_rows(t, node=8, input_id=9, input_mult=-1,
                                                                                   out_mult=2))
    assert np.array_equal(ctx.download(out5, np.int32), t.max(axis=1))
    for b in (da, dm, rows, out, rows2, out2, rows3, dx, dy, rc, rows4, out4, dt, rows5, out5):
        b.free()
    # one graph with all of them: z = max_j sqrt(a)_ij ; r = z % m ; flag = (r < c) ; plus a materialised view
    g = DeviceGraph(ctx)
    ta = g.input(rng.integers(1, 1 << 20, size=(10, 6)))
    tm = g.input(rng.integers(1, 4096, size=10))
    tc = g.constant(1000)
    z = g.max_reduce(g.sqrt(ta), axis=1)
    r = g.rem(z, tm)
    flag = g.output(g.less_than(r, g.broadcast_to(tc, (10,))))
    mat = g.output(g.contiguous(g.expand(tm, 0, 3)))
    tables, luts, bufs = g.gen_trace()
    assert [k for k, _, _ in tables] == [6, 7, 8, 13, 14, 15, 16]
    assert g.read(mat).shape == (3, 10) and np.array_equal(g.read(mat)[2], g.read(tm))
    proof = ctx.prove_tables(tables, luts)
    lib.verify(proof, B.VARIANT_PINNED)
    for b in bufs:
        b.free()
    ctx.close()


def check_evaluate_block(ctx, logs=((5, 6), (9, 10), (12, 13), (13, 14))):
    """lmn_op_evaluate_block == the matching rows of the full evaluation, for 2 / 4 / 8 blocks."""
    from oracle.field import P
    rng = np.random.default_rng(23)
    for log_coeffs, log_domain in logs:
        co = rng.integers(0, P, size=(3, 1 << log_coeffs), dtype=np.uint64).astype(np.uint32)
        full = ctx.evaluate(co, log_domain)
        for g in (1, 2, 3):
            S = 1 << (log_domain - g)
            for b in range(1 << g):
                got = ctx.evaluate_block(co, log_domain, g, b)
                assert np.array_equal(got, full[:, b * S:(b + 1) * S]), (log_coeffs, log_domain, g, b)


def check_device_handle_ops(ctx, log=7):
    """Every `lmn_col_*` / `lmn_tree_*` op against the oracle's restatement of the same stwo operation, on
    HBM-resident handles, and a chained interpolate -> evaluate -> commit -> quotients -> fold pipeline with ONE
    upload and ONE download of column data."""
    from oracle import fft
    from oracle.circle import Coset, LineDomain
    from oracle.field import P, QM31
    from oracle.merkle import MerkleTree
    from oracle.prover import accumulate_quotients, fold_circle_into_line, fold_line
    rng = np.random.default_rng(23)
    n = 1 << log
    q = lambda: QM31(*[int(v) for v in rng.integers(0, P, size=4)])

    # from_cpu / to_cpu / zeros / device_ptr
    ev = rng.integers(0, P, size=(3, n), dtype=np.uint64)
    h = ctx.col_from_cpu(ev)
    assert (h.ncols, h.log_size) == (3, log) and h.device_ptr
    assert np.array_equal(h.to_cpu(), ev.astype(np.uint32))
    z = ctx.col_zeros(2, 5)
    assert not z.to_cpu().any()
    z.free()

    # ColumnOps::bit_reverse_column
    br = ctx.col_from_cpu(ev).bit_reverse()
    idx = np.array([int(format(i, "0%db" % log)[::-1], 2) for i in range(n)])
    assert np.array_equal(br.to_cpu(), ev[:, idx].astype(np.uint32))
    assert np.array_equal(br.bit_reverse().to_cpu(), ev.astype(np.uint32))      # an involution
    br.free()

    # PolyOps::precompute_twiddles / interpolate / evaluate / extend / eval_at_point / evaluate_block
    ctx.precompute_twiddles(log + 2)
    h.interpolate()
    co = fft.interpolate(ev)
    assert np.array_equal(h.to_cpu(), co.astype(np.uint32))
    lde = h.evaluate(log + 1)
    want_lde = fft.evaluate(co, log + 1)
    assert np.array_equal(lde.to_cpu(), want_lde.astype(np.uint32))
    ext = h.extend(log + 2)
    want_ext = np.concatenate([co, np.zeros((3, 3 * n), dtype=np.uint64)], axis=1)
    assert np.array_equal(ext.to_cpu(), want_ext.astype(np.uint32))
    assert np.array_equal(ext.evaluate(log + 2).to_cpu(), fft.evaluate(co, log + 2).astype(np.uint32))
    pt = (q(), q())
    for c in range(3):
        assert h.eval_at_point(c, pt[0].v + pt[1].v) == fft.eval_at_point(co[c], pt).v
    for lg_b in (1, 2):
        for b in range(1 << lg_b):
            blk = h.evaluate_block(log + 1, lg_b, b)
            w = (2 * n) >> lg_b
            assert np.array_equal(blk.to_cpu(), want_lde[:, b * w:(b + 1) * w].astype(np.uint32))
            blk.free()

    # MerkleOps::commit_on_layer chain over mixed sizes, layers kept on the device
    small = ctx.col_from_cpu(rng.integers(0, P, size=(2, n // 4), dtype=np.uint64))
    tree = ctx.commit([small, lde])
    cols_host = [c for c in small.to_cpu()] + [c for c in lde.to_cpu()]
    ref = MerkleTree(cols_host)
    assert tree.root() == ref.root() and tree.log_size == log + 1
    for lg in (0, 1, log + 1):
        assert tree.layer(lg) == [bytes(x) for x in ref.layers[lg]]
    tree.free()
    small.free()

    # AccumulationOps::accumulate
    a = rng.integers(0, P, size=(4, n), dtype=np.uint64)
    b = rng.integers(0, P, size=(4, n), dtype=np.uint64)
    ha, hb = ctx.col_from_cpu(a), ctx.col_from_cpu(b)
    assert np.array_equal(ha.accumulate(hb).to_cpu(), ((a + b) % P).astype(np.uint32))

    # FriOps::fold_line / fold_circle_into_line
    alpha = q()
    dom = LineDomain(Coset.half_odds(log))
    fl = hb.fold_line(alpha.v)
    assert np.array_equal(fl.to_cpu().T.astype(np.uint64), fold_line(b.T, alpha, dom))
    dst = rng.integers(0, P, size=(4, n // 2), dtype=np.uint64)
    hd = ctx.col_from_cpu(dst)
    hd.fold_circle_into_line(hb, alpha.v)
    assert np.array_equal(hd.to_cpu().T.astype(np.uint64), fold_circle_into_line(dst.T, b.T, alpha, log))

    # FriOps::decompose: f = g + lambda * v_n (bit-reversed circle domain: +1 on the first half, -1 on the second)
    g, lam = hb.decompose()
    bq = b.astype(object)
    inv_n = pow(n, P - 2, P)
    want_lam = tuple(int((int(bq[k, :n // 2].sum()) - int(bq[k, n // 2:].sum())) * inv_n % P) for k in range(4))
    assert lam == want_lam
    want_g = b.copy()
    for k in range(4):
        want_g[k, :n // 2] = (b[k, :n // 2] + P - want_lam[k]) % P
        want_g[k, n // 2:] = (b[k, n // 2:] + want_lam[k]) % P
    assert np.array_equal(g.to_cpu(), want_g.astype(np.uint32))
    for x in (ha, hb, fl, hd, g, ext):
        x.free()

    # QuotientOps::accumulate_quotients on resident columns
    pts = [(q(), q()), (q(), q())]
    samples = [[(pts[0], q())], [(pts[0], q())], [(pts[1], q()), (pts[0], q())]]
    L = 2 * n
    want_q = accumulate_quotients(log + 1, [c for c in want_lde], samples, alpha)
    flat = [(c, pts.index(p_), val.v) for c, ss in enumerate(samples) for (p_, val) in ss]
    hq = ctx.col_accumulate_quotients([lde], flat, [p_[0].v + p_[1].v for p_ in pts], alpha.v)
    assert np.array_equal(hq.to_cpu().T.astype(np.uint64), want_q)

    # chained: (ONE upload above: `ev`) interpolate -> evaluate -> commit -> quotients -> circle fold -> line fold,
    # ONE download of the final layer
    layer = ctx.col_zeros(4, log)
    layer.fold_circle_into_line(hq, alpha.v)
    nxt = layer.fold_line(alpha.v)
    w1 = fold_circle_into_line(np.zeros((n, 4), dtype=np.uint64), want_q, alpha, log + 1)
    w2 = fold_line(w1, alpha, LineDomain(Coset.half_odds(log)))
    assert np.array_equal(nxt.to_cpu().T.astype(np.uint64), w2)
    for x in (h, lde, hq, layer, nxt):
        x.free()

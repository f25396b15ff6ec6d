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
    for variant in (0, 1):
        from oracle.channel import ProtocolVariant
        ch = Blake2sChannel(ProtocolVariant(variant))
        ch.mix_u64(12345 + variant)
        nonce = ctx.lib.grind(ch.digest, 9, variant)
        c2 = Blake2sChannel(ProtocolVariant(variant))
        c2.digest = ch.digest
        assert nonce == c2.grind(9)
        c2.mix_u64(nonce)
        assert c2.trailing_zeros() >= 9

"""Full-size checks (BASELINE.json sizes and beyond) through size-independent properties — no oracle pass needed:
shard consistency (a result over N paths equals the concatenation of the results over its shards, bit for bit),
reductions of the whole = combination of the shard reductions, idempotence, exact closed forms."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_8m_paths_shard_consistency_and_reductions(gpu):
    n, shards = 8_000_003, 8                     # config 5's global path count (+3: ragged tail)
    td = gpu.TimeDiscretization(0.0, 2, 0.5)
    par = __import__("importlib").import_module("finmath-lib-cuda-extensions_amd.parallel")

    def pipeline(bm):
        x, y = bm.getBrownianIncrement(0, 0), bm.getBrownianIncrement(1, 0)
        t = x.mult(0.3).add(1.0).discount(y, 0.25)
        return t.squared().cap(4.0).floor(0.01).addProduct(x, y).abs().sqrt().choose(t, x)

    gpu.set_fusion(True)
    try:
        whole = pipeline(gpu.BrownianMotionHip(td, 1, n, 2718))
        m_whole = whole.realizations.moments()
        full = whole.realizations.to_float32()
        parts, moments = [], []
        for r in range(shards):
            off, cnt = par.path_shard(n, shards, r)
            w = pipeline(gpu.BrownianMotionHip(td, 1, cnt, 2718, path_offset=off))
            mm = w.realizations.moments()
            moments.append([mm.sum, mm.sumsq, mm.min, mm.max])
            parts.append(w.realizations.to_float32())
    finally:
        gpu.set_fusion(False)
    cat = np.concatenate(parts)
    assert cat.size == n and (cat.view(np.uint32) == full.view(np.uint32)).all()          # bit-identical under sharding
    comb = par.combine_moments(np.array(moments)[:, None, :])[0].numpy()
    assert abs(comb[0] - m_whole.sum) <= 1e-12 * np.abs(full.astype(np.float64)).sum()
    assert abs(comb[1] - m_whole.sumsq) <= 1e-12 * m_whole.sumsq
    assert comb[2] == m_whole.min and comb[3] == m_whole.max
    # the device reduction against an independent fp64 host sum of the downloaded vector
    assert abs(m_whole.sum - full.astype(np.float64).sum()) <= 1e-12 * np.abs(full.astype(np.float64)).sum()
    assert m_whole.min == float(full.min()) and m_whole.max == float(full.max())


def test_2pow26_paths_idempotence_and_closed_forms(gpu):
    n = 1 << 26                                   # 268 MB per vector: larger than the 256 MB Infinity Cache
    x = gpu.RandomVariableHip(0.0, gpu.DeviceVector.filled(n, 3.0))
    k = gpu.RandomVariableHip(0.0, gpu.DeviceVector.from_host((np.arange(n, dtype=np.int64) % 5).astype(np.float32)))
    y = k.sub(2.0)                                # values -2,-1,0,1,2
    c1 = y.cap(1.0).floor(-1.0)
    c2 = c1.cap(1.0).floor(-1.0)                  # idempotent
    assert c1.sub(c2).abs().getMax() == 0.0
    assert y.abs().abs().sub(y.abs()).getMax() == 0.0
    m = y.realizations.moments()
    q, r = divmod(n, 5)
    assert m.sum == float(sum((v - 2) * (q + (1 if v < r else 0)) for v in range(5)))
    assert m.sumsq == float(sum((v - 2) ** 2 * (q + (1 if v < r else 0)) for v in range(5)))
    assert (m.min, m.max) == (-2.0, 2.0)
    z = x.mult(y).addProduct(x, 2.0).div(3.0)     # (3y + 6)/3 = y + 2 exactly in fp32
    assert z.sub(k).abs().getMax() == 0.0
    assert x.squared().sqrt().sub(3.0).abs().getMax() == 0.0
    gpu.purge()


def test_maximum_vector_size(gpu):
    """The largest vector the boundary accepts: 2^31 elements (the reference's kernels take `int n`)
    = 8.6 GB per vector; 32-bit float4 indexing and the grid arithmetic at their limit, on both execution tiers.  Closed
    forms: every element of filled(0.5).add(0.75).squared() is 1.5625, so Σ = n·1.5625 and Σ² = n·1.5625² exactly in fp64."""
    n = 1 << 31
    with pytest.raises(gpu.FmhipError) as e:
        gpu.DeviceVector.filled(n + 1, 0.0)
    assert e.value.code == -5
    x = gpu.DeviceVector.filled(n, 0.5)
    for tier in (gpu.JIT_OFF, gpu.JIT_SYNC):
        prev = gpu.set_jit(tier)
        try:
            p = gpu.Program(1)
            w = p.op("SQUARED", p.op("ADD_S", 0, s=0.75))
            p.output(w); p.reduce(w); p.compile()
            outs, m = p.run([[x]])
        finally:
            gpu.set_jit(prev)
        assert m[0, 0, 0] == n * 1.5625 and m[0, 0, 1] == n * 1.5625 ** 2 and m[0, 0, 2] == 1.5625 and m[0, 0, 3] == 1.5625
        tail = outs[0][0].v1s1("SUB_S", 1.5625).moments()           # every element, again through the eager path
        assert tail.sum == 0.0 and tail.min == 0.0 and tail.max == 0.0
        del outs
    del x
    gpu.purge()

"""fmhip_set_expectation_comm on the GPU: Monte-Carlo paths sharded over ranks behind an UNCHANGED caller.  One
GPU, one process here, so the second rank is played by the gather function: it returns this rank's moments together with those the
test computed beforehand for the other shard.  With the communicator set, the drop-in class's getAverage / getVariance / getMin /
getMax on the SHARD must be those of the whole vector (fp64 reassociation: 1e-13 relative; min / max exact), a Monte-Carlo price
on half the paths of each "rank" the price on all paths."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_shard_expectations_are_global_expectations(gpu, oracle):
    n = 200_002
    d = oracle.java_random_doubles(4711, n) * 3.0 - 1.0
    d[n // 2 + 17] = -0.0
    f = gpu.RandomVariableHipFactory()
    whole = f.createRandomVariable(0.0, d)
    shards = [f.createRandomVariable(0.0, d[: n // 2]), f.createRandomVariable(0.0, d[n // 2:])]
    want = {"avg": whole.getAverage(), "var": whole.getVariance(), "min": whole.getMin(), "max": whole.getMax(), "svar": whole.getSampleVariance(), "se": whole.getStandardError()}
    assert gpu.expectation_world() == (1, 0)
    try:
        for rank in (0, 1):
            other = shards[1 - rank]
            calls = []

            def gather(local, rank=rank, other=other):
                # the other rank asks for the same expectation with the same shift at the same moment: its answer, computed here without
                # a communicator (this function runs inside the engine call of THIS rank: the other shard's moments were taken before)
                calls.append(local.copy())
                theirs = pending.pop(0)
                return np.stack([local, theirs] if rank == 0 else [theirs, local])

            def theirs_for(shift):
                m = other.realizations.moments(shift)
                return np.array([m.sum, m.sumsq, m.min, m.max])

            mine = shards[rank]
            # the expectations the calls below will ask for, in order: average (shift 0); variance = average, then shifted; min; max; …
            pending = [theirs_for(0.0), theirs_for(0.0), theirs_for(want["avg"]), theirs_for(0.0), theirs_for(0.0)]
            gpu.set_expectation_comm(2, rank, gather)
            assert gpu.expectation_world() == (2, rank)
            got_avg = mine.getAverage()
            got_var = mine.getVariance()
            got_min, got_max = mine.getMin(), mine.getMax()
            gpu.set_expectation_comm(1, 0, None)
            assert len(calls) == 5 and not pending
            assert abs(got_avg - want["avg"]) <= 1e-13 * (1 + abs(want["avg"]))
            assert abs(got_var - want["var"]) <= 1e-12 * want["var"]
            assert got_min == want["min"] and got_max == want["max"]
    finally:
        gpu.set_expectation_comm(1, 0, None)
    assert abs(shards[0].getAverage() - float(np.float32(d[: n // 2]).astype(np.float64).mean())) < 1e-9        # … and local again without one


def test_a_failing_gather_is_an_error_not_a_wrong_number(gpu, oracle):
    x = gpu.RandomVariableHipFactory().createRandomVariable(0.0, oracle.java_random_doubles(5, 1000))
    def broken(local):
        raise RuntimeError("peer lost")
    try:
        gpu.set_expectation_comm(2, 0, broken)
        with pytest.raises(gpu.FmhipError):
            x.getAverage()
    finally:
        gpu.set_expectation_comm(1, 0, None)
    assert np.isfinite(x.getAverage())
    with pytest.raises(gpu.FmhipError):
        gpu.set_expectation_comm(2, 5, lambda m: m)             # rank outside the world

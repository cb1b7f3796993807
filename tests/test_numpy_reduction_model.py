"""CPU: the model of numpy's float reductions that csrc/lrc_stats.h implements on the device (per-frame ScanQuality
statistics, reference s3dis_simulator.py:276-286) IS what this numpy does: buffer chunks of 8192 elements added left
to right, each chunk by pairwise_sum (8 accumulators up to 128 elements, halving above), mean = sum / n and
std = sqrt(sum((a - mean)^2) / n) with every operation in the array's type.  If a numpy upgrade changes the scheme this
test fails here, before the GPU comparison does."""
import numpy as np
import pytest


from lidarcast.npmodel import model_mean_std, model_sum, reductions_match   # the product's own statement of the model


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_model_matches_numpy(dtype):
    rng = np.random.default_rng(3)
    for n in [1, 5, 7, 8, 9, 63, 127, 128, 129, 255, 1000, 4097, 8191, 8192, 8193, 16384, 20011, 40000, 65530, 65536, 70001]:
        for _ in range(2):
            a = (rng.random(n) * 7.5 + 0.3).astype(dtype)
            assert model_sum(a) == np.add.reduce(a), n
            m, s = model_mean_std(a)
            assert m == np.mean(a) and type(np.mean(a)) is dtype, n
            assert s == np.std(a), n
    # a slice of a larger array (what a frame is): contiguous, same scheme
    big = (rng.random(100000) * 5).astype(dtype)
    v = big[12345:12345 + 33333]
    assert model_sum(v) == np.add.reduce(v) and model_mean_std(v)[1] == np.std(v)


def test_engine_self_check_accepts_this_numpy():
    """What the simulator asks before it trusts the device statistics (lidarcast.npmodel.reductions_match)."""
    assert reductions_match() is True


def test_ragged_chunk_slot_scheme_equals_the_recursion():
    """csrc/lrc_stats.h sums the ragged tail chunk without walking pairwise_sum's recursion: it claims that all leaves
    sit on two adjacent depths, reaches the 2^E nodes of the shallower one by index arithmetic and reduces them as a
    balanced tree.  Exhaustive check of that claim: for EVERY tail length 1..8191 the bracket structure it produces is the
    recursion's."""
    def literal(n, start=0):
        if n <= 128:
            return (start, n)
        h = n // 2
        h -= h % 8
        return (literal(h, start), literal(n - h, start + h))

    def slots(n):
        if n <= 128:
            return (0, n)
        size, depth = n, 0
        while size > 128:
            h = size // 2
            h -= h % 8
            size -= h
            depth += 1
        e = depth - 1
        vals = []
        for s in range(1 << e):
            st, m = 0, n
            for lvl in range(e):
                h = m // 2
                h -= h % 8
                if (s >> (e - 1 - lvl)) & 1:
                    st, m = st + h, m - h
                else:
                    m = h
            if m <= 128:
                vals.append((st, m))
            else:
                h = m // 2
                h -= h % 8
                assert h <= 128 and m - h <= 128
                vals.append(((st, h), (st + h, m - h)))
        assert len(vals) <= 64
        while len(vals) > 1:
            vals = [(vals[i], vals[i + 1]) for i in range(0, len(vals), 2)]
        return vals[0]
    assert all(slots(n) == literal(n) for n in range(1, 8192))

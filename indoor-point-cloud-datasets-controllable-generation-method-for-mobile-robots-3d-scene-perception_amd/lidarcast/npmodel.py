"""The model of numpy's float reductions that csrc/lrc_stats.h implements on the device, and the check that the
numpy of THIS process still reduces that way.

The per-frame ScanQuality statistics (reference: s3dis_simulator.py:276-286, np.mean / np.std of 10^4..10^5 values per
frame) are computed on the device in numpy's own summation order so that they carry numpy's bits: contiguous arrays are
consumed in buffer chunks of 8192 elements whose sums are added left to right; a chunk is summed by ``pairwise_sum``
(8 accumulators over rows of 8 up to 128 elements, bracketed ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)), sequential tail; above
128 elements the halves n/2 - (n/2)%8 and the rest, recursively); mean = sum/n, std = sqrt(sum((a-mean)^2)/n), every
operation in the array's type.  That scheme is an implementation detail of numpy (verified on 2.2.x).  ``reductions_match()``
compares the model with np.mean / np.std once per process; when a numpy reduces differently the simulator takes the
range column to the host and lets that numpy reduce it -- slower, and exactly what the reference would have computed."""
import numpy as np


def _pw(a):
    n, T = len(a), a.dtype.type
    if n < 8:
        r = T(0)
        for x in a:
            r = T(r + x)
        return r
    if n <= 128:
        body = a[:n - n % 8].reshape(-1, 8)
        r = body[0].copy()
        for row in body[1:]:
            r = (r + row).astype(a.dtype)
        res = T(T(T(r[0] + r[1]) + T(r[2] + r[3])) + T(T(r[4] + r[5]) + T(r[6] + r[7])))
        for x in a[n - n % 8:]:
            res = T(res + x)
        return res
    n2 = n // 2
    n2 -= n2 % 8
    return T(_pw(a[:n2]) + _pw(a[n2:]))


def model_sum(a):
    res = None
    for i in range(0, len(a), 8192):
        c = _pw(a[i:i + 8192])
        res = c if res is None else a.dtype.type(res + c)
    return res


def model_mean_std(a):
    T = a.dtype.type
    n = T(len(a))
    mean = T(model_sum(a) / n)
    x = (a - mean).astype(a.dtype)
    x = (x * x).astype(a.dtype)
    return mean, T(np.sqrt(T(model_sum(x) / n)))


_OK = None


def reductions_match():
    """True when np.mean / np.std of this numpy equal the model bit for bit on float32 and float64 arrays that exercise
    every branch (several buffer chunks, a ragged tail, sizes around the 128-element leaf).  Evaluated once."""
    global _OK
    if _OK is None:
        rng = np.random.default_rng(11)
        ok = True
        with np.errstate(all="ignore"):
            for dtype, sizes in ((np.float32, (70001, 8193, 127)), (np.float64, (20011, 129))):
                for n in sizes:
                    a = (rng.random(n) * 7.5 + 0.3).astype(dtype)
                    m, s = model_mean_std(a)
                    ok = ok and bool(m == np.mean(a)) and bool(s == np.std(a)) and type(np.mean(a)) is dtype
        _OK = ok
    return _OK

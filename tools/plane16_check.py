#!/usr/bin/env python3
"""tools/plane16_check.py -- can the cloud rebuild read a 16-byte plane record instead of a 32-byte one?  (CPU only.)

lrc_cloud_from_prims_dev recomputes t of a known hit as the trace kernel's tri_hit does:
    T = dot(Ng, v0 - O),  den = dot(Ng, D),  t = (+-T) / |den|      dot(a,b) = fma(a.z,b.z, fma(a.y,b.y, a.x*b.x))
which needs v0 and Ng: 2 x float4 per triangle row.  A 16-byte record (Ng, d0 = dot(Ng, v0)) would halve the gather:
    T' = d0 - dot(Ng, O)
Both are the same real number; this script asks whether they are the same FLOAT32 for triangles and origins like the
benchmark's.  Exact rational arithmetic (fractions.Fraction) with one round-to-nearest-even per float32 operation, so
a difference found here is a difference on any IEEE machine."""
import struct
from fractions import Fraction

import numpy as np


def f32(x):            # round a Fraction / float to the nearest float32 (ties to even), returned as Fraction
    return Fraction(struct.unpack("f", struct.pack("f", float(x)))[0]) if not isinstance(x, Fraction) else _round(x)


def _round(q):
    if q == 0:
        return Fraction(0)
    a = np.float32(float(q))                 # float(q) is correctly rounded to double; double -> float may double-round:
    cands = [Fraction(float(np.nextafter(a, np.float32(-np.inf)))), Fraction(float(a)), Fraction(float(np.nextafter(a, np.float32(np.inf))))]
    best = min(cands, key=lambda c: (abs(c - q), int(np.float32(float(c)).view(np.uint32)) & 1))
    return best


def fma(a, b, c):
    return _round(a * b + c)


def dot(a, b):
    return fma(a[2], b[2], fma(a[1], b[1], _round(a[0] * b[0])))


rng = np.random.default_rng(0)
n, diff, worst, example = 3000, 0, 0, None
for _ in range(n):
    v = [[Fraction(float(np.float32(x))) for x in rng.uniform(0, 5, 3)]]
    e1 = [Fraction(float(np.float32(x))) for x in rng.normal(0, 0.02, 3)]
    e2 = [Fraction(float(np.float32(x))) for x in rng.normal(0, 0.02, 3)]
    v0 = v[0]
    v1 = [_round(v0[k] + e1[k]) for k in range(3)]
    v2 = [_round(v0[k] + e2[k]) for k in range(3)]
    a = [_round(v2[k] - v0[k]) for k in range(3)]
    b = [_round(v0[k] - v1[k]) for k in range(3)]
    ng = [fma(a[1], b[2], -_round(a[2] * b[1])), fma(a[2], b[0], -_round(a[0] * b[2])), fma(a[0], b[1], -_round(a[1] * b[0]))]
    o = [Fraction(float(np.float32(x))) for x in (rng.uniform(1, 4), 2.0, 1.0)]
    c = [_round(v0[k] - o[k]) for k in range(3)]
    T = dot(ng, c)
    T2 = _round(dot(ng, v0) - dot(ng, o))
    if T != T2:
        diff += 1
        rel = abs(T - T2) / abs(T) if T != 0 else 0
        if rel > worst:
            worst, example = rel, (v0, ng, o, T, T2)
print(f"T = dot(Ng, v0 - O)  vs  T' = dot(Ng, v0) - dot(Ng, O): {diff} of {n} random benchmark-like cases differ in float32; "
      f"largest relative difference {float(worst):.3e}")
if example:
    v0, ng, o, T, T2 = example
    print("counter-example (float32 values):")
    print("  v0 =", [float(x) for x in v0], " Ng =", [float(x) for x in ng], " O =", [float(x) for x in o])
    print("  T  =", float(T), " T' =", float(T2))

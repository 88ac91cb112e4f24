"""Independent numpy-float32 restatement of a few oracle functions (cross-check of the C oracle).

np.float32 scalar arithmetic is IEEE single with one rounding per operation and never fuses,
so these loops execute exactly the order written in the reference.
Pure-python loops: small cases only.
"""
import numpy as np

f32 = np.float32


def dot_scalar4(a, b):
    a = np.asarray(a, f32); b = np.asarray(b, f32)
    n = a.size; un = n & ~3
    s = f32(0)
    for i in range(0, un, 4):
        g = a[i] * b[i]
        g = g + a[i + 1] * b[i + 1]
        g = g + a[i + 2] * b[i + 2]
        g = g + a[i + 3] * b[i + 3]
        s = s + g
    for j in range(un, n):
        s = s + a[j] * b[j]
    return f32(s)


def _fma(x, y, z):
    # exact product in float64 (24x24 bits fit), one rounding of the sum to f32 when |result|
    # is far from the double-rounding danger zone; use integer-exact fraction math to be safe
    from fractions import Fraction
    r = Fraction(float(x)) * Fraction(float(y)) + Fraction(float(z))
    return _round_fraction_to_f32(r)


def _round_fraction_to_f32(r):
    from fractions import Fraction
    if r == 0:
        return f32(0)
    # float(Fraction) is correctly rounded to double; round-to-odd trick is not available, so
    # do exact f32 rounding: find neighbours
    d = float(r)
    lo = f32(d)
    # candidates: lo and its neighbours; pick nearest (ties to even)
    cands = [lo, np.nextafter(lo, f32(np.inf)), np.nextafter(lo, f32(-np.inf))]
    best = None
    for c in cands:
        err = abs(Fraction(float(c)) - r)
        if best is None or err < best[0] or (err == best[0] and (int(c.view(np.uint32)) & 1) == 0):
            best = (err, c)
    return best[1]


def dot_avx2(a, b):
    a = np.asarray(a, f32); b = np.asarray(b, f32)
    n = a.size; sn = n & ~7
    acc = [f32(0)] * 8
    for i in range(0, sn, 8):
        for l in range(8):
            acc[l] = _fma(a[i + l], b[i + l], acc[l])
    r = acc[0] + acc[1]
    for l in range(2, 8):
        r = r + acc[l]
    for j in range(sn, n):
        r = r + a[j] * b[j]
    return f32(r)


def total_key(x):
    b = int(np.asarray(x, f32).view(np.uint32))
    return (~b & 0xFFFFFFFF) if (b & 0x80000000) else (b | 0x80000000)


def brute_force(rows, q, k, deleted=None, dotfn=dot_scalar4):
    out = []
    for i in range(len(rows)):
        if deleted is not None and deleted[i]:
            continue
        d = f32(-dotfn(q, rows[i]))
        out.append((total_key(d), i, d))
    out.sort(key=lambda t: (t[0], t[1]))
    out = out[:k]
    return np.array([t[1] for t in out], np.uint32), np.array([t[2] for t in out], f32)


def calibrate(s):
    s = f32(s)
    if not np.isfinite(s):
        return f32(0)
    return f32(1) / (f32(1) + np.exp(f32(-10) * (s - f32(0.5)), dtype=f32))


def fuse_full(w, sem, ent, tag, imp, mom, acc, gs):
    nm = (f32(mom) + f32(1)) / f32(2)
    if nm > f32(0.65):
        am = min(nm * f32(1.5), f32(1))
    elif nm < f32(0.40):
        am = max(nm * f32(0.3), f32(0))
    else:
        am = nm
    if acc == 0:
        a = f32(0)
    else:
        a = min(np.log2(f32(acc) + f32(1), dtype=f32) / f32(4), f32(1))
    r = f32(w[0]) * calibrate(sem) + f32(w[1]) * calibrate(ent)
    r = r + f32(w[2]) * calibrate(tag)
    r = r + f32(w[3]) * calibrate(imp)
    r = r + f32(w[4]) * calibrate(am)
    r = r + f32(w[5]) * calibrate(a)
    r = r + f32(w[6]) * calibrate(gs)
    return f32(r) if np.isfinite(r) else f32(0)


def squared_l2(a, b):
    s = f32(0)
    for x, y in zip(np.asarray(a, f32), np.asarray(b, f32)):
        d = x - y
        s = s + d * d
    return f32(s)


def spann_distance(a, b):
    s = f32(0)
    for x, y in zip(np.asarray(a, f32), np.asarray(b, f32)):
        s = s + x * y
    return f32(1) - s

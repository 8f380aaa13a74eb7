"""Minimax polynomial for 2**f on [-1/2, 1/2] (relative error), Remez exchange in 60-digit arithmetic, then the
coefficients rounded to double and the error re-measured with the rounded coefficients evaluated by double-precision
Horner FMAs (emulated exactly with mpmath).  Prints the table k_kde's exp2 uses (abcsmc_amd/csrc/weights.hip).
    python scripts/exp2_minimax.py [degree]
    python scripts/exp2_minimax.py 6 f32 [lo hi shift]    coefficients rounded to float and an f32 Horner chain, for
        2**(g - shift) on [lo, hi] (default -0.125, 1.125, 0.5): the table of ks_exp2_f32 (split-operand kernel, whose
        argument is g = fract(X) + Y with the half folded into X)"""
import sys
import mpmath as mp

mp.mp.dps = 60
import numpy as np

deg = int(sys.argv[1]) if len(sys.argv) > 1 else 8
F32 = len(sys.argv) > 2 and sys.argv[2] == "f32"
if F32:
    a = mp.mpf(sys.argv[3]) if len(sys.argv) > 3 else mp.mpf(-1) / 8
    b = mp.mpf(sys.argv[4]) if len(sys.argv) > 4 else mp.mpf(9) / 8
    shift = mp.mpf(sys.argv[5]) if len(sys.argv) > 5 else mp.mpf(1) / 2
else:
    a, b, shift = mp.mpf(-1) / 2, mp.mpf(1) / 2, mp.mpf(0)
f = lambda x: mp.power(2, x - shift)
n = deg + 2
xs = [(a + b) / 2 + (b - a) / 2 * mp.cos(mp.pi * (n - 1 - k) / (n - 1)) for k in range(n)]
for it in range(30):
    # solve sum c_j x^j + (-1)^k E f(x_k) = f(x_k)   (relative error equi-oscillation)
    A = mp.matrix(n, n)
    rhs = mp.matrix(n, 1)
    for k, x in enumerate(xs):
        for j in range(deg + 1):
            A[k, j] = x ** j
        A[k, deg + 1] = (-1) ** k * f(x)
        rhs[k] = f(x)
    sol = mp.lu_solve(A, rhs)
    c = [sol[j] for j in range(deg + 1)]
    E = sol[deg + 1]
    err = lambda x: (mp.polyval(c[::-1], x) - f(x)) / f(x)
    # new extrema: scan
    grid = [a + (b - a) * mp.mpf(i) / 4000 for i in range(4001)]
    vals = [err(x) for x in grid]
    ext = []
    for i in range(len(grid)):
        l = vals[i - 1] if i > 0 else None
        r = vals[i + 1] if i + 1 < len(grid) else None
        v = vals[i]
        if (l is None or abs(v) >= abs(l)) and (r is None or abs(v) >= abs(r)):
            if ext and mp.sign(vals[ext[-1]]) == mp.sign(v):
                if abs(v) > abs(vals[ext[-1]]):
                    ext[-1] = i
            else:
                ext.append(i)
    if len(ext) != n:
        break
    new = [grid[i] for i in ext]
    if max(abs(new[k] - xs[k]) for k in range(n)) < mp.mpf(10) ** -12:
        xs = new
        break
    xs = new
print("degree", deg, "levelled relative error", mp.nstr(abs(E), 5))
cd = [float(np.float32(float(x))) if F32 else float(x) for x in c]
rnd = (lambda v: float(np.float32(float(v)))) if F32 else (lambda v: float(v))


def horner_double(x):
    """double-precision Horner with one rounding per FMA"""
    p = mp.mpf(cd[-1])
    for cj in cd[-2::-1]:
        p = mp.mpf(rnd(p * x + mp.mpf(cj)))
    return p


worst = 0
for i in range(20001):
    x = mp.mpf(rnd(a + (b - a) * mp.mpf(i) / 20000))
    worst = max(worst, abs((horner_double(x) - f(x)) / f(x)))
print("max relative error with %s coefficients and %s Horner:" % (("float", "float") if F32 else ("double", "double")), mp.nstr(worst, 5))
if F32:
    for lo, hi in ((-0.3, 1.3), (-0.55, 1.55)):
        w2 = 0
        for i in range(4001):
            x = mp.mpf(rnd(lo + (hi - lo) * i / 4000))
            w2 = max(w2, abs((horner_double(x) - f(x)) / f(x)))
        print("   ... on [%g, %g] (outside the fitted interval):" % (lo, hi), mp.nstr(w2, 5))
for j, v in enumerate(cd):
    print("    c%d = %s   (%s)" % (j, float.hex(v), repr(v)))

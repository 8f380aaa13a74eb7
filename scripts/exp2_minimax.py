"""Minimax polynomial for 2**f on [-1/2, 1/2] (relative error), Remez exchange in 60-digit arithmetic, then the
coefficients rounded to double and the error re-measured with the rounded coefficients evaluated by double-precision
Horner FMAs (emulated exactly with mpmath).  Prints the table k_kde's exp2 uses (abcsmc_amd/csrc/weights.hip).
    python scripts/exp2_minimax.py [degree]"""
import sys
import mpmath as mp

mp.mp.dps = 60
deg = int(sys.argv[1]) if len(sys.argv) > 1 else 8
a, b = mp.mpf(-1) / 2, mp.mpf(1) / 2
f = lambda x: mp.power(2, x)
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
cd = [float(x) for x in c]


def horner_double(x):
    """double-precision Horner with one rounding per FMA"""
    p = mp.mpf(cd[-1])
    for cj in cd[-2::-1]:
        p = mp.mpf(float(p * x + mp.mpf(cj)))
    return p


worst = 0
for i in range(20001):
    x = mp.mpf(float(a + (b - a) * mp.mpf(i) / 20000))
    worst = max(worst, abs((horner_double(x) - f(x)) / f(x)))
print("max relative error with double coefficients and double Horner:", mp.nstr(worst, 5))
for j, v in enumerate(cd):
    print("    c%d = %s   (%s)" % (j, float.hex(v), repr(v)))

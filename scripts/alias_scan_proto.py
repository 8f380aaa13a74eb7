"""Prototype (exact integer arithmetic) of the PARALLEL, VERIFIED formulation of GSL's gsl_ran_discrete_preproc that
abcsmc_amd/csrc/alias_dev.hip implements: every rounded floating-point operation of the two sequential chains -- the running
total `s += w[k]` and the serving loop's `eb -= mean - E[s]` / hand-overs -- is a map x -> 2^k floor((x + g) / 2^k) + b on a
fixed-point grid; such maps compose associatively in closed form (compose()), so the chains are prefix scans; the rounding
levels k are SPECULATED from exact (unrounded) prefix sums and every step of the result is VERIFIED with the real IEEE operation.
    python scripts/alias_scan_proto.py [trials]
Compares with the plain sequential algorithm (bit for bit) on several weight distributions and counts verification failures."""
import math
import struct
import sys

import numpy as np


# ---------------------------------------------------------------------------------------- reference (sequential, IEEE doubles)
def reference(w):
    K = len(w)
    total = 0.0
    for x in w:
        total += x
    E = [x / total for x in w]
    mean = 1.0 / K
    dK = float(K)
    smalls = [k for k in range(K) if E[k] < mean]
    bigs = [k for k in range(K) if not (E[k] < mean)]
    F = [0.0] * K
    A = [0] * K
    E = list(E)
    while smalls:
        s = smalls.pop()
        if not bigs:
            A[s] = s; F[s] = 1.0
            continue
        b = bigs.pop()
        A[s] = b
        F[s] = dK * E[s]
        d = mean - E[s]
        E[s] += d
        E[b] -= d
        if E[b] < mean:
            smalls.append(b)
        elif E[b] > mean:
            bigs.append(b)
        else:
            A[b] = b; F[b] = 1.0
    while bigs:
        b = bigs.pop()
        A[b] = b; F[b] = 1.0
    return total, F, A


# ---------------------------------------------------------------------------------------- exact helpers
def bits_of(x):
    return struct.unpack("<Q", struct.pack("<d", x))[0]


def to_int(x, e0):
    """x (double, >= 0) as an integer multiple of 2^e0 (must be exact)"""
    m, e = math.frexp(x)            # x = m 2^e, 0.5 <= m < 1
    mi = int(m * (1 << 53))
    sh = e - 53 - e0
    if sh >= 0:
        return mi << sh
    assert mi % (1 << -sh) == 0, "not on the grid"
    return mi >> -sh


def to_double(v, e0):
    return math.ldexp(float(v), e0)      # exact when v has <= 53 significant bits


def level(v):
    return max(0, v.bit_length() - 53)


def half(k):
    return (1 << (k - 1)) if k > 0 else 0


# Maps on the integers of the form  f(y) = 2^k (floor((y - t0) / P) + floor((y - t1) / P)) + b,  P = 2^(k+1), t0 <= t1 <= t0 + P:
# non-decreasing, f(y + P) = f(y) + P, two steps of 2^k per period (at t0 + tP, t1 + tP) -- exactly what a round-to-nearest-EVEN
# at level k of (y + c) is (the two thresholds per period differ by the tie rule), and closed under composition.
def rne_map(k, c, sticky=False):
    """y -> RNE at level k of (y + c); sticky: c stands for a real number slightly above the integer c (no ties possible)"""
    if k == 0:
        return (0, -c - 1, -c, 0)
    P = 1 << (k + 1)
    h = 1 << (k - 1)
    t_even = -c - (1 << k) + h                   # reaching an EVEN multiple: a tie rounds up to it
    t_odd = -c + h + (0 if sticky else 1)        # reaching an ODD multiple: a tie stays below
    return (k, t_odd - P, t_even, 0)


def apply(m, y):
    k, t0, t1, b = m
    return (((y - t0) >> (k + 1)) + ((y - t1) >> (k + 1)) << k) + b


def inv_min(m, v):
    """min { y : m(y) >= v }"""
    k, t0, t1, b = m
    P = 1 << (k + 1)
    a = -((-(v - b)) >> k)                       # ceil((v - b) / 2^k): the count N(y) must reach a
    if a & 1:
        return t0 + ((a + 1) >> 1) * P           # N(t0 + tP) = 2t - 1
    return t1 + (a >> 1) * P                     # N(t1 + tP) = 2t


def compose(m1, m2):
    """first m1 then m2"""
    k1, a0, a1, b1 = m1
    k2, c0, c1, b2 = m2
    if k2 < k1:
        return (k1, a0, a1, apply(m2, b1))       # 2^k1 N is a multiple of m2's period: m2(2^k1 N + b1) = 2^k1 N + m2(b1)
    return (k2, inv_min(m1, c0), inv_min(m1, c1), b2)


def scan_apply(maps, x0):
    """inclusive scan by composition (Hillis-Steele, to exercise associativity), applied to x0"""
    n = len(maps)
    pre = list(maps)
    step = 1
    while step < n:
        nxt = list(pre)
        for i in range(step, n):
            nxt[i] = compose(pre[i - step], pre[i])
        pre = nxt
        step *= 2
    return [apply(m, x0) for m in pre]


# ---------------------------------------------------------------------------------------- parallel formulation
class Fallback(Exception):
    pass


def total_by_scan(w, fast_scan=False):
    """the sequential sum s = fl(s + w_k) as a verified scan"""
    K = len(w)
    approx = np.cumsum(np.asarray(w, dtype=np.float64))        # (any parallel prefix sum: only the binades matter)
    tot_a = float(approx[-1])
    if not (tot_a > 0 and math.isfinite(tot_a)):
        raise Fallback("total")
    e_tot = math.frexp(tot_a)[1]                                # tot_a < 2^e_tot
    e0 = e_tot - 53 - 44                                        # grid: 44 bits below the total's ulp
    maps = []
    first = True
    for i, x in enumerate(w):
        if x < 0 or not math.isfinite(x):
            raise Fallback("weight")
        sa = float(approx[i])
        # level of the RESULT s_i on the grid
        si = int(math.ldexp(sa, -e0))
        k = level(si)
        if x == 0.0:
            maps.append(rne_map(0, 0))
            continue
        if first:
            # s_1 = w exactly: must be representable on the grid
            m, e = math.frexp(x)
            if e - 53 < e0:
                raise Fallback("first weight below the grid")
            first = False
        t = math.ldexp(x, -e0)
        ti = int(math.floor(t))
        maps.append(rne_map(k, ti, sticky=(t != ti)))
    vals = scan_apply(maps, 0) if fast_scan else None
    if vals is None:
        vals = []
        x = 0
        for m in maps:
            x = apply(m, x)
            vals.append(x)
    # verification: every step with the real addition
    prev = 0.0
    for i, x in enumerate(w):
        c = to_double(vals[i], e0)
        if vals[i].bit_length() > 53 and vals[i] % (1 << (vals[i].bit_length() - 53)):
            raise Fallback("sum: claimed value not representable")
        if prev + x != c:
            raise Fallback("sum step %d" % i)
        prev = c
    return prev


def alias_by_scan(w, tree_scan=False):
    K = len(w)
    total = total_by_scan(w)
    mean = 1.0 / K
    dK = float(K)
    E = [x / total for x in w]
    smalls = [k for k in range(K) if E[k] < mean][::-1]        # pop order
    bigs = [k for k in range(K) if not (E[k] < mean)][::-1]
    F = [1.0] * K
    A = list(range(K))
    ns, nb = len(smalls), len(bigs)
    if ns == 0 or nb == 0:
        return total, F, A
    e0 = math.frexp(mean)[1] - 53 - 1                           # g0 = ulp(mean) / 2
    MEAN = to_int(mean, e0)
    d = [mean - E[s] for s in smalls]                           # rounded, as the algorithm computes it
    dI = [to_int(x, e0) for x in d]
    VI = [to_int(E[b], e0) for b in bigs]
    for b in bigs:
        if not math.isfinite(E[b]):
            raise Fallback("E")
    D = np.cumsum(np.array(dI, dtype=object)).tolist()          # exact prefix sums (python ints)
    X = np.cumsum(np.array([v - MEAN for v in VI], dtype=object)).tolist()
    # structure: z[j] = first small (1-based) with D_i > X_j
    import bisect
    z = [bisect.bisect_right(D, X[j]) + 1 for j in range(nb)]   # ns + 1: never demoted
    # serving big of small i (1-based): 1 + #{j : X_j < D_{i-1}}
    Dm1 = [0] + D[:-1]
    jof = [1 + bisect.bisect_left(X, Dm1[i]) for i in range(ns)]
    # steps in chain order
    steps = []          # (kind, i or j, map, expect_below)
    # merge: small i at position i + jof(i) - 1, hand-over j (j -> j+1) at z_j + j
    nh = sum(1 for j in range(nb - 1) if z[j] <= ns)
    T = 0
    order = {}
    for i in range(1, ns + 1):
        if jof[i - 1] > nb:
            continue                                             # unserved: all bigs demoted before it
        order[i + jof[i - 1] - 1] = ("s", i)
    for j in range(1, nb):
        if z[j - 1] <= ns:
            order[z[j - 1] + j] = ("h", j)
    Tn = len(order)
    assert sorted(order) == list(range(1, Tn + 1)), "positions are not a permutation"
    maps = []
    meta = []
    for t in range(1, Tn + 1):
        kind, idx = order[t]
        if kind == "s":
            i = idx; j = jof[i - 1]
            star = MEAN + X[j - 1] - D[i - 1]                    # exact value after the step
            k = level(star)
            maps.append(rne_map(k, -dI[i - 1]))
            below = (z[j - 1] == i)                              # demoted right after this small
            meta.append(("s", i, j, below))
        else:
            j = idx; zz = z[j - 1]
            dd_star = D[zz - 1] - X[j - 1]                       # mean - r
            k1 = level(dd_star)
            m1 = rne_map(k1, -MEAN)                              # r -> -dd = RNE(r - mean)  (round-half-even is odd-symmetric)
            star = MEAN + X[j] - D[zz - 1]
            k2 = level(star)
            m2 = rne_map(k2, VI[j])                              # -dd -> E[b'] - dd
            maps.append(compose(m1, m2))
            below = (z[j] == zz)                                 # the new big is demoted at once
            meta.append(("h", j, j + 1, below))
    if tree_scan:
        vals = scan_apply(maps, VI[0])
    else:
        vals = []
        x = VI[0]
        for m in maps:
            x = apply(m, x)
            vals.append(x)
    # verification with real doubles
    prev = E[bigs[0]]
    last_j = 1
    for t in range(Tn):
        kind, a, b, below = meta[t]
        c = vals[t]
        if c < 0 or (c.bit_length() > 53 and c % (1 << (c.bit_length() - 53))):
            raise Fallback("claimed value not representable")
        cd = to_double(c, e0)
        if kind == "s":
            true = prev - d[a - 1]
            j = b
        else:
            dd = mean - prev
            true = E[bigs[b - 1]] - dd
            j = b
        if true != cd:
            raise Fallback("step %d (%s)" % (t, kind))
        # the last big after the last small: whichever way the comparison goes, that big ends as its own alias
        at_last_small = (a == ns) if kind == "s" else (z[a - 1] == ns)
        if not (j == nb and at_last_small):
            if below:
                if not (cd < mean):
                    raise Fallback("expected demotion at %d" % t)
            elif not (cd > mean):
                raise Fallback("expected no demotion at %d" % t)
        # outputs
        if kind == "s":
            s = smalls[a - 1]
            A[s] = bigs[b - 1]
            F[s] = dK * E[s]
        else:
            bj = bigs[a - 1]
            A[bj] = bigs[b - 1]
            F[bj] = dK * prev
        prev = cd
        last_j = j
    # a demoted last big with no successor, unserved smalls, leftover bigs: A = self, F = 1 (defaults)
    return total, F, A


def main():
    trials = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    rng = np.random.default_rng(1)
    fails = 0
    n = 0
    for t in range(trials):
        K = int(rng.choice([7, 64, 1000, 4097, 20000]))
        kind = t % 6
        if kind == 0:
            w = rng.random(K)
        elif kind == 1:
            w = np.exp(1.5 * rng.normal(size=K))
        elif kind == 2:
            w = np.exp(3.0 * rng.normal(size=K))
        elif kind == 3:
            w = rng.random(K) ** 3; w[rng.integers(0, K, max(1, K // 50))] = 0.0
        elif kind == 4:
            w = np.full(K, 1.0 / K) * (1 + 1e-9 * rng.normal(size=K))      # nearly uniform: everything close to the mean
        else:
            w = np.round(rng.random(K) * 8) / 8.0 + 0.125                   # few distinct values: exact ties abound
        w = w / np.linalg.norm(w)
        w = [float(x) for x in w]
        tot, F, A = reference(w)
        n += 1
        try:
            tot2, F2, A2 = alias_by_scan(w, tree_scan=(K <= 1000))
        except Fallback as e:
            fails += 1
            print("trial %d K=%d kind=%d: fallback (%s)" % (t, K, kind, e))
            continue
        ok = (tot == tot2) and all(bits_of(a) == bits_of(b) for a, b in zip(F, F2)) and A == A2
        print("trial %d K=%d kind=%d: %s" % (t, K, kind, "bit-exact" if ok else "MISMATCH"))
        if not ok:
            bad = [k for k in range(K) if bits_of(F[k]) != bits_of(F2[k]) or A[k] != A2[k]]
            print("   first bad entries", bad[:5], [(F[k], F2[k], A[k], A2[k]) for k in bad[:3]])
            sys.exit(1)
    print("%d trials, %d fallbacks" % (n, fails))


if __name__ == "__main__":
    main()

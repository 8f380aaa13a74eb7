"""Numpy emulation of the operand split of the weight kernel k_kde_split (abcsmc_amd/csrc/weights.hip): absolute error of
the pair dot product a.b (= error of the base-2 exponent of a term) against fp64, for the shipped split and for cheaper
ones.  Limbs: l0 = rint(4 v)/4, l1 = rint(U1 (v - l0))/U1, then bf16 roundings of what is left.  X = l0.l0' + l0.l1' + l1.l0'
is accumulated exactly (every partial sum is a multiple of 1/(4 U1) below 2^24/(4 U1)); Y = the remaining products goes
through an f32 accumulator, emulated here with ONE rounding per added product (pessimistic: the MFMA rounds less often).
For the shipped split the f32 evaluation of the term (ks_exp2: X = n + fract(X) exactly, g = fract + Y in f32, 2^g in f32,
then an exact scaling by 2^n) is emulated as well and the RELATIVE error of
2^(a.b) reported: that, not the exponent error, is what a weight inherits.
    python scripts/split_precision.py [P] [n]"""
import sys

import numpy as np

P = int(sys.argv[1]) if len(sys.argv) > 1 else 16
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1500
U1 = 1024.0 if P <= 16 else 512.0
rng = np.random.default_rng(1)
# scaled coordinates of a posterior against the previous one: N(0, 0.85) (sqrt(log2 e) / sqrt(2) proposal sigmas)
a = rng.normal(0, 0.85, (n, P))
b = rng.normal(0, 0.85, (n, P))


def bf16(v):
    f = v.astype(np.float32).view(np.uint32)
    f = (f + 0x7fff + ((f >> 16) & 1)) & 0xffff0000
    return f.view(np.float32).astype(np.float64)


def split(v, nl):
    l0 = np.rint(v * 4) / 4
    r = v - l0
    l1 = np.rint(r * U1) / U1
    r = r - l1
    L = [l0, l1]
    for _ in range(nl - 2):
        l = bf16(r)
        r = r - l
        L.append(l)
    return L


def acc_f32(x, y, acc):
    acc = acc.astype(np.float32)
    for p in range(P):
        acc = (acc.astype(np.float64) + x[:, None, p] * y[None, :, p]).astype(np.float32)
    return acc


ref = a @ b.T
full = [(3, 1), (1, 3), (2, 2), (3, 0), (0, 3), (2, 1), (1, 2), (2, 0), (0, 2), (1, 1)]      # issue order of KS_LA / KS_LB
cases = [("shipped: 4 limbs, 13 products", 4, full),
         ("without (1,3),(3,1)", 4, [p for p in full if p not in [(1, 3), (3, 1)]]),
         ("without (1,3),(3,1),(2,2)", 4, [p for p in full if p not in [(1, 3), (3, 1), (2, 2)]]),
         ("3 limbs, 9 products", 3, [(2, 2), (2, 1), (1, 2), (2, 0), (0, 2), (1, 1)]),
         ("5 limbs, 15 products", 5, [(4, 0), (0, 4)] + full)]
def term_f32(X, Y32):
    """ks_exp2 of weights.hip: n = floor(X) and fract(X) exact in f32, g = fract + Y rounded to f32, 2^g from the hardware's
    v_exp_f32 -- emulated here by the correctly rounded float of exp2(g); the hardware adds at most one more ulp
    (measured 8.2e-8 max / 2.6e-8 rms relative on [-0.3, 1.3], scripts/exp2_hw_accuracy.hip) -- then an exact scaling by 2^n"""
    Xf = X.astype(np.float32)
    assert np.array_equal(Xf.astype(np.float64), X)
    nfl = np.floor(Xf)
    g = ((Xf - nfl).astype(np.float32) + Y32).astype(np.float32)
    pv = np.exp2(g.astype(np.float64)).astype(np.float32)
    return np.ldexp(pv.astype(np.float64), nfl.astype(np.int64)), g


print("P = %d, %d x %d pairs, U1 = %g" % (P, n, n, U1))
for name, nl, pairs in cases:
    A, B = split(a, nl), split(b, nl)
    X = A[0] @ B[0].T + A[0] @ B[1].T + A[1] @ B[0].T          # exact by construction (checked below)
    Xf = acc_f32(A[1], B[0], acc_f32(A[0], B[1], acc_f32(A[0], B[0], np.zeros((n, n), np.float32))))
    assert np.array_equal(Xf.astype(np.float64), X), "X is not exact in f32"
    Ye = sum(A[i] @ B[j].T for i, j in pairs)
    Y = np.zeros((n, n), np.float32)
    for i, j in pairs:
        Y = acc_f32(A[i], B[j], Y)
    e_tr, e_all = np.abs(X + Ye - ref), np.abs(X + Y.astype(np.float64) - ref)
    print("%-32s MFMAs %2d   truncation only: rms %.2e max %.2e   with the f32 accumulator: rms %.2e max %.2e" % (
        name, (3 + len(pairs)) * ((P + 15) // 16) + 2, np.sqrt((e_tr ** 2).mean()), e_tr.max(), np.sqrt((e_all ** 2).mean()), e_all.max()))
    if name.startswith("shipped"):
        t, g = term_f32(X, Y)
        rel = np.abs(t / np.exp2(ref) - 1.0)
        print("   f32 term 2^(a.b) of the shipped split: relative error rms %.2e max %.2e mean %.2e;  g = fract + Y in [%.3f, %.3f]" % (
            np.sqrt((rel ** 2).mean()), rel.max(), (t / np.exp2(ref) - 1.0).mean(), g.min(), g.max()))

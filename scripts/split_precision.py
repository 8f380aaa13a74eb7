"""Numpy emulation of the operand split of the weight kernel k_kde_split (abcsmc_amd/csrc/weights.hip): absolute error of
the pair dot product a.b (= error of the base-2 exponent of a term) against fp64, for the shipped split and for variants.
Limbs (f16 operands of v_mfma_f32_32x32x16_f16): h0 = rint(128 v)/128, h1 = rint(2^18 (v - h0))/2^18, r2 = v - h0 - h1 (enters
scaled: (h0 2^-11).(r2' 2^11)).  X = h0.h0' is accumulated exactly (every partial sum is a multiple of 2^-14 below 2^10); Y =
h0.h1' + h1.h0' + h1.h1' + h0.r2' + r2.h0' goes through an f32 accumulator, emulated here with ONE rounding per added product
(pessimistic: the MFMA rounds less often).  Left out: h1.r2' + r2.h1' (the "cross" variant adds them) and r2.r2'.
For the shipped split the f32 evaluation of the terms (kz_slots: the 16 terms of a lane's batch share n = floor(max X),
X - n exact, the small products accumulated on top of it in the same f32 accumulator, 2^Z in f32, an f32 sum of the 16, then an
exact scaling by 2^n) is emulated as well and the RELATIVE error of the batch sums reported: that, not the exponent error, is
what a weight inherits.
    python scripts/split_precision.py [P] [n]"""
import sys

import numpy as np

P = int(sys.argv[1]) if len(sys.argv) > 1 else 16
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1500
rng = np.random.default_rng(1)
# scaled coordinates of a posterior against the previous one: N(0, 0.85) (sqrt(log2 e) / sqrt(2) proposal sigmas)
a = rng.normal(0, 0.85, (n, P))
b = rng.normal(0, 0.85, (n, P))


def f16(v):
    return v.astype(np.float16).astype(np.float64)          # IEEE binary16, round to nearest even, subnormals kept


def split(v):
    h0 = np.rint(v * 128.0) / 128.0
    r1 = v - h0
    h1 = np.rint(r1 * 2.0 ** 18) / 2.0 ** 18
    r2 = r1 - h1
    ops = [h0, h1, h0 * 2.0 ** -11, f16(r2 * 2.0 ** 11), h1 * 2.0 ** -4, f16(r2 * 2.0 ** 4)]
    for k in (0, 1, 2, 4):
        assert np.array_equal(f16(ops[k]), ops[k]), "operand %d is not exact in f16" % k
    return ops


def acc_f32(x, y, acc):
    acc = acc.astype(np.float32)
    for p in range(P):
        acc = (acc.astype(np.float64) + x[:, None, p] * y[None, :, p]).astype(np.float32)
    return acc


ref = a @ b.T
base = [(0, 1), (1, 0), (1, 1), (2, 3), (3, 2)]                  # (operand of a, operand of b'), issue order of KS_LA / KS_LB
cases = [("shipped: 3 limbs, 6 products", base),
         ("cross: + h1.r2' + r2.h1'", base + [(4, 5), (5, 4)]),
         ("without h1.h1'", [p for p in base if p != (1, 1)])]
def term_f32(X, A, B, pairs):
    """kz_slots of weights.hip: the 16 terms a lane owns of one 32 x 32 block share n = floor(max X); X - n is exact, the
    small products are then added INTO THE SAME f32 accumulator -- emulated with one rounding per product and 16-parameter
    chunk (what an MFMA step does at most) --, 2^Z from the hardware's v_exp_f32 -- emulated by the correctly rounded float of
    exp2(Z); the hardware adds at most one more ulp (measured 8.2e-8 max / 2.6e-8 rms relative on [-0.3, 1.3],
    scripts/exp2_hw_accuracy.hip) --, the 16 terms are added in f32 (four chains of four, then (p0 + p1) + (p2 + p3)) and the
    batch sum is scaled by 2^n exactly.  Returns the batch sums (rows x columns / 16) and Z."""
    Xf = X.astype(np.float32)
    assert np.array_equal(Xf.astype(np.float64), X)
    r, c = Xf.shape
    c16 = c // 16 * 16
    Xb = Xf[:, :c16].reshape(r, -1, 16)
    nfl = np.floor(Xb.max(axis=2, keepdims=True)).astype(np.float32)
    Z = (Xb - nfl).astype(np.float32)
    assert np.array_equal(Z.astype(np.float64), Xb.astype(np.float64) - nfl.astype(np.float64)), "X - n is not exact in f32"
    for i, j in pairs:
        for c0 in range(0, P, 16):
            ch = (A[i][:, c0:c0 + 16] @ B[j][:, c0:c0 + 16].T)[:, :c16].reshape(r, -1, 16)
            Z = (Z.astype(np.float64) + ch).astype(np.float32)
    e = np.exp2(Z.astype(np.float64)).astype(np.float32)
    p = [e[:, :, k] for k in range(4)]
    for i in range(4, 16):
        p[i & 3] = (p[i & 3] + e[:, :, i]).astype(np.float32)
    t = ((p[0] + p[1]).astype(np.float32) + (p[2] + p[3]).astype(np.float32)).astype(np.float32)
    return np.ldexp(t.astype(np.float64), nfl[:, :, 0].astype(np.int64)), Z


print("P = %d, %d x %d pairs" % (P, n, n))
for name, pairs in cases:
    A, B = split(a), split(b)
    X = A[0] @ B[0].T                                           # exact by construction (checked below)
    Xf = acc_f32(A[0], B[0], np.zeros((n, n), np.float32))
    assert np.array_equal(Xf.astype(np.float64), X), "X is not exact in f32"
    Ye = sum(A[i] @ B[j].T for i, j in pairs)
    Y = np.zeros((n, n), np.float32)
    for i, j in pairs:
        Y = acc_f32(A[i], B[j], Y)
    e_tr, e_all = np.abs(X + Ye - ref), np.abs(X + Y.astype(np.float64) - ref)
    print("%-32s MFMAs %2d   truncation only: rms %.2e max %.2e   with the f32 accumulator: rms %.2e max %.2e" % (
        name, (1 + len(pairs)) * ((P + 15) // 16) + 3, np.sqrt((e_tr ** 2).mean()), e_tr.max(), np.sqrt((e_all ** 2).mean()), e_all.max()))
    if name.startswith("shipped"):
        t, g = term_f32(X, A, B, pairs)
        exact = np.exp2(ref)[:, :t.shape[1] * 16].reshape(n, -1, 16).sum(axis=2)
        rel = np.abs(t / exact - 1.0)
        print("   f32 term sums (16 terms 2^(a.b) per batch) of the shipped split: relative error rms %.2e max %.2e mean %.2e;  Z = X - n + Y in [%.3f, %.3f]" % (
            np.sqrt((rel ** 2).mean()), rel.max(), (t / exact - 1.0).mean(), g.min(), g.max()))

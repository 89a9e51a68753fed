#!/usr/bin/env python3
"""One-off accuracy sweep of the device Box-Muller against the oracle's glibc normals (same Philox words):
prints the largest absolute and relative differences over m blocks.  usage: tools/check_normals.py [m]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import sabc_amd as S
from oracle import oracle as O

m = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
seed = 20241220
got = S.op_normal_pairs(seed, 5 * 10**9, m, purpose=1, it=11, k=3)
want = np.array([O.normal_pair(seed, 5 * 10**9 + i, 1, 11, 3) for i in range(m)])
d = np.abs(got - want)
rel = d / np.maximum(np.abs(want), 1e-300)
i = np.unravel_index(np.argmax(d), d.shape)
j = np.unravel_index(np.argmax(rel), rel.shape)
print("pairs", m, "max abs", d.max(), "at z =", want[i], "| max rel", rel.max(), "at z =", want[j], "| p99.99 rel", np.quantile(rel, 0.9999))
print("mean", got.mean(), "var", got.var(), "max |z|", np.abs(got).max())


def exact_pairs(words):
    """Box-Muller of the given Philox words in 50-digit decimal arithmetic (the true values)."""
    from decimal import Decimal, getcontext
    getcontext().prec = 50
    PI = Decimal("3.14159265358979323846264338327950288419716939937510582097494")

    def dsin(x):
        term, total, n = x, x, 1
        while abs(term) > Decimal(10) ** -45:
            term = -term * x * x / ((2 * n) * (2 * n + 1)); total += term; n += 1
        return total

    def dcos(x):
        term, total, n = Decimal(1), Decimal(1), 1
        while abs(term) > Decimal(10) ** -45:
            term = -term * x * x / ((2 * n - 1) * (2 * n)); total += term; n += 1
        return total
    out = []
    for w in words:
        ua = (Decimal(((w[0] & 0xFFFFF) << 32) | w[1]) + Decimal("0.5")) / Decimal(2 ** 52)
        ub = (Decimal(((w[2] & 0xFFFFF) << 32) | w[3]) + Decimal("0.5")) / Decimal(2 ** 52)
        r = (-2 * ua.ln()).sqrt()
        a = 2 * PI * ub
        if a > PI: a -= 2 * PI
        out.append((float(r * dcos(a)), float(r * dsin(a)), r * dcos(a), r * dsin(a)))
    return out


if len(sys.argv) > 2 and sys.argv[2] == "--exact":
    n = 4000
    words = [O.stream_block(seed, 5 * 10**9 + i, 1, 11, 3) for i in range(n)]
    ex = exact_pairs(words)
    worst = 0.0
    for i in range(n):
        for c in (0, 1):
            true = ex[i][2 + c]
            ulp = np.spacing(abs(ex[i][c])) if ex[i][c] != 0 else 5e-324
            e_dev = abs(float((type(true)(float(got[i, c])) - true))) / ulp
            e_orc = abs(float((type(true)(float(want[i, c])) - true))) / ulp
            worst = max(worst, e_dev)
            if e_dev > 2.0: print("pair", i, c, "z", ex[i][c], "device err ulp", e_dev, "oracle err ulp", e_orc)
    print("exact check on", n, "pairs: worst device error", worst, "ulp")

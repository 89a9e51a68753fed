"""An INDEPENDENT restatement of the reference's algorithm in vectorised NumPy -- TEST INFRASTRUCTURE, written straight from
/root/reference/src (SimulatedAnnealingABC.jl:92-137,151-227,251-402; proposals.jl:40-60,101-114; cdf_estimators.jl:23-44)
without looking at oracle/sabc_oracle.c: its own RNG (numpy PCG64), its own interpolation (np.interp), its own root finder
(scipy brentq), its own resampler (np.random.choice semantics).  It shares no code and no random stream with the oracle or
the device, so agreement between them can only be statistical -- which is exactly what makes it a second opinion on the
oracle's reading of the algorithm (tests/test_independent_numpy.py)."""
import math

import numpy as np
from scipy.optimize import brentq


def build_cdf(x):                                   # cdf_estimators.jl:23-44
    x = np.sort(x[x > 0])
    knots = np.concatenate([[0.0], x, [1.5 * x[-1]]])
    probs = np.linspace(0.0, 1.0, len(knots))
    return lambda r: np.interp(r, knots, probs, left=0.0, right=1.0)


def eps_single(ubar, v):                            # :92-95
    if ubar <= np.finfo(float).eps:
        return 0.0
    return brentq(lambda e: e * e + v * e ** 1.5 - ubar * ubar, 0.0, ubar, xtol=1e-300, rtol=1e-14)


def eps_multi(ubar, v):                             # :100-117
    s = len(ubar)
    cn = math.factorial(2 * s + 2) / (math.factorial(s + 1) * math.factorial(s + 2))
    out = np.empty(s)
    for i in range(s):
        q = ubar / ubar[i]
        num = 1 + np.sum(q ** (s / 2))
        den = cn * (s + 1) * ubar[i] ** (1 + s / 2) * np.prod(q)
        f = lambda b: (1 - math.exp(-b) * (1 + b)) / (b * (1 - math.exp(-b))) - ubar[i]
        if ubar[i] < 0.5:                         # the tilted mean falls from 1/2 at beta = 0: positive root
            lo, hi = 1e-4, 1.0 / ubar[i]          # (the literal form of :113 cancels below |beta| ~ 1e-5)
            while f(hi) > 0:
                hi *= 2
        else:                                     # mean u above 1/2 (early, hot): negative root
            lo, hi = -1.0 / (1.0 - ubar[i]) - 1.0, -1e-4
            while f(lo) < 0:
                lo *= 2
        out[i] = 1.0 / (brentq(f, lo, hi, rtol=1e-13) + v * num / den)
    return out


class SabcNumpy:
    """sabc() for a vectorised simulator `simulate(theta [n, d], rng) -> rho [n, s]` and a prior given as
    (sample(n, rng) -> [n, d], logpdf(theta [n, d]) -> [n])."""

    def __init__(self, simulate, prior_sample, prior_logpdf, n, algorithm="single_eps", v=1.0, delta=0.1, seed=0):
        self.sim, self.logpdf = simulate, prior_logpdf
        self.rng = np.random.default_rng(seed)
        self.n, self.alg = n, algorithm
        self.theta = prior_sample(n, self.rng)                        # :172-179
        self.rho = simulate(self.theta, self.rng)
        self.d, self.s = self.theta.shape[1], self.rho.shape[1]
        self.cdfs = [build_cdf(self.rho[:, j]) for j in range(self.s)]   # :187
        self.u = self.cdf(self.rho)                                   # :190-192
        self.resample(delta)                                          # :197 (rho is not permuted)
        self.eps = self.new_eps(v)                                    # :200-204
        self.n_accept, self.n_resampling, self.n_updates = 0, 1, 0

    def cdf(self, rho):
        return np.stack([self.cdfs[j](rho[:, j]) for j in range(self.s)], axis=1)

    def new_eps(self, v):                                             # :350-354
        if self.alg == "multi_eps":
            return eps_multi(self.u.mean(0), v)
        return np.array([eps_single(self.u.mean(), v)])

    def resample(self, delta):                                        # :124-137
        w = np.exp(-(self.u * delta / self.u.mean(0)).sum(1))
        idx = self.rng.choice(self.n, size=self.n, replace=True, p=w / w.sum())
        self.theta, self.u = self.theta[idx], self.u[idx]

    def propose(self, active, inactive, kind, beta, cov):             # proposals.jl
        th = self.theta[active]
        m = len(active)
        if kind == "rw":                                              # :40-43,52-55 with Sigma of :47,59
            if self.d == 1:
                return th + math.sqrt(cov) * self.rng.standard_normal((m, 1)), np.zeros(m)
            L = np.linalg.cholesky(cov)
            return th + self.rng.standard_normal((m, self.d)) @ L.T, np.zeros(m)
        if kind == "de":                                              # :101-114
            i1 = self.rng.integers(0, len(inactive), m)
            i2 = self.rng.integers(0, len(inactive), m)
            clash = i1 == i2
            while clash.any():                                        # :103-107 redraw both until distinct
                i1[clash] = self.rng.integers(0, len(inactive), clash.sum())
                i2[clash] = self.rng.integers(0, len(inactive), clash.sum())
                clash = i1 == i2
            g0 = 2.38 / math.sqrt(2 * self.d)
            gamma = g0 * (1 + 1e-5 * self.rng.standard_normal(m))
            return th + gamma[:, None] * (self.theta[inactive[i1]] - self.theta[inactive[i2]]), np.zeros(m)
        a = 2.0                                                       # stretch move, :137-148
        ip = self.rng.integers(0, len(inactive), m)
        z = ((a - 1) * self.rng.random(m) + 1) ** 2 / a
        p = self.theta[inactive[ip]]
        return p + z[:, None] * (th - p), (self.d - 1) * np.log(z)

    def update(self, n_updates, kind="de", v=1.0, delta=0.1, resample=None, beta=0.8):   # :251-402
        resample = 2 * self.n if resample is None else resample
        half = self.n // 2
        b1, b2 = np.arange(0, half), np.arange(half, self.n)
        cov = None
        for _ in range(n_updates):
            if kind == "rw":                                          # update_proposal!, :284,348
                cov = beta * np.var(self.theta[:, 0], ddof=1) if self.d == 1 else \
                    beta * (np.cov(self.theta.T, ddof=1) + 1e-8 * np.eye(self.d))
            for active, inactive in ((b1, b2), (b2, b1)):             # :304
                thp, logf = self.propose(active, inactive, kind, beta, cov)
                lpp = self.logpdf(thp)
                ok = lpp > -np.inf                                    # :314
                la = np.full(len(active), -np.inf)
                rp = np.zeros((len(active), self.s))
                up = np.zeros((len(active), self.s))
                if ok.any():
                    rp[ok] = self.sim(thp[ok], self.rng)              # :315
                    up[ok] = self.cdf(rp[ok])                         # :316
                    la[ok] = lpp[ok] - self.logpdf(self.theta[active][ok]) + \
                        ((self.u[active][ok] - up[ok]) / self.eps).sum(1) + logf[ok]   # :318-319
                acc = np.log(self.rng.random(len(active))) < la       # :324
                ia = active[acc]
                self.theta[ia], self.u[ia], self.rho[ia] = thp[acc], up[acc], rp[acc]
                self.n_accept += int(acc.sum())
            if self.n_accept >= (self.n_resampling + 1) * resample:   # :340
                self.resample(delta)
                self.n_resampling += 1
            self.eps = self.new_eps(v)                                # :350-354
            self.n_updates += 1
        return self

"""Priors as data.  The reference takes any `Distributions.Distribution` and calls `rand`,
`logpdf` and `length` on it (SimulatedAnnealingABC.jl:163,174,254,314,318); the device path
needs the prior as a descriptor: products of univariate Normal, Uniform, Exponential, LogNormal, Gamma, Beta and
truncated Normal.  Names and parametrisations follow Distributions.jl."""
from __future__ import annotations

from . import _lib


class Distribution:
    def descriptors(self):
        raise NotImplementedError

    def __len__(self):
        return len(self.descriptors())

    @property
    def univariate(self):
        return False


class Normal(Distribution):
    def __init__(self, μ=0.0, σ=1.0):
        if not σ > 0:
            raise ValueError("Normal: the standard deviation σ must be positive")
        self.μ, self.σ = float(μ), float(σ)

    def descriptors(self):
        return [(_lib.PRIOR_NORMAL, self.μ, self.σ)]

    @property
    def univariate(self):
        return True

    def __repr__(self):
        return f"Normal(μ={self.μ}, σ={self.σ})"


class Uniform(Distribution):
    def __init__(self, a=0.0, b=1.0):
        if not b > a:
            raise ValueError("Uniform: the upper bound must exceed the lower bound")
        self.a, self.b = float(a), float(b)

    def descriptors(self):
        return [(_lib.PRIOR_UNIFORM, self.a, self.b)]

    @property
    def univariate(self):
        return True

    def __repr__(self):
        return f"Uniform(a={self.a}, b={self.b})"


class Exponential(Distribution):
    """Exponential(θ) with scale θ (mean θ), as in Distributions.jl."""

    def __init__(self, θ=1.0):
        if not θ > 0:
            raise ValueError("Exponential: the scale θ must be positive")
        self.θ = float(θ)

    def descriptors(self):
        return [(_lib.PRIOR_EXPONENTIAL, self.θ, 0.0)]

    @property
    def univariate(self):
        return True

    def __repr__(self):
        return f"Exponential(θ={self.θ})"


class LogNormal(Distribution):
    """LogNormal(μ, σ): log of the variable is Normal(μ, σ)."""

    def __init__(self, μ=0.0, σ=1.0):
        if not σ > 0:
            raise ValueError("LogNormal: σ must be positive")
        self.μ, self.σ = float(μ), float(σ)

    def descriptors(self):
        return [(_lib.PRIOR_LOGNORMAL, self.μ, self.σ)]

    @property
    def univariate(self):
        return True

    def __repr__(self):
        return f"LogNormal(μ={self.μ}, σ={self.σ})"


class Gamma(Distribution):
    """Gamma(α, θ): shape α, scale θ (mean αθ), as in Distributions.jl."""

    def __init__(self, α=1.0, θ=1.0):
        if not (α > 0 and θ > 0):
            raise ValueError("Gamma: shape α and scale θ must be positive")
        self.α, self.θ = float(α), float(θ)

    def descriptors(self):
        return [(_lib.PRIOR_GAMMA, self.α, self.θ)]

    @property
    def univariate(self):
        return True

    def __repr__(self):
        return f"Gamma(α={self.α}, θ={self.θ})"


class Beta(Distribution):
    """Beta(α, β) on (0, 1)."""

    def __init__(self, α=1.0, β=1.0):
        if not (α > 0 and β > 0):
            raise ValueError("Beta: α and β must be positive")
        self.α, self.β = float(α), float(β)

    def descriptors(self):
        return [(_lib.PRIOR_BETA, self.α, self.β)]

    @property
    def univariate(self):
        return True

    def __repr__(self):
        return f"Beta(α={self.α}, β={self.β})"


class TruncatedNormal(Distribution):
    """truncated(Normal(μ, σ), lower, upper) of Distributions.jl."""

    def __init__(self, μ=0.0, σ=1.0, lower=-1.0, upper=1.0):
        if not (σ > 0 and upper > lower):
            raise ValueError("TruncatedNormal: σ must be positive and upper must exceed lower")
        self.μ, self.σ, self.lower, self.upper = float(μ), float(σ), float(lower), float(upper)

    def descriptors(self):
        return [(_lib.PRIOR_TRUNCNORMAL, self.μ, self.σ, self.lower, self.upper)]

    @property
    def univariate(self):
        return True

    def __repr__(self):
        return f"truncated(Normal(μ={self.μ}, σ={self.σ}), {self.lower}, {self.upper})"


def truncated(dist, lower, upper):
    """`truncated(Normal(μ, σ), lower, upper)` as in Distributions.jl (only Normal is supported)."""
    if not isinstance(dist, Normal):
        raise TypeError("only truncated(Normal(...), lower, upper) is available as a device prior")
    return TruncatedNormal(dist.μ, dist.σ, lower, upper)


class Product(Distribution):
    """`product_distribution([...])` of univariate components (test/runtests.jl:87-88)."""

    def __init__(self, components):
        self.components = list(components)
        if not self.components or not all(isinstance(c, Distribution) and c.univariate for c in self.components):
            raise ValueError("product_distribution needs univariate components (Normal, Uniform, Exponential, LogNormal, Gamma, Beta, TruncatedNormal)")
        if len(self.components) > _lib.MAX_PARA:
            raise ValueError(f"at most {_lib.MAX_PARA} parameters are supported")

    def descriptors(self):
        return [c.descriptors()[0] for c in self.components]

    def __repr__(self):
        return "Product(" + ", ".join(map(repr, self.components)) + ")"


class MvNormal(Distribution):
    """MvNormal(μ, Σ) of Distributions.jl: a joint Gaussian prior over all parameters (Σ symmetric positive definite)."""

    def __init__(self, μ, Σ):
        import numpy as np
        self.μ = np.atleast_1d(np.asarray(μ, dtype=np.float64))
        self.Σ = np.atleast_2d(np.asarray(Σ, dtype=np.float64))
        d = len(self.μ)
        if self.Σ.shape != (d, d) or not np.allclose(self.Σ, self.Σ.T):
            raise ValueError("MvNormal: Σ must be a symmetric d × d matrix")
        if d > _lib.MAX_JOINT_PARA:
            raise ValueError(f"MvNormal as data: at most {_lib.MAX_JOINT_PARA} parameters are supported")
        try:
            self.chol = np.linalg.cholesky(self.Σ)
        except np.linalg.LinAlgError as e:
            raise ValueError("MvNormal: Σ must be positive definite") from e

    def descriptors(self):
        # per dimension (Normal, μ_k, marginal σ_k): the library reads μ from here and Σ's factor from `chol`
        return [(_lib.PRIOR_NORMAL, float(m), float(self.Σ[k, k]) ** 0.5) for k, m in enumerate(self.μ)]

    def __repr__(self):
        return f"MvNormal(μ={self.μ.tolist()}, Σ={self.Σ.tolist()})"


class HostPrior(Distribution):
    """ANY prior, as two host callables -- the reference takes any Distributions.Distribution (SimulatedAnnealingABC.jl:151):
    `sample(ids) -> (m, d)` = rand(prior) for the particles `ids` (:174), `logpdf(Θ (m, d)) -> (m,)` = logpdf(prior, θ)
    (:314, :318; -inf outside the support).  Only together with a host-callable `f_dist`: the per-particle body is then
    already cut at the host, and the prior gate joins the simulator there (sabc_set_host_prior).  Proposal, ECDF transform,
    acceptance, reductions and resampling stay on the GPU."""
    host_prior = True

    def __init__(self, sample, logpdf, n_para, univariate=None):
        self.sample, self.logpdf, self.n_para = sample, logpdf, int(n_para)
        if not 1 <= self.n_para <= _lib.MAX_PARA:
            raise ValueError(f"1 to {_lib.MAX_PARA} parameters are supported")
        self._univariate = (self.n_para == 1) if univariate is None else bool(univariate)
        self.error = None            # an exception raised inside a callback, re-raised by the caller

    @property
    def univariate(self):
        return self._univariate

    def __len__(self):
        return self.n_para

    def descriptors(self):
        return [(_lib.PRIOR_NORMAL, 0.0, 1.0)] * self.n_para          # placeholders: the library never reads them

    def callbacks(self):
        import numpy as np
        d = self.n_para

        def sample_cb(ctx, m, ids, theta_out):
            try:
                out = np.ctypeslib.as_array(theta_out, shape=(d, m))
                x = np.asarray(self.sample(np.ctypeslib.as_array(ids, shape=(m,)).copy()), dtype=np.float64).reshape(m, d)
                out[...] = x.T
                return 0
            except Exception as e:   # never raise through the C frame; the caller re-raises it
                self.error = e
                return -1

        def logpdf_cb(ctx, m, theta, lp_out):
            try:
                th = np.ctypeslib.as_array(theta, shape=(d, m))
                out = np.ctypeslib.as_array(lp_out, shape=(m,))
                out[...] = np.asarray(self.logpdf(np.ascontiguousarray(th.T)), dtype=np.float64).reshape(m)
                return 0
            except Exception as e:
                self.error = e
                return -1

        return _lib.PRIOR_SAMPLE_FN(sample_cb), _lib.PRIOR_LOGPDF_FN(logpdf_cb)

    def __repr__(self):
        return f"HostPrior(n_para={self.n_para})"


class SourcePrior(Distribution):
    """ANY prior next to a simulator given as HIP source (`DeviceSource`): the same source defines

        __device__ void   sabc_user_prior_sample(const double *params, sabc::NormalStream &rng, double *theta_out);
        __device__ double sabc_user_prior_logpdf(const double *theta, const double *params);   // -INFINITY outside the support

    and both are compiled into the fused update kernel (sabc_config::prior_joint = 3): rand(prior) and logpdf(prior, .)
    of SimulatedAnnealingABC.jl:174,314,318 run on the device, no host round trip.  `params` is the simulator's list."""
    source_prior = True

    def __init__(self, n_para, univariate=None):
        self.n_para = int(n_para)
        if not 1 <= self.n_para <= _lib.MAX_PARA:
            raise ValueError(f"1 to {_lib.MAX_PARA} parameters are supported")
        self._univariate = (self.n_para == 1) if univariate is None else bool(univariate)

    @property
    def univariate(self):
        return self._univariate

    def __len__(self):
        return self.n_para

    def descriptors(self):
        return [(_lib.PRIOR_NORMAL, 0.0, 1.0)] * self.n_para          # placeholders: the library never reads them

    def __repr__(self):
        return f"SourcePrior(n_para={self.n_para})"


def from_scipy(dist, seed=None):
    """A scipy.stats frozen distribution (univariate: `stats.cauchy(0, 1)`; multivariate: `stats.multivariate_t(...)`,
    `stats.dirichlet(...)`, ...) or a list of univariate ones (their product) as a HostPrior.  Draws are keyed by the
    particle id and `seed` (a sharded run draws the same population as a single shard)."""
    import numpy as np
    box = {"seed": seed}            # sabc() fills in the run's seed when none was given here
    parts = list(dist) if isinstance(dist, (list, tuple)) else None
    if parts is not None:
        if not parts or not all(hasattr(p, "logpdf") and hasattr(p, "rvs") for p in parts):
            raise TypeError("a list prior needs scipy.stats frozen univariate distributions")
        d = len(parts)

        def sample(ids):
            out = np.empty((len(ids), d))
            for r, i in enumerate(ids):
                rng = np.random.default_rng([int(box["seed"] or 0), int(i)])
                out[r] = [float(p.rvs(random_state=rng)) for p in parts]
            return out

        def logpdf(th):
            return sum(np.asarray(p.logpdf(th[:, k]), dtype=np.float64) for k, p in enumerate(parts))

        hp = HostPrior(sample, logpdf, d, univariate=False)
        hp.seed_box, hp.source = box, dist
        return hp
    if not (hasattr(dist, "logpdf") and hasattr(dist, "rvs")):
        raise TypeError("prior must be a sabc_amd distribution, a scipy.stats frozen distribution or a list of them")
    probe = np.atleast_1d(np.asarray(dist.rvs(random_state=np.random.default_rng(0)), dtype=np.float64))
    d = probe.size
    univariate = np.ndim(dist.rvs(random_state=np.random.default_rng(0))) == 0

    def sample(ids):
        out = np.empty((len(ids), d))
        for r, i in enumerate(ids):
            out[r] = np.atleast_1d(dist.rvs(random_state=np.random.default_rng([int(box["seed"] or 0), int(i)])))
        return out

    def logpdf(th):
        return np.asarray(dist.logpdf(th[:, 0] if univariate else th), dtype=np.float64)

    hp = HostPrior(sample, logpdf, d, univariate=univariate)
    hp.seed_box, hp.source = box, dist
    return hp


def product_distribution(components):
    return Product(components)

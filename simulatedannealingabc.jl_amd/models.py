"""Device-coded simulators.  In the reference `f_dist` is an arbitrary Julia closure
(SimulatedAnnealingABC.jl:164,175,315); a GPU kernel needs it as data, so `sabc` takes a
`DeviceDistance` descriptor naming one of the simulators compiled into csrc/device_models.hpp.
Definitions (and the observed summaries each one is compared with) are in DESIGN.md."""
from __future__ import annotations

import numpy as np

from . import _lib


class DeviceDistance:
    model_id: int = 0
    n_stats: int = 0
    n_para: tuple = ()

    @property
    def params(self):
        raise NotImplementedError

    def __call__(self, θ, *a, **k):
        raise TypeError("a DeviceDistance is evaluated on the GPU; use SabcHandle.simulate() to call it directly")


class GaussianIID(DeviceDistance):
    """y_1..n_obs ~ Normal(θ[1], sd) with sd = θ[2] when the prior has two dimensions;
    ρ = (|obs_mean − mean(y)|,) or with `obs_m2` also |obs_m2 − mean(y.^2)|.
    Covers test/runtests.jl:35,86,128-131,167-170 and BASELINE configs 1-2."""
    model_id = _lib.MODEL_GAUSS_IID
    n_para = (1, 2)

    def __init__(self, n_obs=100, sd=1.0, obs_mean=0.0, obs_m2=None):
        self.n_obs, self.sd, self.obs_mean, self.obs_m2 = int(n_obs), float(sd), float(obs_mean), obs_m2
        self.n_stats = 1 if obs_m2 is None else 2

    @property
    def params(self):
        return [self.n_obs, self.sd, self.obs_mean, 0.0 if self.obs_m2 is None else float(self.obs_m2)]


class Gaussian2D(DeviceDistance):
    """x_1..n_obs ~ N(θ, [[1,r],[r,1]]); ρ = (‖mean − obs‖₂, |var₁+var₂ − obs|, |cov₁₂ − obs|)."""
    model_id = _lib.MODEL_GAUSS2D
    n_para = (2,)
    n_stats = 3

    def __init__(self, n_obs=50, r=0.6, obs_mean=(0.0, 0.0), obs_varsum=2.0, obs_cov=0.6):
        self.n_obs, self.r = int(n_obs), float(r)
        self.obs_mean, self.obs_varsum, self.obs_cov = tuple(map(float, obs_mean)), float(obs_varsum), float(obs_cov)

    @classmethod
    def from_observations(cls, y, r=0.6):
        y = np.asarray(y, dtype=np.float64)
        c = np.cov(y.T)
        return cls(n_obs=len(y), r=r, obs_mean=y.mean(0), obs_varsum=c[0, 0] + c[1, 1], obs_cov=c[0, 1])

    @property
    def params(self):
        return [self.n_obs, self.r, self.obs_mean[0], self.obs_mean[1], self.obs_varsum, self.obs_cov]


class GandK(DeviceDistance):
    """g-and-k: x = A + B(1 + c·tanh(g z/2))(1+z²)^k z, z~N(0,1), n_draws draws;
    ρ_j = |x_(rank_j) − obs_j| for four order statistics (1-based ranks)."""
    model_id = _lib.MODEL_GK
    n_para = (4,)
    n_stats = 4

    def __init__(self, n_draws=128, c=0.8, ranks=(16, 48, 80, 112), obs=(0.0, 0.0, 0.0, 0.0)):
        self.n_draws, self.c = int(n_draws), float(c)
        self.ranks, self.obs = tuple(int(r) for r in ranks), tuple(float(o) for o in obs)

    @property
    def params(self):
        return [self.n_draws, self.c, *self.ranks, *self.obs]


class LotkaVolterra(DeviceDistance):
    """Stochastic Lotka–Volterra by Euler–Maruyama: dX = (aX − bXY)dt + σX dW₁,
    dY = (bXY − cY)dt + σY dW₂, clamped at 0; ρ = |mean X, sd X, mean Y, sd Y − obs|."""
    model_id = _lib.MODEL_LV
    n_para = (3,)
    n_stats = 4

    def __init__(self, n_steps=256, dt=0.05, σ=0.1, x0=50.0, y0=50.0, obs=(0.0, 0.0, 0.0, 0.0)):
        self.n_steps, self.dt, self.σ, self.x0, self.y0 = int(n_steps), float(dt), float(σ), float(x0), float(y0)
        self.obs = tuple(float(o) for o in obs)

    @property
    def params(self):
        return [self.n_steps, self.dt, self.σ, self.x0, self.y0, *self.obs]


class HostDistance(DeviceDistance):
    """Any host callable as `f_dist` (SimulatedAnnealingABC.jl:164,175,315): `fn(θ, *args, **kwargs)` returning a
    scalar, tuple or vector of non-negative distances, exactly what the reference accepts.  The proposal,
    prior gate, ECDF transform, acceptance, reductions and resampling still run on the GPU; only the
    simulator is called on the host, for the proposals that passed the prior gate.

    batched=True: `fn(Θ, ...)` gets all m proposals at once (Θ is m×d, or length m for a univariate
    prior) and returns m×s.  with_ids=True: `fn(θ, particle_id, iteration, ...)`, so that a simulator can
    key its own random streams (used by the parity tests)."""
    model_id = _lib.MODEL_HOST
    params = ()

    def __init__(self, fn, n_stats, n_para, univariate, args=(), kwargs=None, batched=False, with_ids=False):
        self.fn, self.n_stats, self.n_para = fn, int(n_stats), (int(n_para),)
        self.univariate, self.args, self.kwargs = bool(univariate), tuple(args), dict(kwargs or {})
        self.batched, self.with_ids = bool(batched), bool(with_ids)
        self.error = None            # an exception raised inside the callback, re-raised by the caller

    def callback(self):
        d, s = self.n_para[0], self.n_stats

        def cb(ctx, theta, ids, m, it, rho_out):
            try:
                th = np.ctypeslib.as_array(theta, shape=(d, m))
                out = np.ctypeslib.as_array(rho_out, shape=(s, m))
                if self.batched:
                    arg = th[0].copy() if self.univariate else np.ascontiguousarray(th.T)
                    r = np.asarray(self.fn(arg, *self.args, **self.kwargs), dtype=np.float64).reshape(m, s)
                    out[...] = r.T
                else:
                    for i in range(m):
                        x = float(th[0, i]) if self.univariate else th[:, i].copy()
                        extra = (int(ids[i]), int(it)) if self.with_ids else ()
                        out[:, i] = np.atleast_1d(np.asarray(self.fn(x, *extra, *self.args, **self.kwargs), dtype=np.float64))
                return 0
            except Exception as e:   # never raise through the C frame; sabc() re-raises it
                self.error = e
                return -1

        return _lib.SIMULATE_FN(cb)


class DeviceSource(DeviceDistance):
    """`f_dist` as HIP source (SimulatedAnnealingABC.jl:164,175,315 for a simulator of your own ON the device):

        __device__ void sabc_user_simulate(const double *theta, const double *params, sabc::NormalStream &rng, double *rho_out);

    `rng.next()` / `rng.pair(z0, z1)` are N(0,1) draws, `rng.uniform_pair(u0, u1)` U(0,1) draws of the particle's
    simulation stream; `rng.for_pairs(n, [&](double z0, double z1) { ... })` hands the next n pairs to the lambda in stream
    order -- the loop to draw the bulk of a simulation with: small populations (one launch per call, a team of 4 or 16
    lanes per particle) then generate 16 or 64 pairs at a time, four blocks per lane.  The simulator must be a function of its arguments and
    its draws alone (the lanes of a quad run it side by side).  `params` is the `params` list given here.  The source is compiled at run time (hipRTC, gfx950) into
    the same fused propose -> simulate -> ECDF -> accept kernel as the built-in simulators."""
    model_id = _lib.MODEL_USER

    def __init__(self, hip_source: str, n_para: int, n_stats: int, params=()):
        self.source = str(hip_source)
        self.n_para = (int(n_para),)
        self.n_stats = int(n_stats)
        self._params = [float(p) for p in params]
        if len(self._params) > _lib.MAX_MODEL_PARAMS:
            raise ValueError(f"at most {_lib.MAX_MODEL_PARAMS} parameters")

    @property
    def params(self):
        return list(self._params)

    def compile_check(self, with_prior=False):
        """Run the compiler stage only (no GPU needed); raises SABCError with the compiler log on failure.
        with_prior: the source also defines the prior (SourcePrior)."""
        import ctypes as C
        log = C.create_string_buffer(1 << 16)
        fn = _lib.lib().sabc_op_compile_device_simulator_with_prior if with_prior else _lib.lib().sabc_op_compile_device_simulator
        rc = fn(self.source.encode(), self.n_para[0], self.n_stats, log, len(log))
        if rc:
            raise _lib.SABCError(rc, "compiling the device simulator failed:\n" + log.value.decode("utf-8", "replace"))
        return True

"""Host-side mirror of the reference's public API for the hot path: `sabc`,
`update_population!` (here `update_population_`, Python has no `!`), `SABCresult`, `SABCstate`
-- same names, keyword sets, result fields and error behaviour as
/root/reference/src/SimulatedAnnealingABC.jl:28-60,251-259,451-460.  Everything numerical
happens behind the C-ABI (include/sabc_hip.h) on the GPU; this file only marshals arguments,
exactly what the Julia wrapper in julia/ does with `ccall`.
"""
from __future__ import annotations

import logging
import math
import secrets
import sys
import time
import warnings

import numpy as np

from . import _lib
from ._lib import SABCError
from .distributions import Distribution
from .handle import SabcHandle
from .models import DeviceDistance, HostDistance
from .proposals import DifferentialEvolution, Proposal, RandomWalk

log = logging.getLogger("SimulatedAnnealingABC")

_ALGORITHMS = {"single_eps": _lib.ALG_SINGLE_EPS, "multi_eps": _lib.ALG_MULTI_EPS}


def is_logging(io=sys.stderr):
    """SimulatedAnnealingABC.jl:500."""
    import os
    return (not io.isatty()) or os.environ.get("CI") == "true"


class CdfTransform:
    """`state.cdfs_dist_prior`: ρ -> u, evaluated on the device (cdf_estimators.jl:68-70)."""

    def __init__(self, handle: SabcHandle):
        self._h = handle

    def __call__(self, ρ):
        r = np.atleast_1d(np.asarray(ρ, dtype=np.float64))
        if r.shape != (self._h.s,):
            raise ValueError(f"expected {self._h.s} distances")
        return self._h.cdf_apply(r.reshape(self._h.s, 1))[:, 0]

    def knots(self, stat):
        return self._h.cdf_knots(stat)


class SABCstate:
    """Mirror of `mutable struct SABCstate` (SimulatedAnnealingABC.jl:28-42)."""

    def __init__(self):
        self.ϵ = np.zeros(1)
        self.algorithm = "single_eps"
        self.ϵ_history, self.ρ_history, self.u_history = [], [], []
        self.cdfs_dist_prior = None
        self.n_simulation = 0
        self.n_accept = 0
        self.n_resampling = 0
        self.n_population_updates = 0


class SABCresult:
    """Mirror of `struct SABCresult{T,S}` (SimulatedAnnealingABC.jl:55-60): `population`
    (length-n vector for a univariate prior, n×d otherwise), `u` and `ρ` (n×s), `state`.
    On a sharded run these hold the LOCAL shard; `local_offset` is the first global id."""

    def __init__(self, population, u, ρ, state, handle, model, prior, seed):
        self.population, self.u, self.ρ, self.state = population, u, ρ, state
        self._handle, self._model, self._prior, self.seed = handle, model, prior, seed

    @property
    def local_offset(self):
        return self._handle.local_offset

    def __repr__(self):   # show(), SimulatedAnnealingABC.jl:65-82
        n = len(self.population)
        st = self.state
        denom = st.n_simulation - self._handle.cfg.n_particles
        acc = st.n_accept / denom if denom else float("nan")
        return (
            f"Approximate posterior sample with {n} particles:\n"
            f"  - algorithm: :{st.algorithm}\n"
            f"  - simulations used: {st.n_simulation}\n"
            f"  - number of population updates: {st.n_population_updates}\n"
            f"  - average transformed distance: {float(np.mean(self.u)):.4g}\n"
            f"  - ϵ: {np.array2string(np.asarray(st.ϵ), precision=4)}\n"
            f"  - number of population resamplings: {st.n_resampling}\n"
            f"  - acceptance rate: {acc:.4g}\n"
            "The sample can be accessed with the field `population`.\n"
            "The history of ϵ can be accessed with the field `state.ϵ_history`.\n"
            "The history of ρ can be accessed with the field `state.ρ_history`.\n"
            "The history of u can be accessed with the field `state.u_history`."
        )


def _refresh(res: SABCresult, first=False):
    """Write the device state back into the result (`.=` at SimulatedAnnealingABC.jl:387-397)."""
    h = res._handle
    th, u, rho = h.get_population()
    pop = th[0].copy() if res._prior.univariate else np.ascontiguousarray(th.T)
    if first:
        res.population, res.u, res.ρ = pop, u.T.copy(order="F"), rho.T.copy(order="F")
    else:
        res.population[...] = pop
        res.u[...] = u.T
        res.ρ[...] = rho.T
    st = res.state
    st.ϵ = h.eps
    c = h.counters
    st.n_simulation, st.n_accept = c["n_simulation"], c["n_accept"]
    st.n_resampling, st.n_population_updates = c["n_resampling"], c["n_population_updates"]
    e, uh, rh = h.history
    pre = getattr(res, "_history_prefix", ([], [], []))
    st.ϵ_history = list(pre[0]) + [row.copy() for row in e]
    st.u_history = list(pre[1]) + [row.copy() for row in uh]
    st.ρ_history = list(pre[2]) + [row.copy() for row in rh]


def _dist_env(distributed):
    """rank / world / device under torch.distributed (one process per GPU)."""
    if distributed is False:
        return 0, 1, None
    try:
        import torch.distributed as dist
    except Exception:
        return 0, 1, None
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        return dist.get_rank(), dist.get_world_size(), dist
    if distributed is True:
        raise RuntimeError("distributed=True needs an initialised torch.distributed process group")
    return 0, 1, None


def as_prior(prior, seed=None):
    """The reference takes ANY Distributions.Distribution (:151).  Here: the families that exist as data (evaluated inside the
    fused kernel), or a HostPrior / a scipy.stats frozen distribution / a list of univariate ones: rand and logpdf are host
    callbacks then, next to a host-callable f_dist or next to a device-coded simulator (which then runs as its own launch
    between the proposal and the accept kernel)."""
    if isinstance(prior, Distribution):
        return prior
    from .distributions import from_scipy
    try:
        return from_scipy(prior, seed)
    except TypeError:
        raise TypeError("prior must be Normal, Uniform, Exponential, LogNormal, Gamma, Beta, truncated(Normal), "
                        "product_distribution([...]) of those, MvNormal, HostPrior, or a scipy.stats frozen "
                        "distribution (or a list of univariate ones)") from None


def as_host_distance(f_dist, prior, args=(), kwargs=None):
    """Wrap a plain callable as a HostDistance.  Like the reference (:163-165) the number of statistics is
    found by calling it once on a draw from the prior; that call is not counted as a simulation (:213-214).
    The wrapper lives on the SABCresult it initialises (`update_population_` finds it there): nothing is cached
    per process, so a prior of another dimension can never meet a stale wrapper and no closure outlives its result."""
    rng = np.random.default_rng()
    if getattr(prior, "host_prior", False):
        θ = np.asarray(prior.sample(np.array([0], dtype=np.int64)), dtype=np.float64).reshape(-1)
        probe = f_dist(float(θ[0]) if prior.univariate else θ, *args, **(kwargs or {}))
        return HostDistance(f_dist, n_stats=len(np.atleast_1d(np.asarray(probe, dtype=np.float64))), n_para=len(prior),
                            univariate=prior.univariate, args=args, kwargs=kwargs)

    def truncnorm(a, b, c, d):
        from scipy import stats
        return float(stats.truncnorm.rvs((c - a) / b, (d - a) / b, loc=a, scale=b, random_state=rng))
    draw = {_lib.PRIOR_NORMAL: lambda a, b, c, d: rng.normal(a, b), _lib.PRIOR_UNIFORM: lambda a, b, c, d: rng.uniform(a, b),
            _lib.PRIOR_EXPONENTIAL: lambda a, b, c, d: rng.exponential(a), _lib.PRIOR_LOGNORMAL: lambda a, b, c, d: rng.lognormal(a, b),
            _lib.PRIOR_GAMMA: lambda a, b, c, d: rng.gamma(a, b), _lib.PRIOR_BETA: lambda a, b, c, d: rng.beta(a, b),
            _lib.PRIOR_TRUNCNORMAL: truncnorm}
    if getattr(prior, "chol", None) is not None:          # MvNormal
        θ = prior.μ + prior.chol @ rng.standard_normal(len(prior))
    else:
        θ = np.array([draw[desc[0]](*(tuple(desc[1:]) + (0.0, 0.0))[:4]) for desc in prior.descriptors()])
    probe = f_dist(float(θ[0]) if prior.univariate else θ, *args, **(kwargs or {}))
    hd = HostDistance(f_dist, n_stats=len(np.atleast_1d(np.asarray(probe, dtype=np.float64))), n_para=len(prior),
                      univariate=prior.univariate, args=args, kwargs=kwargs)
    return hd


def progress_stops(n_pop, show_checkpoint, show_progressbar):
    """Progress output needs the device loop to come up for air: `update_population!` is cut into several sabc_update calls.
    Returns the update counts after which a call ends: every multiple of `show_checkpoint` (:359), every step of the
    progress bar (a fiftieth of the run), and n_pop.  The cuts may fall anywhere: each call is told how many updates of the
    loop came before it (sabc_update_args::history_phase) and whether another follows (more_chunks_follow), so `ix %
    checkpoint_history` (:367) and the final push (:378-382) see the loop's own numbering and the histories are those of
    the uncut call -- `show_checkpoint` and `checkpoint_history` are independent moduli, as in the reference.  The Julia
    wrapper has the same function."""
    stops = {n_pop}
    if math.isfinite(show_checkpoint) and show_checkpoint >= 1:
        stops.update(range(int(show_checkpoint), n_pop, int(show_checkpoint)))
    if show_progressbar and n_pop > 0:
        stops.update(range(max(n_pop // 50, 1), n_pop, max(n_pop // 50, 1)))
    return sorted(stops)


def next_stop(stops, n_pop, show_checkpoint, done, rate):
    """Where the next sabc_update call ends.  A call costs ~65 us beyond its updates (a launch, two host waits), and the
    reference's progress bar redraws every 0.1 s at most (ProgressMeter's dt): a run of 999 updates at 10 us each is over before
    its first redraw, and 50 calls would be a quarter of its time.  `rate` = population updates per second as measured on the
    previous call (0: unknown): progress-bar stops closer than 0.1 s of work are passed over; multiples of `show_checkpoint`
    (their log line carries the epsilon of that very update, :359-364) and n_pop never are.  Where the cuts fall changes
    nothing in the results (history_phase / more_chunks_follow).  The Julia wrapper has the same function."""
    target = done + (max(1, int(rate * 0.1)) if rate > 0 else 1)
    chk = int(show_checkpoint) if math.isfinite(show_checkpoint) and show_checkpoint >= 1 else 0
    for s in stops:
        if s > done and (s >= target or s == n_pop or (chk and s % chk == 0)):
            return s
    return n_pop


def initialization(f_dist, prior, *args, n_particles, n_simulation, v=1.0, δ=0.1, algorithm="single_eps",
                   seed=None, device=None, distributed=None, **kwargs):
    """SimulatedAnnealingABC.jl:151-227 -> SABCresult."""
    if n_simulation < n_particles:                                            # :155-156
        raise SABCError(-1, f"`n_simulation = {n_simulation}` is too small for {n_particles} particles.")
    alg = str(algorithm).lstrip(":")
    if alg not in _ALGORITHMS:                                                # :462-464
        raise SABCError(-5, f"Argument `algorithm` must be :multi_eps or :single_eps, not `{algorithm}`!")
    prior = as_prior(prior, seed)
    if getattr(prior, "source_prior", False) and getattr(f_dist, "model_id", None) != _lib.MODEL_USER:
        raise TypeError("a SourcePrior is device code inside the simulator's HIP source: f_dist must be a DeviceSource")
    if not isinstance(f_dist, DeviceDistance):
        if not callable(f_dist):
            raise TypeError("f_dist must be a DeviceDistance or a callable f_dist(θ, *args, **kwargs)")
        f_dist = as_host_distance(f_dist, prior, args, kwargs)               # :163-165 shape probe
        args, kwargs = (), {}
    if args or kwargs:
        raise TypeError("a DeviceDistance takes its data at construction; extra args/kwargs are not forwarded")
    if len(prior) not in f_dist.n_para:
        raise ValueError(f"{type(f_dist).__name__} needs a prior with {f_dist.n_para} parameters, got {len(prior)}")
    log.info("Initialization for '%s'", alg)                                  # :158
    rank, world, dist = _dist_env(distributed)
    if device is None:
        import os
        device = int(os.environ.get("LOCAL_RANK", "0")) if world > 1 else 0
    if seed is None:
        seed = secrets.randbits(63)
        if dist is not None:   # every shard must use the same Philox key
            import torch
            t = torch.tensor([seed], dtype=torch.int64)
            if dist.get_backend() == "nccl":
                t = t.cuda(device)
            dist.broadcast(t, 0)
            seed = int(t.item())
    if getattr(prior, "seed_box", None) is not None and prior.seed_box["seed"] is None:
        prior.seed_box["seed"] = seed % (1 << 62)        # a scipy prior draws with the run's seed
    h = SabcHandle(n_particles=n_particles, model=f_dist, prior=prior, algorithm=_ALGORITHMS[alg], v=v, delta=δ,
                   seed=seed, device=device, rank=rank, world=world)
    if dist is not None:
        from .dist import install_collectives
        install_collectives(h, device)
    h.initialize(n_simulation)
    state = SABCstate()
    state.algorithm = alg
    state.cdfs_dist_prior = CdfTransform(h)
    res = SABCresult(None, None, None, state, h, f_dist, prior, seed)
    _refresh(res, first=True)
    return res


def update_population_(population_state: SABCresult, f_dist, prior, *args, n_simulation, v=1.0, δ=0.1,
                       proposal: Proposal = None, resample=None, checkpoint_history=1, show_progressbar=None,
                       show_checkpoint=None, **kwargs):
    """`update_population!` (SimulatedAnnealingABC.jl:251-402): updates the particles with
    `n_simulation` simulations, mutates `population_state` and returns it."""
    if not v > 0:
        raise SABCError(-3, "Annealing speed `v` must be positive.")                 # :261
    if not δ > 0:
        raise SABCError(-4, "Resamping intensity `δ` must be positive.")            # :262
    res = population_state
    if not isinstance(f_dist, DeviceDistance) and callable(f_dist):
        if not (isinstance(res._model, HostDistance) and res._model.fn is f_dist):
            raise ValueError("f_dist differs from the one this SABCresult was initialised with")
        res._model.args, res._model.kwargs = tuple(args), dict(kwargs)      # forwarded to f_dist, like :315
        f_dist, args, kwargs = res._model, (), {}
    if args or kwargs:
        raise TypeError("a DeviceDistance takes its data at construction; extra args/kwargs are not forwarded")
    if getattr(res._prior, "host_prior", False):
        # a host-callback prior is code, not data: it has to be the object (or the scipy distribution) of the initialisation
        if prior is not res._prior and prior is not getattr(res._prior, "source", None):
            raise ValueError("prior differs from the one this SABCresult was initialised with")
        prior = res._prior
    if f_dist is not res._model or prior is not res._prior:
        if type(f_dist) is not type(res._model) or list(f_dist.params) != list(res._model.params) or \
                getattr(f_dist, "source", None) != getattr(res._model, "source", None) or \
                getattr(f_dist, "fn", None) is not getattr(res._model, "fn", None) or \
                prior.descriptors() != res._prior.descriptors() or type(prior) is not type(res._prior) or \
                not np.array_equal(np.asarray(getattr(prior, "chol", 0.0)), np.asarray(getattr(res._prior, "chol", 0.0))):
            raise ValueError("f_dist / prior differ from the ones this SABCresult was initialised with")
    h = res._handle
    if proposal is None:
        proposal = DifferentialEvolution(n_para=len(prior))                           # :254
    if not isinstance(proposal, Proposal):
        raise TypeError("proposal must be RandomWalk, DifferentialEvolution or StretchMove")
    n_global = h.cfg.n_particles
    if resample is None:
        resample = 2 * n_global                                                        # :255
    if show_progressbar is None:
        show_progressbar = not is_logging(sys.stderr)                                  # :257
    if show_checkpoint is None:
        show_checkpoint = 100 if is_logging(sys.stderr) else math.inf                  # :258
    # the reference starts from the arrays held by the result (:264-267): push them to the device
    th = res.population.reshape(1, -1) if res._prior.univariate else np.ascontiguousarray(res.population.T)
    h.set_population(th, np.ascontiguousarray(res.u.T), np.ascontiguousarray(res.ρ.T))

    n_pop = n_simulation // n_global                                                   # :275
    if n_pop > 0 and int(checkpoint_history) == 0:
        raise ZeroDivisionError("integer division or modulo by zero: `ix % checkpoint_history`")   # :367 (a DivideError there)
    pbar = None
    if show_progressbar and n_pop > 0:
        try:
            from tqdm import tqdm
            pbar = tqdm(total=n_pop, desc=f"{n_pop} population updates:", file=sys.stderr)   # :290-291
        except ImportError:
            pbar = None
    done, t0 = 0, time.time()
    stops, rate, first = progress_stops(n_pop, show_checkpoint, pbar is not None), 0.0, True
    while first or done < n_pop:                          # (n_pop = 0: one call, a top-up below one update is a no-op, :275)
        first = False
        stop = next_stop(stops, n_pop, show_checkpoint, done, rate)
        todo = stop - done
        budget = todo * n_global if n_pop > 0 else n_simulation
        t_call = time.perf_counter()
        h.update(n_simulation=budget, proposal=proposal, v=v, delta=δ, resample=resample,
                 checkpoint_history=checkpoint_history, history_phase=done, more_chunks_follow=stop < n_pop)
        rate = todo / max(time.perf_counter() - t_call, 1e-9)
        done = stop
        if pbar is not None:
            pbar.update(todo)
            pbar.set_postfix_str(f"ϵ={np.array2string(h.eps, precision=4)}")                  # :292,374
        if done > 0 and math.isfinite(show_checkpoint) and show_checkpoint >= 1 and done % int(show_checkpoint) == 0:
            eta = (time.time() - t0) / done * (n_pop - done)                              # :359-364
            log.info("Update %d of %d. ϵ: %s, ETA: %.0f s", done, n_pop, np.array2string(h.eps, precision=4), eta)
    if pbar is not None:
        pbar.close()
    if isinstance(proposal, RandomWalk):
        sg = h.proposal_sigma
        proposal.Σ = float(sg[0, 0]) if len(prior) == 1 else sg                       # rw.Σ, proposals.jl:47,59
    _refresh(res)
    log.info("All particles have been updated %d times.", n_pop)                      # :399
    return res


def save_result(path, res: SABCresult):
    """Serialise a result, ECDF knots included, so that `update_population_` can resume from it in
    another process (the reference only resumes in memory, SimulatedAnnealingABC.jl:264-271)."""
    st, h = res.state, res._handle
    knots = [st.cdfs_dist_prior.knots(j) for j in range(h.s)]
    np.savez_compressed(
        path, population=res.population, u=res.u, rho=res.ρ, eps=np.asarray(st.ϵ), algorithm=st.algorithm,
        eps_history=np.array(st.ϵ_history), u_history=np.array(st.u_history), rho_history=np.array(st.ρ_history),
        counters=np.array([st.n_simulation, st.n_accept, st.n_resampling, st.n_population_updates], dtype=np.int64),
        seed=np.uint64(res.seed), n_particles=np.int64(h.cfg.n_particles), knot_len=np.array([len(k) for k in knots]),
        **{f"knots_{j}": k for j, k in enumerate(knots)})


def load_result(path, f_dist, prior, device=0) -> SABCresult:
    """Rebuild a device-resident SABCresult from `save_result` output (single shard)."""
    z = np.load(path if str(path).endswith(".npz") else str(path) + ".npz", allow_pickle=False)
    alg = str(z["algorithm"])
    n = int(z["n_particles"])
    prior = as_prior(prior, int(z["seed"]) % (1 << 62))
    if not isinstance(f_dist, DeviceDistance) and callable(f_dist):
        f_dist = as_host_distance(f_dist, prior)
    h = SabcHandle(n_particles=n, model=f_dist, prior=prior, algorithm=_ALGORITHMS[alg], seed=int(z["seed"]), device=device)
    pop = z["population"]
    th = pop.reshape(1, -1) if prior.univariate else np.ascontiguousarray(pop.T)
    for j in range(h.s):
        h.set_cdf_knots(j, z[f"knots_{j}"])
    h.set_population(th, np.ascontiguousarray(z["u"].T), np.ascontiguousarray(z["rho"].T))
    h.set_eps(z["eps"])
    h.set_counters(*[int(c) for c in z["counters"]])
    state = SABCstate()
    state.algorithm = alg
    state.cdfs_dist_prior = CdfTransform(h)
    res = SABCresult(None, None, None, state, h, f_dist, prior, int(z["seed"]))
    _refresh(res, first=True)
    # the handle starts with an empty history: keep the stored one in front of what later updates append
    res._history_prefix = ([r.copy() for r in z["eps_history"]], [r.copy() for r in z["u_history"]],
                           [r.copy() for r in z["rho_history"]])
    st = res.state
    st.ϵ_history, st.u_history, st.ρ_history = (list(x) for x in res._history_prefix)
    return res


def sabc(f_dist, prior, *args, n_particles=100, n_simulation=10_000, algorithm="single_eps", proposal: Proposal = None,
         resample=None, v=1.0, δ=0.1, checkpoint_history=1, show_progressbar=None, show_checkpoint=None,
         seed=None, device=None, distributed=None, **kwargs):
    """Simulated Annealing ABC (SimulatedAnnealingABC.jl:451-492).  Same keywords as the
    reference plus `seed` (Philox key; random if omitted), `device` and `distributed`."""
    alg = str(algorithm).lstrip(":")
    if alg not in _ALGORITHMS:                                                        # :462-464
        raise SABCError(-5, f"Argument `algorithm` must be :multi_eps or :single_eps, not `{algorithm}`!")
    prior = as_prior(prior, seed)
    if proposal is None:
        proposal = DifferentialEvolution(n_para=len(prior))                           # :454
    if resample is None:
        resample = 2 * n_particles                                                     # :455
    population_state = initialization(f_dist, prior, *args, n_particles=n_particles, n_simulation=n_simulation,
                                      v=v, δ=δ, algorithm=alg, seed=seed, device=device, distributed=distributed,
                                      **kwargs)                                       # :470-473
    n_sim_remaining = n_simulation - population_state.state.n_simulation             # :478
    if n_sim_remaining < n_particles:
        warnings.warn("`n_simulation` too small to update all particles!")           # :479
    update_population_(population_state, f_dist, prior, *args, n_simulation=n_sim_remaining, resample=resample,
                       proposal=proposal, v=v, δ=δ, checkpoint_history=checkpoint_history,
                       show_progressbar=show_progressbar, show_checkpoint=show_checkpoint, **kwargs)   # :481-489
    return population_state

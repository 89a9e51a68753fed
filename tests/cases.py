"""Shared case table: each case describes one (model, prior, algorithm, proposal) combination
and can be instantiated on the oracle and on the HIP engine with the same seed."""
import os

import numpy as np

# the seed of every parity run; tools/seed_sweep.sh repeats the device-vs-oracle suites with other values (SABC_TEST_SEED)
SEED = int(os.environ.get("SABC_TEST_SEED", "20241220"))


def y_obs_mean():
    return float(np.random.default_rng(SEED).normal(1.5, 1.0, 100).mean())


# name -> dict(model=(kind, kwargs), prior=[(kind, a, b)...], d, s)
MODELS = {
    # test/runtests.jl:35-36
    "gauss1_uniform": dict(model=("GaussianIID", dict(n_obs=100, sd=1.0, obs_mean=0.0)),
                           prior=[("U", -10.0, 10.0)], s=1),
    # BASELINE configs 1-2
    "gauss1_cfg2": dict(model=("GaussianIID", dict(n_obs=100, sd=1.0, obs_mean=y_obs_mean())),
                        prior=[("N", 0.0, 2.0)], s=1),
    # a short simulation (20 draws): what a host callable transcribing the device's simulator can afford (tests of the host-f_dist path)
    "gauss1_small": dict(model=("GaussianIID", dict(n_obs=20, sd=1.0, obs_mean=1.4)), prior=[("N", 0.0, 2.0)], s=1),
    # test/runtests.jl:86-88
    "gauss2_meansd": dict(model=("GaussianIID", dict(n_obs=100, sd=1.0, obs_mean=0.0)),
                          prior=[("N", 0.0, 1.0), ("U", 0.0, 1.0)], s=1),
    # test/runtests.jl:125-131
    "gauss1_2stats": dict(model=("GaussianIID", dict(n_obs=10, sd=1.0, obs_mean=0.0, obs_m2=1.0)),
                          prior=[("N", 0.0, 1.0)], s=2),
    # test/runtests.jl:163-170
    "gauss2_2stats": dict(model=("GaussianIID", dict(n_obs=10, sd=1.0, obs_mean=0.0, obs_m2=1.0)),
                          prior=[("N", 0.0, 1.0), ("U", 0.0, 2.0)], s=2),
    # priors beyond the reference's tests: positive scale parameters
    "gauss2_lognormal_sd": dict(model=("GaussianIID", dict(n_obs=50, sd=1.0, obs_mean=0.5)),
                                prior=[("N", 0.0, 1.0), ("L", -0.5, 0.5)], s=1),
    "gauss2_exponential_sd": dict(model=("GaussianIID", dict(n_obs=50, sd=1.0, obs_mean=0.5, obs_m2=1.2)),
                                  prior=[("U", -2.0, 2.0), ("E", 0.7, 0.0)], s=2),
    # prior families added in round 2: Gamma scale parameter; truncated-Normal mean with a Beta-distributed sd in (0, 1)
    "gauss2_gamma_sd": dict(model=("GaussianIID", dict(n_obs=40, sd=1.0, obs_mean=0.3, obs_m2=0.9)),
                            prior=[("N", 0.0, 1.0), ("G", 2.0, 0.5)], s=2),
    "gauss2_truncnormal_beta": dict(model=("GaussianIID", dict(n_obs=40, sd=1.0, obs_mean=0.4)),
                                    prior=[("T", 0.0, 1.0, -0.5, 1.5), ("B", 2.0, 2.5)], s=1),
    # joint Gaussian prior (MvNormal(mu, Sigma)): correlated mean components of the 2-D model; 3 dimensions over LV's rates
    "gauss2d_mvnormal": dict(model=("Gaussian2D", dict(n_obs=50, r=0.6, obs_mean=(1.2, -0.7), obs_varsum=2.1, obs_cov=0.55)),
                             prior=[("N", 0.5, 2.0), ("N", -0.5, 1.5)], cov=[[4.0, -1.8], [-1.8, 2.25]], s=3),
    # BASELINE config 3
    "gauss2d_cfg3": dict(model=("Gaussian2D", dict(n_obs=50, r=0.6, obs_mean=(1.2, -0.7), obs_varsum=2.1, obs_cov=0.55)),
                         prior=[("N", 0.0, 3.0), ("N", 0.0, 3.0)], s=3),
    # BASELINE config 4
    "gk_cfg4": dict(model=("GandK", dict(n_draws=128, c=0.8, ranks=(16, 48, 80, 112), obs=(1.9, 2.7, 3.6, 6.4))),
                    prior=[("U", 0.0, 10.0)] * 4, s=4),
    # the same with c = 0.9: above 0.83 the quantile function need not be increasing, so the device sorts the data
    # themselves (below it sorts the normals and maps the wanted ranks: device_models.hpp, gk_increasing)
    "gk_c09": dict(model=("GandK", dict(n_draws=100, c=0.9, ranks=(10, 40, 60, 95), obs=(1.5, 2.8, 3.4, 7.0))),
                   prior=[("U", 0.0, 10.0)] * 4, s=4),
    # ... and the two together with wanted ranks that are multiples of 16 -- the four-particles-per-wave network of round 4
    # (device_models.hpp: gk_simulate_rows4) sorting the DATA, and with fewer than 128 draws (the tail of the 128 is +inf;
    # rank 112 lies behind the 100 draws: an infinite order statistic, a distance of 1e30)
    "gk_c09_blocks": dict(model=("GandK", dict(n_draws=128, c=0.9, ranks=(16, 48, 80, 112), obs=(1.6, 2.7, 3.7, 6.9))),
                          prior=[("U", 0.0, 10.0)] * 4, s=4),
    "gk_short_blocks": dict(model=("GandK", dict(n_draws=100, c=0.8, ranks=(16, 32, 64, 96), obs=(1.9, 2.3, 3.2, 6.0))),
                            prior=[("U", 0.0, 10.0)] * 4, s=4),
    # BASELINE config 5
    "lv_cfg5": dict(model=("LotkaVolterra", dict(n_steps=256, dt=0.05, σ=0.1, x0=50.0, y0=50.0, obs=(18.0, 17.0, 14.0, 12.0))),
                    prior=[("U", 0.0, 2.0), ("U", 0.0, 0.1), ("U", 0.0, 2.0)], s=4),
}

PROPOSALS = {
    "rw": ("RandomWalk", 0.8, 0.0),
    "de": ("DifferentialEvolution", None, 1e-5),
    "stretch": ("StretchMove", 2.0, 0.0),
}


def oracle_model_params(O, spec):
    kind, kw = spec["model"]
    if kind == "GaussianIID":
        return O.MODEL_GAUSS_IID, [kw["n_obs"], kw["sd"], kw["obs_mean"], kw.get("obs_m2") or 0.0]
    if kind == "Gaussian2D":
        return O.MODEL_GAUSS2D, [kw["n_obs"], kw["r"], *kw["obs_mean"], kw["obs_varsum"], kw["obs_cov"]]
    if kind == "GandK":
        return O.MODEL_GK, [kw["n_draws"], kw["c"], *kw["ranks"], *kw["obs"]]
    if kind == "LotkaVolterra":
        return O.MODEL_LV, [kw["n_steps"], kw["dt"], kw["σ"], kw["x0"], kw["y0"], *kw["obs"]]
    raise KeyError(kind)


def oracle_config(O, name, n, algorithm="single_eps", seed=SEED, v=1.0, delta=0.1):
    spec = MODELS[name]
    mid, params = oracle_model_params(O, spec)
    kinds = {"N": O.PRIOR_NORMAL, "U": O.PRIOR_UNIFORM, "E": O.PRIOR_EXPONENTIAL, "L": O.PRIOR_LOGNORMAL, "G": O.PRIOR_GAMMA,
             "B": O.PRIOR_BETA, "T": O.PRIOR_TRUNCNORMAL}
    prior = [(kinds[p[0]],) + tuple(p[1:]) for p in spec["prior"]]
    alg = O.ALG_MULTI_EPS if algorithm == "multi_eps" else O.ALG_SINGLE_EPS
    chol = np.linalg.cholesky(np.asarray(spec["cov"], dtype=np.float64)) if "cov" in spec else None
    return O.make_config(n_particles=n, n_para=len(prior), n_stats=spec["s"], model_id=mid, model_params=params,
                         prior=prior, algorithm=alg, v=v, delta=delta, seed=seed, prior_chol=chol)


def oracle_proposal(O, prop, d):
    kind, p0, p1 = PROPOSALS[prop]
    k = {"RandomWalk": O.PROP_RANDOMWALK, "DifferentialEvolution": O.PROP_DIFFEVO, "StretchMove": O.PROP_STRETCH}[kind]
    return (k, p0, p1)


def oracle_run(O, name, n, n_simulation, algorithm="single_eps", prop="de", seed=SEED, **upd):
    """sabc() on the oracle: initialization + update_population! with the remaining budget."""
    cfg = oracle_config(O, name, n, algorithm, seed)
    run = O.OracleRun(cfg)
    run.initialize(n_simulation)
    d = len(MODELS[name]["prior"])
    run.update(O.make_update_args(n_simulation=n_simulation - n, proposal=oracle_proposal(O, prop, d), n_para=d,
                                  n_particles=n, **upd))
    return run


def hip_model_prior(S, name):
    spec = MODELS[name]
    kind, kw = spec["model"]
    model = getattr(S, kind)(**kw)
    make = {"N": lambda a, b: S.Normal(a, b), "U": lambda a, b: S.Uniform(a, b), "E": lambda a, b: S.Exponential(a),
            "L": lambda a, b: S.LogNormal(a, b), "G": lambda a, b: S.Gamma(a, b), "B": lambda a, b: S.Beta(a, b),
            "T": lambda a, b, lo, hi: S.truncated(S.Normal(a, b), lo, hi)}
    if "cov" in spec:
        return model, S.MvNormal([p[1] for p in spec["prior"]], spec["cov"])
    comps = [make[p[0]](*p[1:]) for p in spec["prior"]]
    prior = comps[0] if len(comps) == 1 else S.product_distribution(comps)
    return model, prior


def hip_proposal(S, prop, d):
    kind, p0, p1 = PROPOSALS[prop]
    if kind == "RandomWalk":
        return S.RandomWalk(β=p0, n_para=d)
    if kind == "DifferentialEvolution":
        return S.DifferentialEvolution(n_para=d, σ_gamma=p1)
    return S.StretchMove(a=p0)

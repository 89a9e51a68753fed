"""One `update_population!` call cut into several sabc_update calls (the wrapper does that to print the reference's
progress lines, SimulatedAnnealingABC.jl:359-364): `show_checkpoint` and `checkpoint_history` are INDEPENDENT moduli in the
reference (`ix % show_checkpoint` :359, `ix % checkpoint_history` :367, the final push :378-382), so the cuts fall anywhere
relative to the history's cadence.  sabc_update_args::history_phase / more_chunks_follow carry the loop's own numbering
across the calls: the histories are those of the uncut call -- on the host engine (CPU harness) and on the device, and
both equal the oracle's uncut call."""
import logging

import numpy as np
import pytest

from tests.cases import MODELS, SEED, hip_model_prior, hip_proposal, oracle_config, oracle_proposal


def chunked(h, S, n, cuts, total, cph, prop, d):
    done = 0
    for stop in list(cuts) + [total]:
        h.update(n_simulation=(stop - done) * n, proposal=hip_proposal(S, prop, d), checkpoint_history=cph, history_phase=done,
                 more_chunks_follow=stop < total, resample=n // 2)
        done = stop


def oracle_histories(O, name, n, total, cph, prop, d):
    run = O.OracleRun(oracle_config(O, name, n, algorithm="multi_eps"))
    run.initialize(n)
    run.update(O.make_update_args(n_simulation=total * n, n_para=d, n_particles=n, proposal=oracle_proposal(O, prop, d),
                                  checkpoint_history=cph, resample=n // 2))
    return run


@pytest.mark.parametrize("prop", ["de", "rw"])
@pytest.mark.parametrize("total,cph,cuts", [(13, 3, (7,)), (14, 3, (7,)), (12, 3, (7,)), (13, 5, (2, 4, 6, 8, 10, 12)), (9, 4, (9,)),
                                            (10, 1, (3, 6)), (7, 10, (2, 5)), (6, -3, (4,))])
def test_host_engine_chunks_keep_the_uncut_history(S, O, total, cph, cuts, prop):
    from tests import cpu_engine
    name, n = "gauss2_2stats", 300
    d = len(MODELS[name]["prior"])
    model, prior = hip_model_prior(S, name)
    H = cpu_engine.handle_class()
    hs = []
    for c in ((), tuple(x for x in cuts if x < total)):
        h = H(n_particles=n, model=model, prior=prior, seed=SEED, algorithm=S._lib.ALG_MULTI_EPS)
        h.initialize(n)
        chunked(h, S, n, c, total, cph, prop, d)
        hs.append(h)
    run = oracle_histories(O, name, n, total, cph, prop, d) if cph > 0 else None
    a, b = hs
    assert a.counters == b.counters
    for x, y in zip(a.history, b.history):
        np.testing.assert_array_equal(x, y)                  # the same rows, the same numbers
    for x, y in zip(a.get_population(), b.get_population()):
        np.testing.assert_array_equal(x, y)                  # ... and the same particles, bit for bit: a chunk that continues a call
                                                             # starts from the control block as the previous chunk left it
    expected = 1 + total // abs(cph) + (1 if total % abs(cph) else 0)
    assert len(a.history[0]) == expected
    if run is not None:
        assert len(run.history[0]) == expected
        np.testing.assert_allclose(b.history[0], run.history[0], rtol=1e-9)
    for h in hs:
        h.close()


def test_checkpoint_history_zero_is_the_references_divide_error(S, O):
    """`ix % 0` throws in the reference as soon as the loop runs (:367); a call that runs no update does not get there."""
    from tests import cpu_engine
    name, n = "gauss1_cfg2", 200
    model, prior = hip_model_prior(S, name)
    h = cpu_engine.handle_class()(n_particles=n, model=model, prior=prior, seed=SEED)
    h.initialize(n)
    with pytest.raises(S.SABCError, match="DivideError"):
        h.update(n_simulation=2 * n, proposal=hip_proposal(S, "rw", 1), checkpoint_history=0)
    h.update(n_simulation=n - 1, proposal=hip_proposal(S, "rw", 1), checkpoint_history=0)      # no update, no division
    assert h.counters["n_population_updates"] == 0
    h.update(n_simulation=2 * n, proposal=hip_proposal(S, "rw", 1), checkpoint_history=1)      # the failed call left the handle usable
    assert h.counters["n_population_updates"] == 2
    h.close()
    run = O.OracleRun(oracle_config(O, name, n))
    run.initialize(n)
    with pytest.raises(O.OracleError):
        run.update(O.make_update_args(n_simulation=2 * n, n_particles=n, proposal=oracle_proposal(O, "rw", 1), checkpoint_history=0))


@pytest.mark.gpu
@pytest.mark.parametrize("show_checkpoint,cph,total", [(7, 3, 13), (7, 3, 14), (5, 2, 12), (4, 9, 11)])
def test_show_checkpoint_is_independent_of_checkpoint_history_on_device(S, O, gpu, caplog, show_checkpoint, cph, total):
    """sabc(...; show_checkpoint = 7, checkpoint_history = 3): progress lines at 7 (and 14), history rows at 3, 6, 9, 12 and the
    final one -- the uncut call's, which is the oracle's."""
    name, n = "gauss1_cfg2", 500
    model, prior = hip_model_prior(S, name)
    kw = dict(n_particles=n, n_simulation=n * (total + 1), proposal=S.RandomWalk(n_para=1), seed=SEED, checkpoint_history=cph)
    a = S.sabc(model, prior, **kw)
    with caplog.at_level(logging.INFO, logger="SimulatedAnnealingABC"):
        b = S.sabc(model, prior, show_checkpoint=show_checkpoint, **kw)
    for k in range(show_checkpoint, total + 1, show_checkpoint):
        assert any(f"Update {k} of {total}" in r.message for r in caplog.records)
    expected = 1 + total // cph + (1 if total % cph else 0)
    assert len(a.state.ϵ_history) == len(b.state.ϵ_history) == expected
    np.testing.assert_array_equal(np.array(a.state.ϵ_history), np.array(b.state.ϵ_history))
    np.testing.assert_array_equal(np.array(a.state.u_history), np.array(b.state.u_history))
    np.testing.assert_array_equal(a.population, b.population)
    c = S.sabc(model, prior, show_progressbar=True, **kw)            # the bar's own cuts (a fiftieth of the run: every update here)
    np.testing.assert_array_equal(np.array(a.state.ϵ_history), np.array(c.state.ϵ_history))
    run = O.OracleRun(oracle_config(O, name, n))
    run.initialize(n * (total + 1))
    run.update(O.make_update_args(n_simulation=n * total, n_particles=n, proposal=oracle_proposal(O, "rw", 1), checkpoint_history=cph))
    assert len(run.history[0]) == expected
    np.testing.assert_allclose(np.array(b.state.ϵ_history).ravel(), run.history[0].ravel(), rtol=1e-9)
    with pytest.raises(ZeroDivisionError):
        S.sabc(model, prior, **dict(kw, checkpoint_history=0))

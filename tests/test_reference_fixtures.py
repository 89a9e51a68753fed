"""Values captured from the REAL reference by tools/reference_fixtures.jl (deterministic functions only: build_cdf,
update_epsilon_*, the weights of resample_population, update_proposal!).  Julia is not installed in the build
container, so the files are absent there and these tests report PARITY UNPINNED; once someone with Julia has run the
script and committed tests/golden/reference_*.json, the same tests pin the oracle (CPU) and the device operators (GPU)
to the reference's own numbers."""
import json
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    p = os.path.join(GOLD, name)
    if not os.path.exists(p):
        pytest.skip(f"parity unpinned: {name} has not been generated (tools/reference_fixtures.jl needs Julia + the reference)")

    def num(x):
        if isinstance(x, str):
            return {"Inf": np.inf, "-Inf": -np.inf, "NaN": np.nan}.get(x, x)
        if isinstance(x, list):
            return [num(e) for e in x]
        if isinstance(x, dict):
            return {k: num(v) for k, v in x.items()}
        return x
    return num(json.load(open(p)))


def test_fixture_script_is_committed():
    src = open(os.path.join(os.path.dirname(GOLD), "..", "tools", "reference_fixtures.jl")).read()
    for fn in ("SABC.build_cdf", "SABC.update_epsilon_single_eps", "SABC.update_epsilon_multi_eps", "SABC.resample_population",
               "SABC.update_proposal!"):
        assert fn in src
    assert "rand(" not in src.replace("no rand()", "")           # deterministic by construction


# ---- the oracle against the reference's numbers (CPU) ----
def test_oracle_cdf_against_reference(O):
    ref = load("reference_cdf.json")
    for case in ref["vector_cases"]:
        knots = O.build_cdf(np.array(case["x"], dtype=float))
        got = np.array([O.cdf_apply(knots, q) for q in case["q"]])
        np.testing.assert_allclose(got, case["cdf"], rtol=1e-12, atol=1e-15, err_msg=case["label"])
    m = ref["matrix"]
    rho = np.array(m["rho"], dtype=float)
    tables = [O.build_cdf(rho[:, j]) for j in range(rho.shape[1])]
    for row, want in zip(m["query_rows"], m["u"]):
        np.testing.assert_allclose([O.cdf_apply(tables[j], row[j]) for j in range(len(row))], want, rtol=1e-12, atol=1e-15)


def test_oracle_epsilon_against_reference(O):
    ref = load("reference_epsilon.json")
    for c in ref["single_eps"]:
        np.testing.assert_allclose(O.eps_single(c["ubar"], c["v"]), c["eps"][0], rtol=1e-9, atol=0)
    for c in ref["multi_eps"]:
        np.testing.assert_allclose(O.eps_multi(np.array(c["ubar"]), c["v"]), c["eps"], rtol=1e-9)


def test_oracle_resample_weights_against_reference(O):
    ref = load("reference_resample.json")
    for c in ref["cases"]:
        u = np.array(c["u"], dtype=float)
        w = np.exp(-(u * c["delta"] / u.mean(0)).sum(1))         # :127 restated; the oracle's scan must give the same ESS
        cum, bs, totals = O.weight_scan(w)
        np.testing.assert_allclose(totals[0] ** 2 / totals[1], c["ess"], rtol=1e-12)


def test_oracle_proposal_sigma_against_reference(O):
    ref = load("reference_proposal.json")
    for c in ref["cases"]:
        pop = np.array(c["population"], dtype=float)
        d = int(c["d"])
        cov = np.atleast_2d(np.cov(pop.T, ddof=1))
        want = c["beta"] * (cov + (1e-8 * np.eye(d) if d > 1 else 0.0))     # proposals.jl:47,59
        np.testing.assert_allclose(want, np.array(c["sigma"], dtype=float), rtol=1e-12)


# ---- the device operators against the reference's numbers (GPU, through the C-ABI) ----
@pytest.mark.gpu
def test_device_operators_against_reference(S, gpu):
    ref = load("reference_cdf.json")
    for case in ref["vector_cases"]:
        knots = S.op_build_cdf(np.array(case["x"], dtype=float))
        np.testing.assert_allclose(S.op_cdf_eval(knots, np.array(case["q"], dtype=float)), case["cdf"], rtol=1e-12, atol=1e-15)
    eps = load("reference_epsilon.json")
    for c in eps["single_eps"]:
        np.testing.assert_allclose(S.op_eps_single(c["ubar"], c["v"]), c["eps"][0], rtol=1e-9, atol=0)
    for c in eps["multi_eps"]:
        np.testing.assert_allclose(S.op_eps_multi(np.array(c["ubar"]), c["v"]), c["eps"], rtol=1e-9)

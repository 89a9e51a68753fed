#!/usr/bin/env python3
"""Generates tests/golden/*.json from the CPU oracle.

These vectors are SELF-DERIVED: the reference is Julia, Julia is not installed in the build
container and the reference's tests hold no values, so nothing here was
captured from the reference itself.  They pin the oracle against regressions and give the
HIP engine a fixed target that does not need the oracle at run time.

  python tests/golden/make_golden.py
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import oracle as O          # noqa: E402
from tests.cases import MODELS, oracle_run  # noqa: E402

CASES = [
    # (model, n, updates, algorithm, proposal)
    ("gauss1_cfg2", 256, 8, "single_eps", "rw"),
    ("gauss1_uniform", 255, 8, "multi_eps", "de"),
    ("gauss2_2stats", 256, 8, "multi_eps", "stretch"),
    ("gauss2d_cfg3", 256, 8, "single_eps", "rw"),
    ("gk_cfg4", 128, 6, "multi_eps", "de"),
    ("lv_cfg5", 128, 6, "single_eps", "rw"),
]


def main():
    for name, n, k, alg, prop in CASES:
        run = oracle_run(O, name, n, (k + 1) * n, algorithm=alg, prop=prop, seed=20241220, resample=n)
        e, u, r = run.history
        out = {
            "provenance": "self-derived from oracle/sabc_oracle.c; NOT captured from the Julia reference",
            "case": dict(model=name, n_particles=n, n_simulation=(k + 1) * n, algorithm=alg, proposal=prop,
                         seed=20241220, resample=n),
            "counters": run.counters,
            "eps": run.eps.tolist(),
            "theta": run.theta.tolist(), "u": run.u.tolist(), "rho": run.rho.tolist(),
            "eps_history": e.tolist(), "u_history": u.tolist(), "rho_history": r.tolist(),
            "cdf_len": [int(len(run.cdf_knots(j))) for j in range(MODELS[name]["s"])],
            "cdf_knots_head": [run.cdf_knots(j)[:8].tolist() for j in range(MODELS[name]["s"])],
        }
        path = os.path.join(HERE, f"{name}_{alg}_{prop}.json")
        with open(path, "w") as f:
            json.dump(out, f)
        print(path, run.counters)


if __name__ == "__main__":
    main()

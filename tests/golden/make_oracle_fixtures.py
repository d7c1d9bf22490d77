#!/usr/bin/env python3
"""Writes tests/golden/oracle_fixtures.json: small seeded problems with the values the CPU oracle (oracle/lmm_oracle.py)
gives for them.  The reference itself cannot be run in this pipeline (no Julia), so these are NOT reference outputs: they
freeze the oracle -- which is pinned by the reference's relational tests and notebook literals (tests/test_oracle.py) -- so
that an accidental change of the oracle and of the HIP path in the same direction cannot pass unnoticed.  Every case stores
its full inputs; rerun this script only when the oracle is changed on purpose.

    python tests/golden/make_oracle_fixtures.py
"""
import json, os, sys
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from oracle import lmm_oracle as O  # noqa: E402


def case(name, kinds, p, n, ns, s2, seed, orthogonal):
    rng = np.random.default_rng(seed)
    m = len(kinds)
    x = np.sort(rng.uniform(0.0, 6.0, n)); xs = np.sort(rng.uniform(0.0, 6.0, ns))
    gps = [{"kind": k, "variance": float(rng.uniform(0.5, 1.5)), "lengthscale": float(rng.uniform(0.6, 1.6)),
            "mean": float(rng.uniform(-0.3, 0.3))} for k in kinds]
    A = rng.uniform(0.1, 1.0, (p, m))
    y = rng.standard_normal(n * p); ys = rng.standard_normal(ns * p)
    out = {"name": name, "orthogonal": orthogonal, "p": p, "m": m, "n": n, "ns": ns, "sigma2": s2, "gps": gps,
           "x": x.tolist(), "xs": xs.tolist(), "y": y.tolist(), "ys": ys.tolist()}
    if orthogonal:
        U, _, _ = np.linalg.svd(A, full_matrices=False)
        S = np.linspace(2.0, 1.0, m)
        out["U"], out["S"] = U.tolist(), S.tolist()
        out["logpdf"] = O.oilmm_logpdf(gps, U, S, x, s2, y)
        post = O.oilmm_posterior(gps, U, S, x, s2, y)
        mu, var = O.oilmm_mean_var(post, U, S, xs, s2)
        out["post_logpdf"] = O.oilmm_logpdf(post, U, S, xs, s2, ys)
    else:
        out["H"] = A.tolist()
        out["logpdf"] = O.ilmm_logpdf(gps, A, x, s2, y)
        post = O.ilmm_posterior(gps, A, x, s2, y)
        mu, var = O.ilmm_mean_var(post, A, xs, s2)
        out["post_logpdf"] = O.ilmm_logpdf(post, A, xs, s2, ys)
    out["post_mean"], out["post_var"] = mu.tolist(), var.tolist()
    return out


cases = [
    case("oilmm_c0_like", ["se", "se", "se"], 5, 40, 7, 0.1, 11, True),
    case("oilmm_mixed_kernels", ["se", "matern32", "matern52"], 4, 33, 5, 0.05, 12, True),
    case("ilmm_dense", ["se", "matern32"], 3, 21, 4, 0.1, 13, False),
]
with open(os.path.join(HERE, "oracle_fixtures.json"), "w") as f:
    json.dump({"_source": "oracle/lmm_oracle.py via tests/golden/make_oracle_fixtures.py (NOT reference outputs)", "cases": cases}, f)
print("wrote", len(cases), "cases")

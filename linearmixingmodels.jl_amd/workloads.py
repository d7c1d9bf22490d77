"""Synthetic workloads of SURVEY.md section 8d (shared by bench.py and the measurement tools): pure NumPy input generation,
no arithmetic of the model.  (The CPU oracle keeps its own copy so that it stays self-contained test infrastructure; the two
are compared in tests/test_oracle.py.)"""
from __future__ import annotations

from typing import Dict

import numpy as np


def synthetic_problem(m: int, p: int, n: int, kind: str, orthogonal: bool, s2: float = 0.1, seed: int = 0) -> Dict:
    """x_i = i*20/575 (the reference notebook's density); unit-variance, unit-lengthscale kernels of one kind, zero mean;
    H from svd(uniform(p, m)) with S = linspace(2, 1, m) (OILMM) or dense uniform(0, 1) (ILMM); y standard normal.
    Seeds: H -> seed + 2, y -> seed + 3."""
    x = np.arange(n, dtype=np.float64) * (20.0 / 575.0)
    gps = [{"kind": kind, "variance": 1.0, "lengthscale": 1.0, "mean": 0.0} for _ in range(m)]
    A = np.random.default_rng(seed + 2).uniform(0.0, 1.0, (p, m))
    out = {"x": x, "gps": gps, "s2": s2, "m": m, "p": p, "n": n}
    if orthogonal:
        U, _, _ = np.linalg.svd(A, full_matrices=False)
        out["U"], out["S"] = np.ascontiguousarray(U), np.linspace(2.0, 1.0, m)
        out["H"] = out["U"] * np.sqrt(out["S"])[None, :]
    else:
        out["H"] = A
    out["y"] = np.random.default_rng(seed + 3).standard_normal(n * p)
    return out

"""Shared by tests/golden/make_golden.py (which runs the REFERENCE to produce the
fixtures) and the tests that replay them: scenario forcing from stored hourly
knots, expanded with plain numpy float64 arithmetic (deterministic, IEEE)."""
from __future__ import annotations

import os

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
KNOT_FIELDS = ("tair", "tdew", "vz", "rhz", "prec", "sw", "lw")


def expand_knots(knots: dict, simlen: int, spk: int, start=(2024, 1, 10, 0, 0, 0)) -> dict:
    """knots[field][n, nk] -> step-resolution arrays [n, simlen] (reference layout).
    v = k0 + (r * (k1 - k0)) / spk ; PrecPhase from the later knot between knots
    (the rule of examples/example1/src/JsonSource.cpp:115-172)."""
    import oracle_helpers as oh

    t = np.arange(simlen)
    k = t // spk
    r = (t - k * spk).astype(np.float64)
    f = {}
    for name in KNOT_FIELDS:
        K = knots[name]
        k0, k1 = K[:, k], K[:, np.minimum(k + 1, K.shape[1] - 1)]
        v = k0 + (r * (k1 - k0)) / float(spk)
        f[name] = np.ascontiguousarray(np.where(r == 0, k0, v))
    ph = knots["phase"]
    f["precphase"] = np.ascontiguousarray(
        np.where(r == 0, ph[:, k], ph[:, np.minimum(k + 1, ph.shape[1] - 1)]).astype(np.int32))
    n = f["tair"].shape[0]
    f["sw_dir"] = np.ascontiguousarray(0.6 * f["sw"])
    f["lw_net"] = np.full((n, simlen), -40.0)
    f["tsurfobs"] = np.full((n, simlen), -9999.9)
    f["tsurfobs"][:, 0] = knots["tsurf0"]
    f["depth"] = np.full((n, simlen), -9999.9)
    f.update(oh.time_axis(simlen, 3600.0 / spk, start))
    return f


def load(name: str):
    return np.load(os.path.join(GOLDEN_DIR, name), allow_pickle=False)

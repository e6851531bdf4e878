"""Config 5 study: fp32 flavour against the fp64 reference over 7 days (SimLen 20161)."""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, os.path.join(ROOT, 'tools'))
import numpy as np
import oracle_helpers as oh
from roadsurf_amd import abi
from f32_experiment import run_f32

n = int(os.environ.get("N", 2048)); L = 168 * 120 + 1
for seed in (20240110, 777):
    f = oh.synth_forcing(n, L, seed=seed)
    s = abi.default_settings(L); p = abi.default_parameters(); l = abi.default_local(); l.InitLenI = 1
    ora, _, _ = oh.run_oracle("ref" if oh.have_ref() else "port", f, s, p, l)
    res = run_f32(n, L, seed)
    d = np.abs(res["tsurf"] - ora["tsurf"])
    over = d > 0.05
    # longest excursion per point, in steps
    longest = 0
    runs = []
    for q in np.where(over.any(1))[0]:
        x = np.flatnonzero(np.diff(np.concatenate([[0], over[q].astype(np.int8), [0]])))
        runs += list(x[1::2] - x[0::2])
    runs = np.array(runs) if runs else np.zeros(1)
    print(f"seed {seed}: tsurf rms {np.sqrt((d**2).mean()):.2e} p99 {np.percentile(d,99):.2e} p99.9 {np.percentile(d,99.9):.2e} "
          f"max {d.max():.3f} frac>0.05K {over.mean():.2e} points ever>0.05K {over.any(1).mean():.3f} "
          f"excursions: n={len(runs)} median {np.median(runs)*0.5:.0f} min, longest {runs.max()*0.5/60:.1f} h")
    for day in range(7):
        sl = slice(day * 2880, (day + 1) * 2880)
        print(f"   day {day+1}: rms {np.sqrt((d[:, sl]**2).mean()):.2e} frac>0.05 {over[:, sl].mean():.2e} max {d[:, sl].max():.3f}")
    for k in ("snow", "water", "ice", "deposit", "ice2"):
        e = np.abs(res[k] - ora[k])
        print(f"   {k:8s} rms {np.sqrt((e**2).mean()):.2e} p99.9 {np.percentile(e,99.9):.2e} max {e.max():.3f} frac>0.05mm {(e>0.05).mean():.2e}")

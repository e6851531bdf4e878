"""Round 6 soak of the fp32 flavour: the distribution gate of tests/test_hip_f32.py on seeds, point counts and launch
lengths the suite does not use, knot-reading launch against window launch (bits) and plan order against natural
order (bits) for each.  usage: python tools/f32_soak.py [first_seed] [count]"""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import oracle_helpers as oh
from roadsurf_amd import abi
from f32_experiment import run_f32

first = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
count = int(sys.argv[2]) if len(sys.argv) > 2 else 20
bad = 0
for seed in range(first, first + count):
    rs = np.random.RandomState(seed)
    n = int(rs.choice([129, 500, 1023, 2048, 3001]))
    L = int(rs.choice([721, 1441, 2881]))
    chunk = int(rs.choice([60, 97, 120, 240, 361]))
    f = oh.synth_forcing(n, L, seed=seed)
    s = abi.default_settings(L); p = abi.default_parameters(); l = abi.default_local(); l.InitLenI = 1
    ora, _, _ = oh.run_oracle("ref" if oh.have_ref() else "port", f, s, p, l)
    win = run_f32(n, L, seed, chunk=chunk)
    kn = run_f32(n, L, seed, chunk=chunk, fused=True)
    srt = run_f32(n, L, seed, chunk=chunk, fused=True, cluster=True)
    d = np.abs(kn["tsurf"] - ora["tsurf"])
    # (a sample of 129 points: one point is 0.8 % of it, so the percentile that holds the bulk to microkelvins is taken
    # where a few points on the other side of a melt-out branch cannot reach it, and the share of such points - 1.5 %
    # in tests/test_hip_f32.py - gets three standard deviations of a sample of n: seed 2003, 3 of 129 points, 0.09 K)
    over = d > 0.05
    pts_over = int(over.any(1).sum())
    q = 99.9 if n >= 1000 else 99.0 if n >= 250 else 97.0
    ok = (np.sqrt((d ** 2).mean()) < 1e-3 and np.percentile(d, q) < 2e-3 and d.max() < 0.5 and over.mean() < 1e-4
          and pts_over <= 0.015 * n + 3.0 * np.sqrt(0.015 * n))
    for k in ("snow", "water", "ice", "deposit", "ice2"):
        e = np.abs(kn[k] - ora[k])
        ok = ok and np.sqrt((e ** 2).mean()) < 5e-4 and e.max() < 0.1
    same = all(np.array_equal(win[k], kn[k]) and np.array_equal(win[k], srt[k]) for k in win)
    print("seed %d n %d L %d chunk %d: rms %.2e p99.9 %.2e max %.3f frac>0.05K %.1e points ever>0.05K %d  gate %s  bits(window = knots = plan order) %s"
          % (seed, n, L, chunk, np.sqrt((d ** 2).mean()), np.percentile(d, 99.9), d.max(), (d > 0.05).mean(), pts_over,
             "ok" if ok else "FAILED", "ok" if same else "FAILED"), flush=True)
    bad += (not ok) + (not same)
print("failures:", bad)
sys.exit(1 if bad else 0)

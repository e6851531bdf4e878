import sys
sys.path.insert(0, "tests"); sys.path.insert(0, "tools"); sys.path.insert(0, ".")
import numpy as np
import oracle_helpers as oh
from roadsurf_amd import abi
from f32_experiment import run_f32
seed, n, L, chunk = 3001, 3001, 1441, 97
f = oh.synth_forcing(n, L, seed=seed)
s = abi.default_settings(L); p = abi.default_parameters(); l = abi.default_local(); l.InitLenI = 1
ora, _, _ = oh.run_oracle("ref" if oh.have_ref() else "port", f, s, p, l)
kn = run_f32(n, L, seed, chunk=chunk, fused=True)
for k in ("tsurf", "snow", "water", "ice", "deposit", "ice2"):
    e = np.abs(kn[k] - ora[k])
    q = np.unravel_index(e.argmax(), e.shape)
    print(k, "rms %.2e p99.9 %.2e max %.3f at point %d index %d; points with > 0.05: %d" % (np.sqrt((e**2).mean()), np.percentile(e, 99.9), e.max(), q[0], q[1], int((e > 0.05).any(1).sum())))
pt = np.unravel_index(np.abs(kn["water"] - ora["water"]).argmax(), ora["water"].shape)[0]
t = int((np.abs(kn["water"][pt] - ora["water"][pt]) > 0.02).argmax())
for k in ("tsurf", "snow", "water", "ice", "deposit", "ice2"):
    print(k, "ora", np.round(ora[k][pt, t-2:t+4], 4), "f32", np.round(kn[k][pt, t-2:t+4], 4))

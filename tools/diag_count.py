import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests'))
import numpy as np
import oracle_helpers as oh
from roadsurf_amd import abi, device
n = int(sys.argv[1]); off = int(sys.argv[2]); seed = int(sys.argv[3]) if len(sys.argv) > 3 else 20240110
L = 5761
f = oh.synth_forcing(n, L, seed=seed, point_offset=off)
s = abi.default_settings(L); p = abi.default_parameters(); l = abi.default_local(); l.InitLenI = 1
ora, _, _ = oh.run_oracle("ref" if oh.have_ref() else "port", f, s, p, l)
res, _ = device.run_points(f, s, p, l)
d = np.maximum.reduce([np.abs(res[k] - ora[k]) for k in oh.F64_OUT])
dts = np.abs(res['tsurf'] - ora['tsurf'])
print('points', n, 'points with any |diff|>1e-9:', int((d.max(1) > 1e-9).sum()), ' tsurf>1e-6:', int((dts.max(1) > 1e-6).sum()),
      'max tsurf diff', dts.max(), 'bit-identical points:', int((d.max(1) == 0).sum()))

import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests'))
import numpy as np
import oracle_helpers as oh
from roadsurf_amd import abi, device
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
off = int(sys.argv[2]) if len(sys.argv) > 2 else 0
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 20240110
L = 5761
f = oh.synth_forcing(n, L, seed=seed, point_offset=off)
s = abi.default_settings(L); p = abi.default_parameters(); l = abi.default_local(); l.InitLenI = 1
ora, _, _ = oh.run_oracle("ref" if oh.have_ref() else "port", f, s, p, l)
res, _ = device.run_points(f, s, p, l)
for k in oh.F64_OUT:
    d = np.abs(res[k] - ora[k])
    pt, t = np.unravel_index(d.argmax(), d.shape)
    print(k, 'max', d.max(), 'at point', pt, 't', t, 'n>1e-9:', int((d > 1e-9).sum()), 'n>1e-12:', int((d>1e-12).sum()))

dall = np.maximum.reduce([np.abs(res[k] - ora[k]) for k in oh.F64_OUT])
bad = np.where(dall.max(1) > 1e-9)[0]
print('bad points', bad)
for pt in bad[:2]:
    t0 = int(np.argmax(dall[pt] > 1e-13))
    print('point', pt, 'first t with any diff>1e-13:', t0)
    for t in range(max(0, t0 - 3), t0 + 4):
        print(t, ' '.join('%s %.17g|%.17g' % (k[:3], res[k][pt, t], ora[k][pt, t]) for k in ('tsurf', 'snow', 'water', 'ice', 'deposit')), 'tair %.6f vz %.4f rh %.3f prec %.4f ph %d sw %.2f lw %.2f hr %d' % (f['tair'][pt, t], f['vz'][pt,t], f['rhz'][pt,t], f['prec'][pt,t], f['precphase'][pt,t], f['sw'][pt,t], f['lw'][pt,t], f['hour'][t]))

"""Diagnostic for tests/test_hip_random_configs.py: where does a drawn configuration differ?"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import oracle_helpers as oh
from roadsurf_amd import abi, lib
from test_hip_random_configs import _draw

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 6
L = lib.load()

def run(f, s, p, ls):
    n, SL = f["tair"].shape
    g = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in f.items()}
    out = {k: np.full((n, SL), np.nan) for k in oh.F64_OUT}
    ips = (abi.InputPointers * n)(); ops = (abi.OutputPointers * n)(); keep = []
    for pt in range(n):
        ip, op, kp = oh.point_pointers(g, pt, out)
        ips[pt], ops[pt] = ip, op
        keep.append(kp)
    larr = (abi.LocalParameters * n)(*ls)
    st = C.c_int32(99)
    L.runsimulation_batch(n, ops, ips, C.byref(s), C.byref(p), larr, C.byref(st))
    assert st.value == 0, lib.last_error()
    return out

f, s, p, ls = _draw(seed)
for label, mod in (("as drawn", lambda s: None), ("no coupling", lambda s: setattr(s, "use_coupling", 0)),
                   ("no depth", lambda s: setattr(s, "tsurfOutputDepth", -9999.9)),
                   ("NL15", lambda s: setattr(s, "NLayers", 15))):
    s2 = abi.InputSettings.from_buffer_copy(s)
    mod(s2)
    ora, _, _ = oh.run_oracle("port", f, s2, p, ls)
    out = run(f, s2, p, ls)
    d = out["tsurf"] != ora["tsurf"]
    pts = np.nonzero(d.any(1))[0]
    print("RESULT", label, "differing values", int(d.sum()), "points", pts[:8])
    for q in pts[:2]:
        i = np.nonzero(d[q])[0][0]
        print("RESULT   point", q, "first diff at index", i, "gpu", out["tsurf"][q, i - 2:i + 3], "ora", ora["tsurf"][q, i - 2:i + 3],
              "cpl", ls[q].couplingIndexI, ls[q].couplingTsurf, "initlen", ls[q].InitLenI, "sky", ls[q].sky_view)

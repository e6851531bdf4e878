#!/usr/bin/env bash
# round 6, last pass: the storages for a point PAIR in one basic block (x2_road_condition / x2_melting, rs_kernels_f32.hip)
# against the one-point source run twice (-DRS_X2_ROAD_SCALAR).  Bits first (same outputs, LEAN and FULL, 7 days), then the
# rate at config 5's shape.  Build the A/B library first (CPU box):
#   make -C roadsurf_amd OBJ=build_x2scalar LIB=lib/libroadsurf_hip_x2scalar.so EXTRA=-DRS_X2_ROAD_SCALAR -j8
set -e
OUT=gpurun_out/r6_x2road
mkdir -p $OUT
for a in "" x2scalar; do
  export ROADSURF_HIP_LIB=$PWD/roadsurf_amd/lib/libroadsurf_hip${a:+_$a}.so
  python3 - "$OUT/out_${a:-pair}.npz" <<'PY'
import sys
sys.path.insert(0, "tests"); sys.path.insert(0, "tools"); sys.path.insert(0, ".")
import numpy as np
import oracle_helpers as oh
from roadsurf_amd import abi, device
from f32_experiment import run_f32
res = run_f32(2048, 20161, 20240110)
out = {"lean_" + k: v.astype(np.float32) for k, v in res.items()}
res = run_f32(1500, 5761, 777, cluster=True, fused=True)
out.update({"knots_" + k: v.astype(np.float32) for k, v in res.items()})
n, L = 2048, 2881
f = oh.synth_forcing(n, L, seed=17)
s = abi.default_settings(L); s.use_relaxation = 1
p = abi.default_parameters()
rs = np.random.RandomState(5)
ls = []
for i in range(n):
    li = abi.default_local(); li.InitLenI = int(rs.choice([1, 240, 600, 721, 1000]))
    li.tair_relax = float(f["tair"][i, min(li.InitLenI, L - 1)] + rs.uniform(-2, 2)); li.VZ_relax = 3.0; li.RH_relax = 80.0
    ls.append(li)
f["tsurfobs"][:, :] = f["tair"] + 1.0
r2, _ = device.run_points(f, s, p, ls, precision=32)
out.update({"full_" + k: v.astype(np.float32) for k, v in r2.items()})
np.savez(sys.argv[1], **out)
PY
done
python3 - <<'PY'
import numpy as np
a = np.load("gpurun_out/r6_x2road/out_pair.npz"); b = np.load("gpurun_out/r6_x2road/out_x2scalar.npz")
for k in a.files:
    same = np.array_equal(a[k], b[k])
    print("%-14s %s  values %d  differing %d" % (k, "identical" if same else "DIFFERENT", a[k].size, int((a[k] != b[k]).sum())))
PY
rm -f $OUT/out_*.npz
B="--f32 --points 1250000 --hours 168 --no-natural-leg --no-extra-legs --no-cpu-baseline --steps 3 --warmup 1"
for rep in 1 2; do for a in "" x2scalar; do
  export ROADSURF_HIP_LIB=$PWD/roadsurf_amd/lib/libroadsurf_hip${a:+_$a}.so
  python3 bench.py $B > $OUT/bench_${a:-pair}_$rep.json 2>/dev/null
  python3 -c "import json;d=json.load(open('$OUT/bench_${a:-pair}_$rep.json'));print('${a:-pair} rep $rep LEAN  %.4e  ms/pass %.1f'%(d['value'],d['ms_per_step']))"
  python3 bench.py $B --full > $OUT/bench_full_${a:-pair}_$rep.json 2>/dev/null
  python3 -c "import json;d=json.load(open('$OUT/bench_full_${a:-pair}_$rep.json'));print('${a:-pair} rep $rep FULL  %.4e  ms/pass %.1f'%(d['value'],d['ms_per_step']))"
done; done

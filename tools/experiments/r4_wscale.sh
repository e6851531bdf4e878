#!/bin/bash
# round 4: how the step kernel scales with the waves resident per SIMD, one plan, one launch at a time
mkdir -p gpurun_out/r4_wscale
for N in 65536 131072 196608 262144; do
  for V in 1 3; do
    python3 bench.py --total-points $N --plans-per-gpu 1 --cluster 0 --variant $V --chunk 240 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r4_wscale/nat_${N}_v$V.json 2> gpurun_out/r4_wscale/nat_${N}_v$V.err
    python3 bench.py --total-points $N --plans-per-gpu 1 --cluster 1 --variant $V --chunk 240 --steps 5 --warmup 2 --no-cpu-baseline --no-natural-leg > gpurun_out/r4_wscale/srt_${N}_v$V.json 2> gpurun_out/r4_wscale/srt_${N}_v$V.err
  done
done
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4_wscale/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); r=d['roofline']
        n=d['config']['points_per_gpu']; waves=(n+63)//64
        # quads per wave-step at 2.3 GHz
        q = r['avg_launch_ms']*1e-3*2.3e9/4/ (r['units_per_launch']/n)
        print(f.split('/')[-1], '%.4g'%d['value'], 'ms/pass %.1f'%d['ms_per_step'], 'avg_launch %.3f'%r['avg_launch_ms'], 'kernel-only %.4g'%r['step_kernel_only_value'], 'waves/SIMD %.2f'%(waves/1024), 'quads per step %.0f'%q)
    except Exception as e:
        print(f, 'ERR', e)
PY

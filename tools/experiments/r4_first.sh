#!/bin/bash
# round 4, first GPU call: issue-model microbenchmark + today's baseline numbers
set -x
mkdir -p gpurun_out/r4_first
tools/bin/issue_model > gpurun_out/r4_first/issue_model.txt 2>&1
python3 bench.py --steps 5 --warmup 2 --no-natural-leg --no-cpu-baseline > gpurun_out/r4_first/bench_1m.json 2> gpurun_out/r4_first/bench_1m.err
python3 bench.py --total-points 125000 --steps 10 --warmup 3 --no-natural-leg --no-cpu-baseline > gpurun_out/r4_first/bench_125k.json 2> gpurun_out/r4_first/bench_125k.err
python3 bench.py --total-points 125000 --steps 10 --warmup 3 --no-natural-leg --no-cpu-baseline --variant 1 > gpurun_out/r4_first/bench_125k_v1.json 2> gpurun_out/r4_first/bench_125k_v1.err
python3 bench.py --total-points 250000 --steps 10 --warmup 3 --no-natural-leg --no-cpu-baseline > gpurun_out/r4_first/bench_250k.json 2> gpurun_out/r4_first/bench_250k.err
cat gpurun_out/r4_first/issue_model.txt
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4_first/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); r=d['roofline']
        print(f, '%.4g'%d['value'], 'ms/pass %.1f'%d['ms_per_step'], 'avg_launch %.3f'%r['avg_launch_ms'], 'conc %.2f'%r['concurrent_launches'], 'variant', d['config']['kernel_variant'])
    except Exception as e:
        print(f, 'ERR', e)
PY

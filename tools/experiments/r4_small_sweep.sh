#!/bin/bash
# small shards: launch length / plan count / flavour with the round-4 two-wavefront kernel
run() { python3 bench.py --steps 8 --warmup 2 --no-natural-leg --no-cpu-baseline --no-extra-legs "$@" | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']; print(c['points_per_gpu'], 'plans', c['plans_per_gpu'], 'chunk', c['chunk_steps'], 'variant', c['kernel_variant'], '%.4g'%d['value'], 'launch %.3f ms'%d['roofline']['avg_launch_ms'], 'conc %.2f'%d['roofline']['concurrent_launches'])"; }
for ch in 240 300 360 480; do run --total-points 125000 --chunk $ch; done
for k in 3 4; do run --total-points 125000 --plans-per-gpu $k --variant 3; done
for v in 1 3; do run --total-points 250000 --variant $v; done
run --total-points 250000 --variant 3 --chunk 120
for v in 1 3; do run --total-points 500000 --variant $v; done

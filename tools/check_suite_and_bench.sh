#!/bin/bash
# GPU test suite + default bench (the driver's round-end commands), logs under gpurun_out/
TAG=${1:-suite}
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r4_${TAG}.log 2>&1
rc=$?
tail -5 gpurun_out/r4_${TAG}.log
[ $rc -ne 0 ] && exit $rc
python bench.py --steps 5 --warmup 2 > gpurun_out/r4_${TAG}_bench.json 2> gpurun_out/r4_${TAG}_bench.err || { tail -20 gpurun_out/r4_${TAG}_bench.err; exit 1; }
python3 - <<PY
import json
d=json.loads(open('gpurun_out/r4_${TAG}_bench.json').read().strip().splitlines()[-1])
print('value %.4g'%d['value'], 'natural %.4g'%d.get('natural_order_value',0), 'frac %.3f'%d['roofline']['frac'], 'valu', d['roofline']['valu_issue_frac'])
for k in ('full_feature_value','driver_path_relax_value','driver_path_coupling_value','driver_path_sky_value'):
    print(k, '%.4g'%d.get(k,0))
print('extra seconds', d.get('extra_legs',{}).get('seconds'), 'cpu', d.get('cpu_baseline',{}).get('value'))
PY

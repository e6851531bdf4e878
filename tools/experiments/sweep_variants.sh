for v in 0 21 31 41 32 42; do
  python bench.py --steps 2 --warmup 1 --variant $v --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('variant',d['config']['kernel_variant'],'value %.3e'%d['value'],'kernel-only %.3e'%r['step_kernel_only_value'],'avg launch ms %.2f'%r['avg_launch_ms'])"
done

B="--full --no-extra-legs --no-natural-leg --no-cpu-baseline --steps 5 --warmup 2"
for N in 250000 125000; do for V in 4 3 1; do
v=$(python bench.py --total-points $N --variant $V $B 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4g launch %.3f ms in flight %.2f K %d chunk %d'%(d['value'], d['roofline']['avg_launch_ms'], d['roofline']['concurrent_launches'], d['config']['plans_per_gpu'], d['config']['chunk_steps']))")
echo "points $N variant $V: $v"; done; done

B="--full --no-extra-legs --no-natural-leg --no-cpu-baseline --steps 5 --warmup 2"
for KC in "3 120" "3 90" "3 60" "2 90" "4 120"; do set -- $KC
v=$(python bench.py --plans-per-gpu $1 --chunk $2 $B 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4g launch %.3f ms in flight %.2f'%(d['value'], d['roofline']['avg_launch_ms'], d['roofline']['concurrent_launches']))")
echo "full 1M plans $1 chunk $2: $v"; done

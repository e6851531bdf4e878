mkdir -p gpurun_out
OUT=gpurun_out/r4_key10_sweep.txt
: > $OUT
B="--no-extra-legs --no-natural-leg --no-cpu-baseline --steps 10 --warmup 2"
run() { v=$(python bench.py --total-points $1 --plans-per-gpu $2 --chunk $3 $B 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4g launch %.3f ms in flight %.2f'%(d['value'], d['roofline']['avg_launch_ms'], d['roofline']['concurrent_launches']))"); echo "points $1 plans $2 chunk $3: $v" | tee -a $OUT; }
for KC in "3 90" "3 60" "2 60" "2 90" "3 120" "4 60"; do run 1000000 $KC; done
for KC in "4 120" "4 90" "3 90" "4 60"; do run 250000 $KC; done
for KC in "2 240" "2 180" "2 120" "3 180"; do run 125000 $KC; done
for KC in "3 90" "3 60" "2 60"; do run 500000 $KC; done

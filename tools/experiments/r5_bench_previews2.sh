B="--no-cpu-baseline --no-natural-leg --no-extra-legs --steps 6 --warmup 2"
for P in 3 2 3 2; do
  ROADSURF_HIP_PREVIEWS=$P timeout -k 10 200 python3 bench.py $B | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lean previews $P', d['value'])"
  ROADSURF_HIP_PREVIEWS=$P timeout -k 10 200 python3 bench.py $B --total-points 500000 | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('500k previews $P', d['value'])"
done

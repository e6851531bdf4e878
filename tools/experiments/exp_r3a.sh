#!/usr/bin/env bash
# round 3, first GPU call: baselines at 1 M / 250 k / 125 k points from HEAD, the counters that say
# where a wave's cycles go when fewer than two waves share a SIMD, and the existing variants
set -e
OUT=gpurun_out/r3a
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 -L > $OUT/counters_avail.txt 2>&1 || true
run() { # name, bench args
  local name=$1; shift
  python bench.py --no-cpu-baseline --no-natural-leg "$@" > $OUT/$name.json 2> $OUT/$name.err || { tail -5 $OUT/$name.err; exit 1; }
  python - <<PY
import json
d=json.load(open("$OUT/$name.json")); r=d["roofline"]
print("$name value %.4e ms/pass %.1f kernel-only %.4e avg launch %.2f ms conc %.2f"%(d["value"],d["ms_per_step"],r["step_kernel_only_value"],r["avg_launch_ms"],r["concurrent_launches"]), flush=True)
PY
}
run n1m --steps 5
run n250k --total-points 250000 --steps 10
run n125k --total-points 125000 --steps 10
# variants at 125 k: waves-per-SIMD bound of the register flavour (10*W + 1), plans, launch length
run n125k_v21 --total-points 125000 --steps 10 --variant 21
run n125k_v31 --total-points 125000 --steps 10 --variant 31
run n125k_k1 --total-points 125000 --steps 10 --plans-per-gpu 1
run n125k_k4 --total-points 125000 --steps 10 --plans-per-gpu 4
run n125k_k2_c480 --total-points 125000 --steps 10 --chunk 480
run n125k_nat --total-points 125000 --steps 10 --cluster 0
# counters at 125 k
G1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM"
G2="SQ_WAVES SQ_INSTS_SMEM SQ_INST_CYCLES_SMEM SQ_INST_CYCLES_SALU SQ_INSTS_BRANCH SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC SQ_IFETCH"
G3="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
G4="SQ_WAVES SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_IFETCH_LEVEL SQ_INSTS_FLAT SQ_INSTS_VSKIPPED SQ_INST_CYCLES_VMEM"
i=1
for G in "$G1" "$G2" "$G3" "$G4"; do
  rocprofv3 --kernel-trace --pmc $G --output-format csv -d $OUT/pmc_g$i -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-natural-leg --total-points 125000 > $OUT/pmc_bench$i.json 2> $OUT/pmc_err$i.txt || { tail -20 $OUT/pmc_err$i.txt; echo "group $i failed"; }
  i=$((i+1))
done
python3 tools/summarize_pmc.py $OUT > $OUT/pmc_summary.txt || true
grep -E "step_kernel" $OUT/pmc_summary.txt || true
# drop the raw csv (large) but keep the summary
rm -rf $OUT/pmc_g*

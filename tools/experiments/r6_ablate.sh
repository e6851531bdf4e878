#!/usr/bin/env bash
# round 6: what the parts of step_kernel_f32duo cost - VALU instructions per launch and the whole-job rate of builds
# with one part cut out (-DRS_ABL_*: wrong results, counts only).  Build the libraries first (CPU box):
#   for a in NOROAD NOBL NOLAYERS NOFROZEN; do make -C roadsurf_amd OBJ=build_abl_$a LIB=lib/libroadsurf_hip_abl_$a.so EXTRA=-DRS_ABL_$a -j8; done
OUT=gpurun_out/r6_ablate
mkdir -p $OUT
export TMPDIR=/tmp
B="--f32 --points 1250000 --hours 168 --no-natural-leg --no-extra-legs --no-cpu-baseline --plans-per-gpu 2 --chunk 240"
for a in "" abl_NOROAD abl_NOBL abl_NOLAYERS abl_NOFROZEN; do
  L=roadsurf_amd/lib/libroadsurf_hip${a:+_$a}.so
  [ -f $L ] || continue
  export ROADSURF_HIP_LIB=$PWD/$L
  python3 bench.py $B > $OUT/bench_${a:-full}.json 2>/dev/null
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $OUT/pmc_${a:-full} -- python3 bench.py --steps 1 --warmup 0 $B > /dev/null 2> $OUT/err_${a:-full}.txt
  python3 - <<PY
import json,csv,glob
d=json.load(open("$OUT/bench_${a:-full}.json"))
f=glob.glob("$OUT/pmc_${a:-full}/*/*counter_collection.csv")[0]
tot={}
n=0
for r in csv.DictReader(open(f)):
    if "f32duo" not in r["Kernel_Name"]: continue
    tot[r["Counter_Name"]]=tot.get(r["Counter_Name"],0)+float(r["Counter_Value"])
wg=4883*20161*2/2  # workgroup-steps per pass (2 plans x 4883 workgroups ... per plan 20161 steps)
wg=4883*2*20161
print("%-14s %.3e pt-steps/s   per 128 point-steps: VALU %.0f  VALU-active quad-cycles %.0f  SALU %.0f   wait/wave-cycles %.2f"%("${a:-full}", d["value"], tot["SQ_INSTS_VALU"]/wg, tot["SQ_ACTIVE_INST_VALU"]/wg, tot["SQ_INSTS_SALU"]/wg, tot["SQ_WAIT_INST_ANY"]/tot["SQ_WAVE_CYCLES"]))
PY
  rm -rf $OUT/pmc_${a:-full}
done

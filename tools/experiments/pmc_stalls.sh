#!/usr/bin/env bash
# where the step kernel's waves spend their cycles: rocprofv3 --pmc, two counter groups, one bench pass each
set -e
OUT=gpurun_out/pmcs_${1:-x}
shift || true
mkdir -p $OUT
export TMPDIR=/tmp
G1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM"
G2="SQ_WAVES SQ_INSTS_SMEM SQ_INST_CYCLES_SMEM SQ_INST_CYCLES_SALU SQ_INSTS_BRANCH SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC SQ_IFETCH"
G3="SQ_WAVES SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT64"
i=1
for G in "$G1" "$G2" "$G3"; do
rocprofv3 --kernel-trace --pmc $G --output-format csv -d $OUT/pmc_g$i -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-natural-leg "$@" > $OUT/bench$i.json 2> $OUT/err$i.txt || { tail -20 $OUT/err$i.txt; exit 1; }
i=$((i+1))
done
python3 tools/summarize_pmc.py $OUT | grep -E "step_kernel"

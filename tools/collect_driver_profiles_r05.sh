#!/usr/bin/env bash
# after tools/profile_driver_r05.sh, tools/experiments/r5_pmc_driver.sh, r5_final_driver_modes.sh and
# tools/wave_stats.py on the GPU box (outputs under gpurun_out/): the driver-path evidence into profiles/
# usage: collect_driver_profiles_r05.sh <tag of the gpurun_out/r5_<tag>_* files of that call>
set -e
T=$1
SHA=$(python3 -c "from roadsurf_amd import provenance; print(provenance.csrc_sha16())")
cp gpurun_out/profiles_r05/r05_driver_path_* profiles/
for M in relax skyview coupling; do
  { echo "# tools/experiments/r5_pmc_driver.sh $M: SQ counters of rs_driver_run's kernels, 1 M points x 48 h, distinct series, two calls (rocprofv3 --pmc, two passes); kernel sources $SHA"; cat gpurun_out/r5_pmc_drv_$M/pmc_summary.txt; } > profiles/r05_driver_path_${M}_pmc_summary.txt
done
{ echo "# tools/experiments/r5_final_driver_modes.sh at HEAD ($SHA): rs_driver_run, 1 M points x 48 h, distinct series for every point, host arrays in and out"; cat gpurun_out/r5_${T}_driver_modes.txt; } > profiles/r05_driver_modes.txt
cp gpurun_out/r5_${T}_ws.txt profiles/r05_wave_stats.txt

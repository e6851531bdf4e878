#!/usr/bin/env bash
# after tools/profile_r06.sh on the GPU box: traffic summaries (stamped with the hash of the kernel
# sources) and the rocprofv3 summaries from gpurun_out/prof_r06* into profiles/
set -e
python tools/make_traffic_json.py gpurun_out/prof_r06 profiles/r06_traffic.json
TRAFFIC_POINTS=1250000 TRAFFIC_SIMLEN=20161 TRAFFIC_BYTES_PER_UNIT=52 python tools/make_traffic_json.py gpurun_out/prof_r06_f32 profiles/r06_f32_traffic.json "step_kernel_f32duo<1, false, false, false>" 360 2
G=gpurun_out
cp $G/prof_r06/bench.json profiles/r06_bench.json
cp $G/prof_r06_f32/bench.json profiles/r06_f32_bench.json
cp $G/prof_r06/bench_under_rocprof.json profiles/r06_bench_under_rocprof.json
cp $G/prof_r06/kernel_stats.csv profiles/r06_kernel_stats.csv
cp $G/prof_r06/pmc_summary.txt profiles/r06_pmc_summary.txt
cp $G/prof_r06/timeline.txt profiles/r06_timeline.txt
cp $G/prof_r06_f32/kernel_stats.csv profiles/r06_f32_kernel_stats.csv
cp $G/prof_r06_f32/pmc_summary.txt profiles/r06_f32_pmc_summary.txt
cp $G/prof_r06_f32/timeline.txt profiles/r06_f32_timeline.txt
cp $G/prof_r06_full/bench.json profiles/r06_full_bench.json
cp $G/prof_r06_full/kernel_stats.csv profiles/r06_full_kernel_stats.csv
cp $G/prof_r06_full/pmc_summary.txt profiles/r06_full_pmc_summary.txt
cp $G/prof_r06_full/timeline.txt profiles/r06_full_timeline.txt
cp $G/prof_r06_small_v1/bench_250k.json profiles/r06_small_shard_bench_250k.json
cp $G/prof_r06_small_v1/bench.json profiles/r06_small_shard_one_point_per_lane_bench_125k.json
cp $G/prof_r06_small_v1/pmc_summary.txt profiles/r06_small_shard_one_point_per_lane_pmc_summary.txt
cp $G/prof_r06_small_v3/bench.json profiles/r06_small_shard_two_wavefronts_bench_125k.json
cp $G/prof_r06_small_v3/pmc_summary.txt profiles/r06_small_shard_two_wavefronts_pmc_summary.txt

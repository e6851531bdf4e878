#!/usr/bin/env bash
# after tools/profile_r04.sh on the GPU box: traffic summaries (stamped with the hash of the kernel
# sources) and the rocprofv3 summaries from gpurun_out/prof_r04* into profiles/
set -e
python tools/make_traffic_json.py gpurun_out/prof_r04 profiles/r04_traffic.json
TRAFFIC_POINTS=1250000 TRAFFIC_SIMLEN=20161 TRAFFIC_BYTES_PER_UNIT=52 python tools/make_traffic_json.py gpurun_out/prof_r04_f32 profiles/r04_f32_traffic.json "step_kernel_f32_lds" 120 4
G=gpurun_out
cp $G/prof_r04/bench.json profiles/r04_bench.json
cp $G/prof_r04/bench_under_rocprof.json profiles/r04_bench_under_rocprof.json
cp $G/prof_r04/kernel_stats.csv profiles/r04_kernel_stats.csv
cp $G/prof_r04/pmc_summary.txt profiles/r04_pmc_summary.txt
cp $G/prof_r04/timeline.txt profiles/r04_timeline.txt
cp $G/prof_r04_f32/kernel_stats.csv profiles/r04_f32_kernel_stats.csv
cp $G/prof_r04_f32/pmc_summary.txt profiles/r04_f32_pmc_summary.txt
cp $G/prof_r04_f32/timeline.txt profiles/r04_f32_timeline.txt
cp $G/prof_r04_full/bench.json profiles/r04_full_bench.json
cp $G/prof_r04_full/kernel_stats.csv profiles/r04_full_kernel_stats.csv
cp $G/prof_r04_full/pmc_summary.txt profiles/r04_full_pmc_summary.txt
cp $G/prof_r04_full/timeline.txt profiles/r04_full_timeline.txt
cp $G/prof_r04_small_v1/bench_250k.json profiles/r04_small_shard_bench_250k.json
cp $G/prof_r04_small_v1/bench.json profiles/r04_small_shard_one_point_per_lane_bench_125k.json
cp $G/prof_r04_small_v1/pmc_summary.txt profiles/r04_small_shard_one_point_per_lane_pmc_summary.txt
cp $G/prof_r04_small_v3/bench.json profiles/r04_small_shard_two_wavefronts_bench_125k.json
cp $G/prof_r04_small_v3/pmc_summary.txt profiles/r04_small_shard_two_wavefronts_pmc_summary.txt

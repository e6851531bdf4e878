#!/bin/bash
# round-4 evidence that is not a rocprof summary: issue-model microbenchmark, the literal drop-in
# table, more launches in flight for the headline; all under gpurun_out/profiles_r04/
P=gpurun_out/profiles_r04; mkdir -p $P
{ echo "# tools/issue_model.hip on an MI355X box of the pool (round 4): what W resident wavefronts with the road model's instruction mix get out of a SIMD"; tools/bin/issue_model; } > $P/r04_issue_model.txt 2>&1
# (per-wavefront times: profiles/r04_wave_times.txt was collected at commit 68a2f12 with a library built by
#  `make -C roadsurf_amd OBJ=build_wt LIB=lib/libroadsurf_hip_wt.so EXTRA=-DRS_WAVE_TIMING` - the one-point-per-lane
#  flavour of that commit; rebuild that library before running tools/wave_times.py again)
python3 tools/bench_dropin.py 768 1,16,64,256 48 > $P/r04_dropin.txt 2>&1
{ echo "# bench.py, 1 M points x 48 h: more plans (launches in flight) with GPU_MAX_HW_QUEUES=8 - does the launch tail limit the headline?";
  run() { python3 bench.py --steps 6 --warmup 2 --no-natural-leg --no-cpu-baseline --no-extra-legs "$@" | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']; print(c['plans_per_gpu'], 'plans, launches of', c['chunk_steps'], 'indices: %.4g point-timesteps/s,'%d['value'], 'avg launch %.3f ms,'%d['roofline']['avg_launch_ms'], '%.2f launches in flight'%d['roofline']['concurrent_launches'])"; }
  run; export GPU_MAX_HW_QUEUES=8; run --plans-per-gpu 6; run --plans-per-gpu 8; run --plans-per-gpu 8 --chunk 60; } > $P/r04_more_launches_in_flight.txt 2>&1
tail -3 $P/r04_dropin.txt; cat $P/r04_more_launches_in_flight.txt

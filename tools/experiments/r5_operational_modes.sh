#!/usr/bin/env bash
# the operational shape (401 stations x 8 881, coupling windows that end hours apart) under the driver's options
echo "default: $(timeout -k 10 100 python3 tools/bench_operational.py files 4 2>&1 | tail -1)"
echo "ROADSURF_HIP_DRIVER_WINDOWS=1: $(ROADSURF_HIP_DRIVER_WINDOWS=1 timeout -k 10 100 python3 tools/bench_operational.py files 4 2>&1 | tail -1)"
echo "ROADSURF_HIP_CPL_WHOLE=1: $(ROADSURF_HIP_CPL_WHOLE=1 timeout -k 10 100 python3 tools/bench_operational.py files 4 2>&1 | tail -1)"
echo "ROADSURF_HIP_CHUNK_STEPS=480: $(ROADSURF_HIP_CHUNK_STEPS=480 timeout -k 10 100 python3 tools/bench_operational.py files 4 2>&1 | tail -1)"
echo "ROADSURF_HIP_CLUSTER=0: $(ROADSURF_HIP_CLUSTER=0 timeout -k 10 100 python3 tools/bench_operational.py files 4 2>&1 | tail -1)"
echo "ROADSURF_HIP_CLUSTER=0 ROADSURF_HIP_CHUNK_STEPS=960: $(ROADSURF_HIP_CLUSTER=0 ROADSURF_HIP_CHUNK_STEPS=960 timeout -k 10 100 python3 tools/bench_operational.py files 4 2>&1 | tail -1)"
echo "sky, default: $(timeout -k 10 100 python3 tools/bench_operational.py sky 4 2>&1 | tail -1)"
echo "sky, ROADSURF_HIP_DRIVER_WINDOWS=1: $(ROADSURF_HIP_DRIVER_WINDOWS=1 timeout -k 10 100 python3 tools/bench_operational.py sky 4 2>&1 | tail -1)"

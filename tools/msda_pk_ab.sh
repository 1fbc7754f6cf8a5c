#!/bin/bash
# same-box A/B: msda.hip built without (old) / with (new) packed-fp32 instructions; output hashes must agree between the builds
for which in old new old new; do
  cp gpurun_ab/lib_$which.so gomatching_amd/libgomatching_hip.so
  echo "== $which"
  timeout 300 python3 tools/msda_window_bench.py 2>&1 | tail -6
done
for which in new; do
  cp gpurun_ab/lib_$which.so gomatching_amd/libgomatching_hip.so
  timeout 900 python3 -m pytest tests/test_determinism_gpu.py tests/test_ops_gpu.py -q -m gpu -x -k "determinism or msda or repro" 2>&1 | tail -5
done
bash tools/ab_builds.sh
cp gpurun_ab/lib_old.so gomatching_amd/libgomatching_hip.so

#!/bin/bash
# same-box A/B of two builds of the library: gpurun_ab/lib_old.so against gpurun_ab/lib_new.so, alternating
for rnd in 1 2 3; do
  for which in old new; do
    cp gpurun_ab/lib_$which.so gomatching_amd/libgomatching_hip.so
    timeout 300 python3 bench.py --no-alt-backends --no-cpu-baseline --no-config-legs --steps 15 > gpurun_out/bench_ab.json 2> gpurun_out/bench_ab.err
    python3 - "$which" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/bench_ab.json").read().strip().splitlines()[-1])
print("%-6s %8.2f frames/s  (hbm-resident %8.2f)  %7.3f ms/step" % (sys.argv[1], d["value"], d.get("value_hbm_resident"), d["ms_per_step"]))
PY
  done
done

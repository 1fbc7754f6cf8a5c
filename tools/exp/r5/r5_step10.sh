#!/bin/bash
timeout 900 python3 -m pytest tests/test_bneck_gpu.py -q -x 2>&1 | tail -4
timeout 400 python3 tools/bneck_bench.py 2>&1 | tail -4
timeout 1500 python3 -m pytest tests/test_model_gpu.py tests/test_fullsize_gpu.py -q -x 2>&1 | tail -5
for rnd in 1 2 3; do
  for v in "GOM_BNECK2=0" "GOM_BNECK2=1"; do
    env $v GOM_BENCH_ALL_SHAPES=gpurun_out/shapes_$v.txt timeout 300 python3 bench.py --no-alt-backends --no-cpu-baseline --no-config-legs --steps 15 > gpurun_out/bench_ab.json 2> gpurun_out/bench_ab.err
    python3 - "$v" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/bench_ab.json").read().strip().splitlines()[-1])
f = d.get("roofline_bneck", {})
t = d.get("roofline_tile_gemm", {})
print("%-14s %8.2f frames/s  %7.3f ms/step  bneck avg %.0f us x %s | tile gemm share %.3f x %s" % (sys.argv[1], d["value"], d["ms_per_step"], f.get("avg_launch_us", 0), f.get("launches_per_step"), t.get("share_of_step_time", 0), t.get("launches_per_step")))
PY
  done
done

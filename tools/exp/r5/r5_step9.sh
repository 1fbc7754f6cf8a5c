#!/bin/bash
timeout 900 python3 -m pytest tests/test_proj_ln_gpu.py -q -x 2>&1 | tail -3
timeout 1500 python3 -m pytest tests/test_model_gpu.py tests/test_fullsize_gpu.py -q -x 2>&1 | tail -5
for rnd in 1 2 3; do
  for v in "GOM_PROJ_LN_V2=0" "GOM_PROJ_LN_V2=1"; do
    env $v timeout 300 python3 bench.py --no-alt-backends --no-cpu-baseline --no-config-legs --steps 15 > gpurun_out/bench_ab.json 2> gpurun_out/bench_ab.err
    python3 - "$v" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/bench_ab.json").read().strip().splitlines()[-1])
f = d.get("roofline_proj_ln", {})
print("%-18s %8.2f frames/s  %7.3f ms/step  proj_ln frac %.3f avg %.0f us x %s" % (sys.argv[1], d["value"], d["ms_per_step"], f.get("frac", 0), f.get("avg_launch_us", 0), f.get("launches_per_step")))
PY
  done
done

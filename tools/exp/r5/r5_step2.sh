#!/bin/bash
timeout 900 python3 -m pytest tests/test_dec_tail_gpu.py -q -x 2>&1 | tail -5
timeout 300 python3 tools/dec_tail_bench.py 2>&1 | tail -5
timeout 1500 python3 -m pytest tests/test_model_gpu.py tests/test_fullsize_gpu.py -q -x -k "not fixture" 2>&1 | tail -5
for rnd in 1 2; do
  for v in "GOM_DEC_TAIL=0" "GOM_DEC_TAIL_PROJ=0" "GOM_DEC_TAIL_PROJ=1"; do
    env $v timeout 300 python3 bench.py --no-alt-backends --no-cpu-baseline --no-config-legs --steps 15 > gpurun_out/bench_ab.json 2> gpurun_out/bench_ab.err
    python3 - "$v" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/bench_ab.json").read().strip().splitlines()[-1])
q = d.get("roofline_decoder_qside", {})
print("%-20s %8.2f frames/s  %7.3f ms/step  qside frac %.3f  us/step %.0f  %s" % (sys.argv[1], d["value"], d["ms_per_step"], q.get("frac", 0), q.get("us_per_step", 0), {k: round(v["us"]) for k, v in q.get("by_kernel_us_per_step", {}).items()}))
PY
  done
done

#!/bin/bash
timeout 900 python3 -m pytest tests/test_dec_attn_gpu.py -q -x 2>&1 | tail -3
for rnd in 1 2; do
    timeout 300 python3 bench.py --no-alt-backends --no-cpu-baseline --no-config-legs --steps 15 > gpurun_out/bench_ab.json 2> gpurun_out/bench_ab.err
    python3 - <<'PY'
import json, sys
d = json.loads(open("gpurun_out/bench_ab.json").read().strip().splitlines()[-1])
q = d.get("roofline_decoder_qside", {})
print("%8.2f frames/s  %7.3f ms/step  qside frac %.3f  us/step %.0f  %s" % (d["value"], d["ms_per_step"], q.get("frac", 0), q.get("us_per_step", 0), {k: round(v["us"]) for k, v in q.get("by_kernel_us_per_step", {}).items()}))
PY
done

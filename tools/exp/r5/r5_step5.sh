#!/bin/bash
timeout 900 python3 -m pytest tests/test_dist_gpu.py tests/test_dec_attn_gpu.py -q -x 2>&1 | tail -4
for rnd in 1 2; do
for cfg in "1 0 x" "8 32 --replicate-short-term" "8 32 x" "8 0 x"; do
  set -- $cfg
  extra=""; [ "$3" != "x" ] && extra="$3"
  timeout 300 python3 bench.py --emulate-world $1 --tracker-cus $2 $extra --steps 8 --warmup 2 --no-alt-backends --no-cpu-baseline --no-config-legs > gpurun_out/emu.json 2> gpurun_out/emu.err
  python3 - "$cfg" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/emu.json").read().strip().splitlines()[-1])
st = d["stage_ms_per_step"]
print("%-34s %7.2f ms/step  %7.2f frames/s  tracker alone %s  stages %s" % (sys.argv[1], d["ms_per_step"], d["value"], d["config"]["tracker_alone_ms_per_step"], {k: round(v, 2) for k, v in st.items() if k.startswith("finish") or k in ("short_match", "long_match")}))
PY
done
done

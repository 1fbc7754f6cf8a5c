#!/bin/bash
timeout 300 python3 tools/ffn_reg_bench.py 2>&1 | tail -12
for cfg in "8 16 x" "8 24 x" "8 32 x" "8 48 x"; do
  set -- $cfg
  timeout 300 python3 bench.py --emulate-world $1 --tracker-cus $2 --steps 8 --warmup 2 --no-alt-backends --no-cpu-baseline --no-config-legs > gpurun_out/emu.json 2> gpurun_out/emu.err
  python3 - "$cfg" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/emu.json").read().strip().splitlines()[-1])
st = d["stage_ms_per_step"]
print("%-34s %7.2f ms/step  %7.2f frames/s  tracker alone %s  stages %s" % (sys.argv[1], d["ms_per_step"], d["value"], d["config"]["tracker_alone_ms_per_step"], {k: round(v, 2) for k, v in st.items() if k.startswith("finish") or k in ("short_match", "long_match")}))
PY
done

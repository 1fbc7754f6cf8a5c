#!/bin/bash
# the one-launch long-term match (csrc/match_fused.hip): parity tests, then the 8-GPU-load step with it and with the chain, same box
out=gpurun_out/r5_step22; mkdir -p $out
timeout 900 python3 -m pytest tests/test_match_fused_gpu.py -x -q 2>&1 | tail -15
timeout 900 python3 -m pytest tests/test_model_gpu.py tests/test_dist_gpu.py -x -q -k "match or track or hoist or dist or rank" 2>&1 | tail -4
run() { # name, extra args, cus
  timeout 200 python3 bench.py --emulate-world 8 --tracker-cus $3 $2 --steps 6 --warmup 2 --no-alt-backends --no-cpu-baseline --no-config-legs > $out/$1.json 2> $out/$1.err
  python3 - $out/$1.json $1 <<'P'
import json,sys
d=json.load(open(sys.argv[1])); s=d.get("stage_ms_per_step",{})
print("%-22s %.2f ms/step  long %.2f  wait_det %.2f  track %.2f  alone %s" % (sys.argv[2], d["ms_per_step"], s.get("long_match",0), s.get("finish_wait_detector",0), s.get("finish_track",0), d.get("tracker_alone_ms")))
P
}
for rep in 1 2; do
run chain_cu32 "" 32
run fused_cu32 --match-fused 32
run chain_cu0 "" 0
run fused_cu0 --match-fused 0
done
run fused_cu16 --match-fused 16
run fused_cu8 --match-fused 8
timeout 200 python3 bench.py --match-fused --steps 6 --warmup 2 --no-alt-backends --no-cpu-baseline --no-config-legs 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print('N=1 fused', d['ms_per_step'])"
timeout 200 python3 bench.py --steps 6 --warmup 2 --no-alt-backends --no-cpu-baseline --no-config-legs 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print('N=1 chain', d['ms_per_step'])"

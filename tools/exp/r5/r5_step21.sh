#!/bin/bash
# experiment: what the long-term chain's ten inner launches cost the 8-GPU-load step (GOM_EXP_SKIP_CHAIN: gather + logits + score only)
out=gpurun_out/r5_step21; mkdir -p $out
run() { # name, env, cus
  env $2 timeout 200 python3 bench.py --emulate-world 8 --tracker-cus $3 --steps 6 --warmup 2 --no-alt-backends --no-cpu-baseline --no-config-legs > $out/$1.json 2> $out/$1.err
  python3 - $out/$1.json $1 <<'P'
import json,sys
d=json.load(open(sys.argv[1])); s=d.get("stage_ms_per_step",{})
print("%-22s %.2f ms/step  long %.2f  wait_det %.2f  track %.2f" % (sys.argv[2], d["ms_per_step"], s.get("long_match",0), s.get("finish_wait_detector",0), s.get("finish_track",0)))
P
}
for rep in 1 2; do
run full_cu32 GOM_EXP_SKIP_CHAIN=0 32
run skip_cu32 GOM_EXP_SKIP_CHAIN=1 32
run full_cu0 GOM_EXP_SKIP_CHAIN=0 0
run skip_cu0 GOM_EXP_SKIP_CHAIN=1 0
done
timeout 200 python3 bench.py --steps 6 --warmup 2 --no-alt-backends --no-cpu-baseline --no-config-legs 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print('N=1', d['ms_per_step'])"

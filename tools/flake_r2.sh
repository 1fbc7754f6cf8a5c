#!/bin/bash
# usage: flake_r2.sh <runs> -- fresh-process loop per config "mode:switch[,switch..]:tap" with the double check on; prints every
# diagnostic line (cut) and a count per config
runs=${1:-40}
for cfg in ${CFGS:-"bf16x6::0" "bf16x6::1" "bf16x6:prewarm_kernels:0" "bf16x6:prewarm_alloc:0"}; do
  IFS=: read mode sw tap <<< "$cfg"
  hits=0
  for i in $(seq 1 $runs); do
    out=$(GOM_TRACKER_DOUBLE_CHECK=1 GOM_TRACKER_TAP=$tap FLAKE_SWITCH=$sw timeout 120 python tools/swin_flake.py $mode 2>&1 | grep -E "MISMATCH|DIFF|TAP|finite|Error|error" | cut -c1-${CUT:-300})
    if [ -n "$out" ]; then hits=$((hits+1)); echo "[$cfg run $i]"; echo "$out"; fi
  done
  echo "== $cfg runs=$runs processes_with_a_line=$hits"
done

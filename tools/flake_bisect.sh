#!/bin/bash
# usage: flake_bisect.sh <runs>  -- fresh-process repro loop of tools/swin_flake.py under each switch
runs=${1:-60}
for cfg in ${CFGS:-"bf16x6:" "bf16x6:emptycache" "bf16x6:syncafter" "bf16x6:zerows"}; do
  mode=${cfg%%:*}; sw=${cfg##*:}
  same=0; diff=0
  for i in $(seq 1 $runs); do
    out=$(FLAKE_SWITCH=$sw timeout 120 python tools/swin_flake.py $mode 2>&1 | grep -E "SAME|DIFF|saved|Error" | cut -c1-300)
    case "$out" in SAME*) same=$((same+1));; DIFF*) diff=$((diff+1)); echo "$cfg $out";; *) echo "$cfg $out";; esac
  done
  echo "== $cfg same=$same diff=$diff"
done

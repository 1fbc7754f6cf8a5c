#!/bin/bash
# fresh-process loop with GOM_TRACKER_DOUBLE_CHECK=1: which device result of the tracker is not reproducible?
runs=${1:-120}
for i in $(seq 1 $runs); do
  GOM_TRACKER_DOUBLE_CHECK=1 timeout 120 python tools/swin_flake.py bf16x6 2>&1 | grep -E "MISMATCH|DIFF" | cut -c1-200
done | sort | uniq -c | sort -rn | head -20
echo "== done $runs"

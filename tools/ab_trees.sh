#!/bin/bash
# same-box A/B of two source trees (a git worktree of an older commit under gpurun_ab/<name>, built there, against the repo root):
# one bench run per tree, alternating.   usage (on the GPU box): [BENCH_ARGS='--emulate-world 8'] bash tools/ab_trees.sh gpurun_ab/r6mid [rounds]
old=$1
for rnd in $(seq 1 ${2:-3}); do
  for which in "$old" "."; do
    (cd $which && timeout 300 python3 bench.py --no-alt-backends --no-cpu-baseline --no-config-legs --steps 15 $BENCH_ARGS 2> /dev/null | python3 -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
q = d.get('roofline_decoder_qside', {})
print('%-18s %8.2f frames/s %7.3f ms/step   decoder Q-side %.3f of 833, %.0f us per step' % (sys.argv[1], d['value'], d['ms_per_step'], q.get('frac', 0), q.get('us_per_step', 0)))" $which)
  done
done

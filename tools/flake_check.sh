#!/bin/bash
# baseline first; only a box that reproduces the flake is worth the longer run of the candidate fix
CFGS='bf16x6:' bash tools/flake_bisect.sh 70 > gpurun_out/check_base.log 2>&1
grep "==" gpurun_out/check_base.log
if grep -q "diff=0" gpurun_out/check_base.log; then echo "box does not reproduce"; exit 0; fi
CFGS='bf16x6:h2dkernel' bash tools/flake_bisect.sh 160 > gpurun_out/check_fix.log 2>&1
grep "==" gpurun_out/check_fix.log

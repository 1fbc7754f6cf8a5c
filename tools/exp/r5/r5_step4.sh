#!/bin/bash
for v in v5 v3; do
cp gpurun_ab/lib_$v.so gomatching_amd/libgomatching_hip.so
echo "== $v"
timeout 900 python3 -m pytest tests/test_dec_attn_gpu.py -q -x -k intra 2>&1 | tail -3
timeout 300 python3 tools/dec_attn_bench.py 2>&1 | tail -7
done

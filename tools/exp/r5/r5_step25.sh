#!/bin/bash
# separable position table of the encoder's projection (gom_gemm_k256_rs_f32): parity, then same-box A/B of the step
out=gpurun_out/r5_step25; mkdir -p $out
timeout 900 python3 -m pytest tests/test_gemm_k256_gpu.py -x -q 2>&1 | tail -4
for rep in 1 2 3; do
for sep in 1 0; do
  GOM_POS_SEPARABLE=$sep timeout 300 python3 bench.py --steps 10 --warmup 3 --no-alt-backends --no-cpu-baseline --no-config-legs > $out/sep$sep.$rep.json 2> $out/err.log
  python3 - $out/sep$sep.$rep.json $sep <<'P'
import json,sys
d=json.load(open(sys.argv[1]))
k=d.get("roofline_k256_long",{})
print("POS_SEPARABLE=%s  %.2f frames/s  %.3f ms/step   k256_long: %s" % (sys.argv[2], d["value"], d["ms_per_step"], {a:k.get(a) for a in ("launches","avg_us","achieved","frac")}))
P
done
done
timeout 1500 python3 -m pytest tests/test_model_gpu.py tests/test_fullsize_gpu.py tests/test_deepsolo_gpu.py -x -q 2>&1 | tail -4

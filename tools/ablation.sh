#!/bin/bash
# Same-box A/B of the f16x3 path's build-time switches (box-to-box variance on this pool is +-3 %, more than most of them are
# worth): one bench run per configuration, first and last = everything on.   usage (on the GPU box): bash tools/ablation.sh
for cfg in "" "GOM_DEC_ATTN=0" "GOM_DEC_ATTN_INTRA=0" "GOM_DEC_ATTN_INTER=0" "GOM_BNECK_FUSED=0" "GOM_PROJ_LN=0" "GOM_FUSED_FFN=0" "GOM_K256_GEMM=0" "GOM_STEM_POOL=0" "GOM_REF_UPDATE=0" "GOM_PROPOSAL_DOT=0" "GOM_CONV3_PATCH=0" "GOM_MSDA_WINDOW=0" "GOM_DEC_TAIL=0" "GOM_DEC_TAIL_PROJ=0" "GOM_DEC_ATTN_RAW=0" "GOM_BNECK2=0" "GOM_DEC_TAIL2=0" "GOM_DEC_ATTN2=0" "GOM_DEC_TAIL2_WAVES8=0" "GOM_MSDA_WINDOW_POLICY=0" "GOM_MSDA_WINDOW_L1=0" ""; do
  env $cfg timeout 300 python3 bench.py --no-alt-backends --no-cpu-baseline --no-config-legs --steps 15 > gpurun_out/bench_ab.json 2> gpurun_out/bench_ab.err
  python3 - "$cfg" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/bench_ab.json").read().strip().splitlines()[-1])
print("%-22s %8.2f frames/s  (hbm-resident %8.2f)  %7.3f ms/step" % (sys.argv[1] or "all on", d["value"], d.get("value_hbm_resident"), d["ms_per_step"]))
PY
done

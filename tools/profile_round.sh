#!/bin/bash
# Round profiles (run on the GPU box through gpurun): bench line, rocprofv3 kernel stats, the two PMC passes, per-shape GEMM
# trace.  Everything lands under gpurun_out/final/; copy what is judged into profiles/.
#   usage: bash tools/profile_round.sh [tag]
tag=${1:-r06}
out=gpurun_out/final_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o stats -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-alt-backends --no-config-legs > $out/${tag}_bench_under_rocprof.json 2> $out/stats.err
echo "stats rc $?"
cp $out/stats/stats_kernel_stats.csv $out/${tag}_kernel_stats.csv 2>/dev/null
python3 tools/trace_gap_sites.py $out/stats/stats_kernel_trace.csv passagg best > $out/${tag}_detector_pass_census.txt 2>> $out/stats.err
rm -f $out/stats/stats_kernel_trace.csv
GOM_BENCH_WRITE_GRIDS=$out/gemm_api_grids.json timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/pmc_fetch -o f -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-alt-backends --no-config-legs > /dev/null 2> $out/pmc_fetch.err
echo "fetch rc $?"
timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/pmc_write -o w -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-alt-backends --no-config-legs > /dev/null 2> $out/pmc_write.err
echo "write rc $?"
python3 tools/pmc_traffic.py $out/pmc_fetch/f_counter_collection.csv $out/pmc_write/w_counter_collection.csv $tag $out/gemm_api_grids.json | head -14
cp profiles/pmc_traffic.json $out/pmc_traffic.json
# the round's bench line AFTER the PMC passes: `roofline.traffic` then carries the counters of this very build
timeout 600 python3 bench.py --steps 20 --warmup 3 > $out/${tag}_bench_final.json 2> $out/bench_final.err
echo "bench rc $?"
rm -rf $out/pmc_fetch $out/pmc_write
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $out/shapes -o s -- python3 tools/gemm_shapes.py > $out/${tag}_gemm_shapes_hip_events.log 2> $out/shapes.err
echo "shapes rc $?"
python3 tools/gemm_shapes_csv.py $out/shapes/s_kernel_trace.csv gpurun_out/gemm_shapes_order.json $out/${tag}_gemm_shapes.csv
rm -rf $out/shapes
# multi-GPU tracker load emulated on one GPU (same box): N = 1, 4 and 8 GPUs' frames per tracker, without / with the CU lane
for cfg in "1 0" "4 0" "8 0" "8 32"; do
  set -- $cfg
  timeout 200 python3 bench.py --emulate-world $1 --tracker-cus $2 --steps 6 --warmup 2 --no-alt-backends --no-cpu-baseline --no-config-legs > $out/emu_w$1_cu$2.json 2> $out/emu.err
done
# ... and with every rank scoring every frame pair (rounds 1-4) instead of its own share + the second all-gather
timeout 200 python3 bench.py --emulate-world 8 --tracker-cus 32 --replicate-short-term --steps 6 --warmup 2 --no-alt-backends --no-cpu-baseline --no-config-legs > $out/emu_w8_cu32_replicated.json 2> $out/emu.err
bash tools/ablation.sh > $out/${tag}_ablation_same_box.log 2>&1

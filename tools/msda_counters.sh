#!/bin/bash
# TA / TCP counter passes over encoder-sized fused MSDA launches: is the kernel at the roof of the vector-L1 /
# texture-address path?  Each --pmc group is its own run (kernel trace only).  Output: gpurun_out/msda_counters/summary.txt
out=gpurun_out/msda_counters
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python3 tools/msda_counters.py > $out/plain.txt 2>&1
i=0
for grp in "TA_TA_BUSY_sum TA_BUSY_avr GRBM_GUI_ACTIVE" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_READ_sum" \
           "TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_GATE_EN1_sum" \
           "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_BUFFER_TOTAL_CYCLES_sum TA_BUFFER_READ_WAVEFRONTS_sum" \
           "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum" \
           "SQ_WAVES SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_BUSY_CYCLES"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $out/p$i -o c -- python3 tools/msda_counters.py > $out/p$i.log 2>&1
  echo "pass $i ($grp) rc $?"
done
python3 tools/msda_counters.py summarise $out/p*/c_counter_collection.csv > $out/summary.txt 2>&1
cat $out/plain.txt $out/summary.txt
rm -rf $out/p?/c_kernel_trace.csv

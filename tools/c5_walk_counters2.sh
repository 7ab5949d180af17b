#!/bin/bash
# Second set of counter passes over one C5-shaped batch (tools/c5_walk_counters.py): the LATENCIES the walk sees -- L1->L2 read latency, reads outstanding behind the
# L2, how many go to DRAM -- at the rule's residency (9 walk blocks per CU) and at 6 blocks per CU (WA_WALK_LDS_PAD=8192): does the memory system answer
# more waves with proportionally more latency?
#   gpurun --timeout 1500 -- 'bash tools/c5_walk_counters2.sh'        -> gpurun_out/r06w/c5_walk_latency.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06w
S=/tmp/weld_r06w2
mkdir -p $O $S
cd /tmp && export TMPDIR=/tmp
ARGS="${C5W_ARGS:-224 256}"
: > $O/c5_walk_latency.txt
for PAD in 0 8192; do
  export WA_WALK_LDS_PAD=$PAD
  echo "# ---- WA_WALK_LDS_PAD=$PAD" >> $O/c5_walk_latency.txt
  python3 $R/tools/c5_walk_counters.py $ARGS >> $O/c5_walk_latency.txt 2>&1
  i=0
  for C in "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum" "TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum" "TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_GMI_32B_sum TCC_EA0_RDREQ_IO_32B_sum" \
           "SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY" "TCP_TCP_LATENCY_sum TCP_TOTAL_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum" "SQ_IFETCH SQ_INSTS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS" \
           "TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_TAG_STALL_sum TCC_REQ_sum TCC_BUSY_sum"; do
    i=$((i+1))
    timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $S/p${PAD}_$i -- python3 $R/tools/c5_walk_counters.py $ARGS > $S/p${PAD}_$i.log 2>&1 || echo "pass $i ($C) failed: $(tail -2 $S/p${PAD}_$i.log | cut -c1-300)" >> $O/c5_walk_latency.txt
  done
  python3 $R/tools/c5_walk_counters.py --summary $S/p${PAD}_? 2>&1 | grep "k_walk_dev\|^##" >> $O/c5_walk_latency.txt
done
cat $O/c5_walk_latency.txt | cut -c1-170

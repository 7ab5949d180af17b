#!/bin/bash
# Counter passes over ONE C5-shaped batch (tools/c5_walk_counters.py): what the walk of 224 concurrent searches on 256^3 fetches, hits, issues and waits for.
#   gpurun --timeout 1500 -- 'bash tools/c5_walk_counters.sh'        -> gpurun_out/r06w/c5_walk_counters.txt
# rocprofv3 needs cwd = /tmp and TMPDIR=/tmp; counter passes are separate runs (--pmc with --kernel-trace only); the program stands directly behind `--`.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06w
S=/tmp/weld_r06w
mkdir -p $O $S
cd /tmp && export TMPDIR=/tmp
ARGS="${C5W_ARGS:-224 256}"
python3 $R/tools/c5_walk_counters.py $ARGS > $O/c5_walk_counters.txt 2>&1
rocprofv3 --kernel-trace --output-format csv -d $S/trace -- python3 $R/tools/c5_walk_counters.py $ARGS >> $O/c5_walk_counters.txt 2>&1
i=0
for C in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
         "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
         "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" "GRBM_GUI_ACTIVE TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $S/pmc$i -- python3 $R/tools/c5_walk_counters.py $ARGS > $S/pmc$i.log 2>&1 || echo "pass $i ($C) failed: $(tail -2 $S/pmc$i.log)" >> $O/c5_walk_counters.txt
done
python3 $R/tools/c5_walk_counters.py --summary $S/trace $S/pmc* >> $O/c5_walk_counters.txt 2>&1
cat $O/c5_walk_counters.txt

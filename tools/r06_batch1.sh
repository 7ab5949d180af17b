#!/bin/bash
# round 6, first GPU batch: (1) the sweep-gap question with counters (VERDICT r05 task 4), (2) occupancy of saturated walk launches at a
# constant tabu table (WA_WALK_LDS_PAD: what would twice the resident walk blocks buy? VERDICT r05 task 1), (3) the driver's line as it stands
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06b1
S=/tmp/weld_r06_scratch
mkdir -p $O $S
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $O/counters_list.txt 2>&1
G=$R/build/sweep_gap_pmc
{
  rocm-smi --showclocks 2>&1 | grep -i -E "sclk|mclk|fclk|socclk" | head -8
  for M in b2b idle valu tiny touch twin idle_b2b2; do $G $M 60; done
  rocm-smi --showclocks 2>&1 | grep -i -E "sclk|mclk|fclk|socclk" | head -8
} > $O/sweep_gap_plain.txt 2>&1
for M in b2b idle valu tiny touch idle_b2b2; do
  rocprofv3 --kernel-trace --output-format csv -d $S/${M}_trace -- $G $M 60 > /dev/null 2>&1
  for C in GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum TCC_EA0_RDREQ_sum TCP_TCC_READ_REQ_sum; do
    timeout 120 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $S/${M}_$C -- $G $M 60 > /dev/null 2>&1
  done
  python3 $R/tools/sweep_gap_report.py $S $M GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum TCC_EA0_RDREQ_sum TCP_TCC_READ_REQ_sum >> $O/sweep_gap_pmc.txt 2>&1
done
cd $R
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench20.json 2> $O/bench20.err
python3 tools/walk_direct_ab.py --c5 --direct 0 --hash 12,11 --pad 0,8192,16384 --reps 3 > $O/occ_c5.jsonl 2>&1
python3 tools/walk_direct_ab.py --ms 32 --kinds lazy --groups 1,2 --direct 0 --hash 12,11 --pad 0,8192,16384 --reps 2 > $O/occ_ms_lazy.jsonl 2>&1
python3 tools/walk_direct_ab.py --ms 16 --kinds dense --groups 1,2 --direct 0 --hash 12,11 --pad 0,8192,16384 --reps 2 > $O/occ_ms_dense.jsonl 2>&1
ls -la $O

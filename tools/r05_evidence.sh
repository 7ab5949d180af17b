#!/bin/bash
# Everything under profiles/r05/ that comes from the GPU, in one go (run on the MI355X box from the repo root through gpurun):
#   gpurun --timeout 3000 -- 'bash tools/r05_evidence.sh'        -> gpurun_out/r05e/*, copied into profiles/r05/ afterwards
# rocprofv3 needs cwd = /tmp and TMPDIR=/tmp; counter passes are separate runs (--pmc with --kernel-trace only); the program stands
# directly behind `--` (python3 script ...): no env / bash -c / launcher in between.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05e
S=/tmp/weld_r05_scratch
mkdir -p $O $S
cd /tmp && export TMPDIR=/tmp
B="--gpus 1 --warmup 5 --no-cpu --no-extras --no-roofline-256"
# 1. the driver's line and the 500-generation line
python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench20.json 2> $O/bench20.err
sleep 8   # (the process before has just given 190 GB back: while the driver wipes them the host's launches are slower -- 63 instead of 43 us per generation seen once)
python3 $R/bench.py --gpus 1 --steps 500 --warmup 5 --no-extras > $O/bench500.json 2>> $O/bench20.err
# 2. kernel trace + stats of warm-up + timed region ONLY (--no-extras: nothing else is enqueued on the solver), cut at the timed region
for K in 20 500; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $S/b$K -- python3 $R/bench.py --steps $K $B > $O/bench${K}_traced.json 2> /dev/null
  cp $(find $S/b$K -name "*kernel_stats.csv" | head -1) $O/bench${K}_kernel_stats.csv
  python3 $R/tools/timed_window_stats.py $S/b$K $O/bench${K}_traced.json --csv $O/bench${K}_window.csv > $O/bench${K}_window.txt 2>&1
done
# 3. the sweep alone: kernel stats (rocprofv3) at 128^3 and 256^3, and its HBM traffic; the in-loop launch's traffic (100 generations)
for N in 128 256; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $S/sw_$N -- python3 $R/tools/sweep_only.py $N > $O/sweep${N}.json 2> /dev/null
  cp $(find $S/sw_$N -name "*kernel_stats.csv" | head -1) $O/sweep${N}_kernel_stats.csv
done
for C in FETCH_SIZE WRITE_SIZE; do
  for N in 128 256; do
    rocprofv3 --pmc $C --kernel-trace --output-format csv -d $S/sw_${C}_$N -- python3 $R/tools/sweep_only.py $N > /dev/null 2>&1
    python3 $R/tools/pmc_summary.py $S/sw_${C}_$N $C > $O/pmc_${C}_sweep$N.csv 2>&1
  done
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $S/b100_$C -- python3 $R/bench.py --steps 100 $B > /dev/null 2>&1
  python3 $R/tools/pmc_summary.py $S/b100_$C $C > $O/pmc_${C}_bench100.csv 2>&1
done
# 4. what a saturated walk launch moves, with the look-ahead (WA_WALK_DIRECT=0, the rule) and without it (=1): eight dense searches, one stream
for D in 0 1; do
  export WA_WALK_DIRECT=$D
  echo "# rocprofv3 --pmc passes over tools/pipeline_curve.py --P 8 --G 1 --kinds dense --gens 40, WA_WALK_DIRECT=$D (0: one-step look-ahead, the rule; 1: no look-ahead)" > $O/pmc_walk_p8_direct$D.txt
  for C in FETCH_SIZE WRITE_SIZE TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum; do
    rocprofv3 --pmc $C --kernel-trace --output-format csv -d $S/wp8_${D}_$C -- python3 $R/tools/pipeline_curve.py --P 8 --G 1 --kinds dense --gens 40 > /dev/null 2>&1
    echo "== $C" >> $O/pmc_walk_p8_direct$D.txt
    python3 $R/tools/pmc_summary.py $S/wp8_${D}_$C $C | head -6 >> $O/pmc_walk_p8_direct$D.txt 2>&1
  done
  rocprofv3 --kernel-trace --stats --output-format csv -d $S/wp8_${D}_t -- python3 $R/tools/pipeline_curve.py --P 8 --G 1 --kinds dense --gens 40 > $O/pipeline_p8_direct$D.json 2> /dev/null
  cp $(find $S/wp8_${D}_t -name "*kernel_stats.csv" | head -1) $O/pipeline_p8_direct${D}_kernel_stats.csv
done
unset WA_WALK_DIRECT
cd $R
# 5. the smaller measurements
python3 tools/walk_direct_ab.py --ms 8,16 --kinds dense,lazy --groups 1,2 --reps 2 > $O/walk_direct_ab_ms.jsonl 2>&1
python3 tools/walk_direct_ab.py --c5 --hash 0,11,13 --reps 3 > $O/walk_direct_ab_c5.jsonl 2>&1
python3 tools/walk_direct_ab.py --c5 --direct 0 --reps 3 --batches > $O/c5_batches.jsonl 2>&1
python3 tools/pipeline_curve.py --P 1,2,4,8,16,32 --G 1,2 --kinds dense,lazy > $O/pipeline_curve.jsonl 2>&1
python3 tools/sweep_nt.py > $O/sweep_nt.txt 2>&1
python3 tools/ref_time.py 500 > $O/ref_time.txt 2>&1
python3 tests/tools/nb26_time.py 300 > $O/nb26_time.txt 2>&1   # (under tests/: it times the CPU oracle beside the kernel, and only tests may load oracle/)
python3 examples/plan_batch.py --grid 256 --points 64 --lazy > $O/plan_batch_c5.jsonl 2>&1
python3 examples/plan_batch.py --grid 256 --points 64 --lazy >> $O/plan_batch_c5.jsonl 2>&1
build/vmm_probe 64 > $O/vmm_probe.txt 2>&1
build/vmm_reuse > $O/vmm_reuse.txt 2>&1
ls -la $O

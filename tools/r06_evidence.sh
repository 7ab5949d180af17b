#!/bin/bash
# Everything under profiles/r06/ that comes from the GPU and belongs to the FINAL build, in one go (run on the MI355X box from the repo root through gpurun):
#   gpurun --timeout 3000 -- 'bash tools/r06_evidence.sh'        -> gpurun_out/r06e/*, copied into profiles/r06/ afterwards
# (the experiments of the round have scripts / tools of their own: tools/r06_batch1.sh, tools/ubench/sweep_gap_pmc.hip, tools/walk_direct_ab.py,
#  tools/lazy_crossover.py, tools/straggler_lazy.py, tools/walk_ab.py, tests/tools/tabu16_model.py)
# rocprofv3 needs cwd = /tmp and TMPDIR=/tmp; counter passes are separate runs (--pmc with --kernel-trace only); the program stands directly behind `--`.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06e
S=/tmp/weld_r06_scratch
mkdir -p $O $S
cd /tmp && export TMPDIR=/tmp
B="--gpus 1 --warmup 5 --no-cpu --no-extras --no-roofline-256"
# 1. the driver's line and the 500-generation line
python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench20.json 2> $O/bench20.err
sleep 8
python3 $R/bench.py --gpus 1 --steps 500 --warmup 5 --no-extras > $O/bench500.json 2>> $O/bench20.err
# 2. kernel trace + stats of warm-up + timed region ONLY, cut at the timed region
for K in 20 500; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $S/b$K -- python3 $R/bench.py --steps $K $B > $O/bench${K}_traced.json 2> /dev/null
  cp $(find $S/b$K -name "*kernel_stats.csv" | head -1) $O/bench${K}_kernel_stats.csv
  python3 $R/tools/timed_window_stats.py $S/b$K $O/bench${K}_traced.json --csv $O/bench${K}_window.csv > $O/bench${K}_window.txt 2>&1
done
# 3. the sweep alone: kernel stats at 128^3 and 256^3, and its HBM traffic; the in-loop launch's traffic (100 generations)
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
cd $R
# 4. the saturated workloads as the final rule runs them
python3 tools/pipeline_curve.py --P 1,2,4,8,16,32 --G 1,2 --kinds dense,lazy > $O/pipeline_curve.jsonl 2>&1
python3 tools/pipeline_curve.py --P 8,16,32 --G 0 --kinds dense,lazy > $O/pipeline_curve_by_rule.jsonl 2>&1              # (--G 0: the library's rule -- three groups for lazy batches of big colonies)
python3 tools/pipeline_curve.py --P 8,16,32 --G 0 --kinds dense,lazy --gens 500 > $O/pipeline_curve_500_by_rule.jsonl 2>&1
python3 examples/plan_batch.py --grid 256 --points 64 --lazy > $O/plan_batch_c5.jsonl 2>&1
python3 examples/plan_batch.py --grid 256 --points 64 --lazy >> $O/plan_batch_c5.jsonl 2>&1
python3 tools/ref_time.py 500 > $O/ref_time.txt 2>&1
python3 tests/tools/nb26_time.py 300 > $O/nb26_time.txt 2>&1
ls -la $O

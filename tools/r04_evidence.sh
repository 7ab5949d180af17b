#!/bin/bash
# Everything under profiles/r04/ that comes from the GPU, in one go (run on the MI355X box from the repo root through gpurun):
#   gpurun --timeout 3000 -- 'bash tools/r04_evidence.sh'        -> gpurun_out/r04/*, copied into profiles/r04/ afterwards
# rocprofv3 needs cwd = /tmp and TMPDIR=/tmp; counter passes are separate runs (--pmc with --kernel-trace only).
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04
S=/tmp/weld_r04_scratch
mkdir -p $O $S
cd /tmp && export TMPDIR=/tmp
B="--gpus 1 --warmup 5 --no-cpu --no-extras --no-roofline-256"
# 1. the driver's line and the 500-generation line
python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench20.json 2> $O/bench20.err
python3 $R/bench.py --gpus 1 --steps 500 --warmup 5 --no-extras > $O/bench500.json 2>> $O/bench20.err
# 2. kernel stats of the same commands
rocprofv3 --kernel-trace --stats --output-format csv -d $S/b20 -- python3 $R/bench.py --steps 20 $B > /dev/null 2>&1
cp $(find $S/b20 -name "*kernel_stats.csv" | head -1) $O/bench20_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $S/b500 -- python3 $R/bench.py --steps 500 $B > /dev/null 2>&1
cp $(find $S/b500 -name "*kernel_stats.csv" | head -1) $O/bench500_kernel_stats.csv
# 3. HBM traffic: the sweep alone at 128^3 / 256^3 and the in-loop launch (100 generations)
for C in FETCH_SIZE WRITE_SIZE; do
  for N in 128 256; do
    rocprofv3 --pmc $C --kernel-trace --output-format csv -d $S/sw_${C}_$N -- python3 $R/tools/sweep_only.py $N > /dev/null 2>&1
    python3 $R/tools/pmc_summary.py $S/sw_${C}_$N $C > $O/pmc_${C}_sweep$N.csv 2>&1
  done
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $S/b100_$C -- python3 $R/bench.py --steps 100 $B > /dev/null 2>&1
  python3 $R/tools/pmc_summary.py $S/b100_$C $C > $O/pmc_${C}_bench100.csv 2>&1
done
# 4. pipelined groups: eight dense searches on one stream and in two groups (kernel trace), the problems-per-GPU curve
for G in 1 2; do
  rocprofv3 --kernel-trace --output-format csv -d $S/tr_g$G -- python3 $R/tools/pipeline_curve.py --P 8 --G $G --kinds dense --gens 100 > $O/pipeline_p8_g$G.json 2>&1
  python3 $R/tools/trace_overlap.py $S/tr_g$G > $O/pipeline_p8_g${G}_overlap.txt 2>&1
  python3 $R/tools/trace_overlap.py $S/tr_g$G --series > $O/pipeline_p8_g${G}_series.txt 2>&1
  python3 $R/tools/trace_overlap.py $S/tr_g$G --skip-first 400 --timeline 80 > $O/pipeline_p8_g${G}_timeline.txt 2>&1
done
cd $R
python3 tools/pipeline_curve.py --P 1,2,4,8,16,32 --G 1,2,3,4 --kinds dense,lazy > $O/pipeline_curve.jsonl 2>&1
# 5. the smaller measurements
python3 tools/sweep_nt.py > $O/sweep_nt.txt 2>&1
python3 tools/ref_time.py 500 > $O/ref_time.txt 2>&1
WA_REF_SPEC=0 python3 tools/ref_time.py 500 2>&1 | head -1 >> $O/ref_time.txt
python3 tools/ref_profile.py 500 > $O/ref_profile.txt 2>&1
python3 tools/scalar_calls.py $O/scalar_calls.txt > /dev/null 2>&1
python3 tests/tools/nb26_time.py 300 > $O/nb26_time.txt 2>&1
python3 examples/plan_batch.py --grid 256 --points 64 --lazy > $O/plan_batch_c5.jsonl 2>&1
python3 examples/plan_batch.py --grid 256 --points 64 --lazy >> $O/plan_batch_c5.jsonl 2>&1

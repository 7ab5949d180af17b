set -x
mkdir -p gpurun_out/r03
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python bench.py --steps 20 --warmup 5 > gpurun_out/r03/bench20.json 2> gpurun_out/r03/bench20.err
python bench.py > gpurun_out/r03/bench500.json 2> gpurun_out/r03/bench500.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03/kt20 -- python3 bench.py --steps 20 --warmup 5 --no-cpu --no-extras --no-roofline-256 > gpurun_out/r03/bench20_under_rocprof.json 2>gpurun_out/r03/kt20.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03/kt500 -- python3 bench.py --no-cpu --no-extras --no-roofline-256 > gpurun_out/r03/bench500_under_rocprof.json 2>gpurun_out/r03/kt500.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/r03/pmc_fetch -- python3 bench.py --steps 100 --warmup 5 --no-cpu --no-extras --no-roofline-256 > /dev/null 2>gpurun_out/r03/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/r03/pmc_write -- python3 bench.py --steps 100 --warmup 5 --no-cpu --no-extras --no-roofline-256 > /dev/null 2>gpurun_out/r03/pmc_write.err
python tools/pmc_summary.py gpurun_out/r03/pmc_fetch FETCH_SIZE > gpurun_out/r03/pmc_fetch_summary.csv
python tools/pmc_summary.py gpurun_out/r03/pmc_write WRITE_SIZE > gpurun_out/r03/pmc_write_summary.csv
find gpurun_out/r03 -name "*kernel_stats.csv" | head
du -sh gpurun_out/r03
# keep only summaries (the raw traces are large)
for d in kt20 kt500; do f=$(find gpurun_out/r03/$d -name "*kernel_stats.csv" | head -1); cp $f gpurun_out/r03/${d}_kernel_stats.csv; done
rm -rf gpurun_out/r03/kt20 gpurun_out/r03/kt500 gpurun_out/r03/pmc_fetch gpurun_out/r03/pmc_write

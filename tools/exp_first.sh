mkdir -p gpurun_out/r03
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "acs_dev or batch_of_problems or tabu_spill" > gpurun_out/r03/ovl_first.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r03/ovl_first.log
tail -3 gpurun_out/r03/ovl_first.log
for ov in 1 0; do echo WA_OVERLAP=$ov; WA_OVERLAP=$ov timeout 300 python tools/gen_loop_time.py; done
show() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['value']), {k: round(v*1e3,1) for k,v in d['kernel_ms_per_generation'].items()}, d.get('cost_check',{}).get('bit_equal_trace'))"; }
for ov in 1 0 1 0; do for st in 20 500; do
WA_OVERLAP=$ov timeout 300 python bench.py --steps $st --warmup 5 --cpu-gens 3 --no-extras --no-roofline-256 | show "overlap=$ov steps=$st"
done; done

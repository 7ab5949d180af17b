B="python bench.py --no-cpu --no-extras --no-roofline-256"
P='import json,sys; d=json.loads(sys.stdin.read()); print("%8.1f gen/s  %s" % (d["value"], {k: round(v,4) for k,v in d["kernel_ms_per_generation"].items()}))'
for k in 20 500; do
echo "== steps $k: default (fused sweep+rank+mark)"; for i in 1 2; do $B --steps $k --warmup 5 | python -c "$P"; done
for blocks in 4096 512 128 32; do
echo "== steps $k: WA_SWEEP_STREAM=1, sweep grid $blocks"; for i in 1 2; do WA_SWEEP_STREAM=1 WA_SWEEP_STREAM_BLOCKS=$blocks $B --steps $k --warmup 5 | python -c "$P"; done
done
done
WA_SWEEP_STREAM=1 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edges.py -m gpu -x -q 2>&1 | tail -2

show() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['value']), {k: round(v*1e3,1) for k,v in d['kernel_ms_per_generation'].items()})"; }
for rep in 1 2; do
for L in build/exp/ovl.so build/exp/ovl_ntl.so build/exp/ovl_ntls.so; do
for cfg in "0 0" "15 0" "3 0"; do
  set -- $cfg
  WA_EXP_OVERLAP_WAVES=$1 WA_EXP_OVERLAP_CONTROL=$2 WELDACS_LIB=$PWD/$L python bench.py --steps 20 --warmup 5 --no-cpu --no-extras --no-roofline-256 | show "$L waves=$1"
done
done
done

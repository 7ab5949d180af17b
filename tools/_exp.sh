cd $GRAFT_REPO_ROOT
run() { echo "== $*"; env "$@" python tools/pipeline_curve.py --P $P --G $G --kinds $K 2>&1 | python3 -c "
import sys, json
for l in sys.stdin:
    try: j = json.loads(l)
    except Exception: print(l.strip()); continue
    k = j['kernel_ms_per_launch']
    print('%s P=%d G=%d  %.1f k pg/s  %.1f us/gen-of-all  host %.1f  walk %.1f sweep %.1f apply %.1f  same=%s' % (j['kind'], j['P'], j['groups'], j['problem_generations_per_s']/1e3, j['ms_per_generation_of_all']*1e3, j['host_enqueue_ms_per_generation_of_all']*1e3, k['walk']*1e3, k['evaporate']*1e3, k['deposit']*1e3, j['identical_to_first']))
"; }
export WA_SWEEP_NT=3
for B in 1024 2048 3072 6144 8192; do K=dense P=8 G=1,2 run WA_EVAP_BLOCKS=$B; done
for S in 16 32 128; do K=dense P=8 G=2 run WA_STRAGGLER_GENS=$S; done
K=dense P=8 G=2 run WA_HASH_LOG2=11
K=dense P=8 G=2 run WA_REENTRY_STABLE=8
K=dense P=8 G=2 run WA_REENTRY_STABLE=32

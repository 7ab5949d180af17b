mkdir -p gpurun_out/r03
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for w in 15 12; do
  rm -rf /tmp/ovt_$w
  WA_EXP_OVERLAP_WAVES=$w WELDACS_LIB=$R/build/exp/ovl.so rocprofv3 --kernel-trace --output-format csv -d /tmp/ovt_$w -- python3 $R/bench.py --steps 500 --warmup 5 --no-cpu --no-extras --no-roofline-256 --profile-every 0 > /tmp/ovt_$w.log 2>&1
  python3 - /tmp/ovt_$w $w <<'PY'
import csv, glob, sys, statistics as st
rows=[]
for f in glob.glob(sys.argv[1]+"/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:40]))
rows.sort()
gens=[]
for i in range(len(rows)-3):
    a,b,c,d=rows[i:i+4]
    if "k_walk" in a[2] and "k_evap_rank_mark" in b[2] and "k_apply_table" in c[2] and "k_walk" in d[2]:
        gens.append((a[1]-a[0], b[0]-a[1], b[1]-b[0], c[0]-b[1], c[1]-c[0], d[0]-c[1], a[2]))
print("waves", sys.argv[2], "generations", len(gens), "kernel", gens[0][6] if gens else None)
conv=[g for g in gens[-300:]]
for j,lab in enumerate(("walk(+sweep)","gap","rank+mark","gap","apply+table","gap")):
    print("   last 300 generations: %-14s median %7.2f us mean %7.2f" % (lab, st.median([g[j] for g in conv])/1e3, st.mean([g[j] for g in conv])/1e3))
print("   sum of means %.2f us" % (sum(st.mean([g[j] for g in conv]) for j in range(6))/1e3))
PY
done

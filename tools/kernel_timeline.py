import csv, glob, sys
rows=[]
for f in glob.glob(sys.argv[1]+"/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:40]))
rows.sort()
# find the timed region: consecutive k_walk_dev -> k_evap_rank_mark -> k_apply_table triples
out=[]
for i in range(len(rows)-3):
    a,b,c,d=rows[i],rows[i+1],rows[i+2],rows[i+3]
    if "k_walk_dev" in a[2] and "k_evap_rank_mark" in b[2] and "k_apply_table" in c[2] and "k_walk_dev" in d[2]:
        out.append((a[1]-a[0], b[0]-a[1], b[1]-b[0], c[0]-b[1], c[1]-c[0], d[0]-c[1]))
import statistics as st
def col(j, sel): return [o[j] for o in sel]
# (the first triples of the trace are the search's first generations: bench.py's warm-up and timed generations follow each other)
for name, sel in (("generations 0-19 of the search", out[:20]), ("generations 20-39", out[20:40]), ("exploratory (walk > 100 us)", [o for o in out if o[0]>100000]), ("converged (walk < 12 us)", [o for o in out if o[0]<12000])):
    if not sel: continue
    print("%s: %d generations" % (name, len(sel)))
    for j,lab in enumerate(("walk kernel","gap walk->fused","fused kernel","gap fused->apply","apply+table kernel","gap apply->next walk")):
        print("   %-22s median %8.2f us   mean %8.2f us" % (lab, st.median(col(j,sel))/1e3, st.mean(col(j,sel))/1e3))
    print("   %-22s mean %8.2f us per generation" % ("sum", sum(st.mean(col(j,sel)) for j in range(6))/1e3))

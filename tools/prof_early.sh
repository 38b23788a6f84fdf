OUT=/root/repo/gpurun_out/tlq; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/trace -o tl -- python /root/repo/bench.py --steps 2 --warmup 1 --no-cpu --warm-steps 0 --stream-reads 8000000 > $OUT/bench.json 2> $OUT/trace.err
F=$(find $OUT -name "*kernel_trace.csv" | head -1)
python /root/repo/tools/timeline.py $F 0.0 > $OUT/timeline_all.txt 2>&1
M=$(find $OUT -name "*memory_copy_trace.csv" | head -1)
python - "$F" "$M" > $OUT/early.txt <<'PY'
import csv, sys
rows=[]
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].split(" ")[-1], "q"+r["Queue_Id"]))
for r in csv.DictReader(open(sys.argv[2])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY "+r.get("Direction","")+" "+r.get("Bytes", r.get("Size","")), "dma"))
rows.sort()
chains=[r for r in rows if r[2]=="k_arith_chain" and r[1]-r[0]>30e6]
c=chains[-1]
first=[r for r in rows if r[2]=="k_nl_count" and r[0]<c[0]][-1]
t0=first[0]
for s,e,n,q in rows:
    if t0<=s<=c[0]+2e6 and (q!="q1" or e-s>0.2e6):
        print("%9.3f %9.3f %-4s %s"%((s-t0)/1e6,(e-t0)/1e6,q,n))
PY
find $OUT -name "*_trace.csv" -delete

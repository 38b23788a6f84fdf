# when does the HOST hand the long streams' first work to the second handle, and when does the DEVICE run it? (streamed form)
OUT=/root/repo/gpurun_out/tlq; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --hip-runtime-trace --memory-copy-trace --output-format csv -d $OUT/trace -o tl -- python /root/repo/bench.py --steps 2 --warmup 1 --no-cpu --warm-steps 0 --stream-reads 8000000 > $OUT/bench.json 2> $OUT/trace.err
ls $OUT/trace/*/ 2>/dev/null | head; F=$(find $OUT -name "*kernel_trace.csv" | head -1); A=$(find $OUT -name "*hip_api_trace.csv" | head -1); M=$(find $OUT -name "*memory_copy_trace.csv" | head -1)
python - "$F" "$A" "$M" > $OUT/early_api.txt <<'PY'
import csv, sys
ker=[(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].split(" ")[-1], "q"+r["Queue_Id"]) for r in csv.DictReader(open(sys.argv[1]))]
ker.sort()
chains=[r for r in ker if r[2]=="k_arith_chain" and r[1]-r[0]>30e6]
c=chains[-1]
t0=[r for r in ker if r[2]=="k_nl_count" and r[0]<c[0]][-1][0]
rows=[(s,e,n,q) for s,e,n,q in ker if t0<=s<=c[0]+1e6 and (q!="q1" or e-s>0.3e6)]
api=list(csv.DictReader(open(sys.argv[2])))
print("api columns", list(api[0].keys()))
for r in api:
    s=int(r["Start_Timestamp"]); e=int(r["End_Timestamp"])
    if t0-1e6<=s<=c[0]+1e6 and (e-s>0.15e6 or r["Function"] in ("hipEventSynchronize","hipStreamSynchronize","hipEventRecord","hipStreamWaitEvent")):
        rows.append((s,e,"API "+r["Function"],"host"))
rows.sort()
for s,e,n,q in rows: print("%9.3f %9.3f %-5s %s"%((s-t0)/1e6,(e-t0)/1e6,q,n))
PY
find $OUT -name "*_trace.csv" -delete

# timings, then rocprofv3 WRITE_SIZE / FETCH_SIZE per dispatch (every configuration is launched 4 times: the last one is listed)
OUT=/root/repo/gpurun_out/scatter; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
/root/repo/tools/ubench_scatter.bin 1024 > $OUT/times.txt; cat $OUT/times.txt
for C in WRITE_SIZE FETCH_SIZE; do
  timeout 600 rocprofv3 --pmc $C --output-format csv -d $OUT/$C -o sc -- /root/repo/tools/ubench_scatter.bin 1024 > /dev/null 2> $OUT/$C.err
  F=$(find $OUT/$C -name "*counter_collection.csv" | head -1)
  python3 - "$F" $C <<'PY' > $OUT/$C.txt
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if r["Counter_Name"] == sys.argv[2]]
rows.sort(key=lambda r: int(r["Dispatch_Id"]))
vals = [float(r["Counter_Value"]) for r in rows]
print(" ".join("%.0f" % (vals[i + 3]) for i in range(0, len(vals) - 3, 4)))
PY
  echo "$C (KB reported, per configuration in the order above):"; cat $OUT/$C.txt
done
find $OUT -name "*counter_collection.csv" -delete

# SQ instruction counters of the model kernel (serialised kernels: GZ_NO_PIPELINE, codecs pinned): bash tools/prof_sq.sh <tag>
TAG=${1:-sq}
OUT=/root/repo/gpurun_out/sq_$TAG; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
for C in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES"; do
  N=$(echo $C | cut -d' ' -f1)
  GZ_NO_PIPELINE=1 GZ_ZIP_NO_OVERLAP=1 timeout 900 rocprofv3 --pmc $C --output-format csv -d $OUT/$N -o sq -- python /root/repo/bench.py --steps 2 --warmup 1 --no-cpu --pin-codecs --warm-steps 0 > $OUT/bench_$N.json 2> $OUT/$N.err
  F=$(find $OUT/$N -name "*counter_collection.csv" | head -1)
  python - "$F" <<'PY' > $OUT/$N.txt
import csv, sys, collections
tot = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0]; tot[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
for k in sorted(tot, key=lambda k: -sum(tot[k].values()))[:12]:
    print(k, {c: int(v) for c, v in tot[k].items()}, "dispatches", max(n[(k, c)] for c in tot[k]))
PY
  cat $OUT/$N.txt
done
find $OUT -name "*counter_collection.csv" -size +5M -delete; du -sh $OUT

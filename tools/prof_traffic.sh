# HBM counters of one configuration with an environment of choice (separate --pmc passes, kernels serialised: GZ_NO_PIPELINE):   prof_traffic.sh <tag> [VAR=value ...] -- <bench args>
set -x
TAG=$1; shift
ENVS=""; while [ "$1" != "--" ] && [ $# -gt 0 ]; do ENVS="$ENVS $1"; shift; done; shift
OUT=/root/repo/gpurun_out/traffic_$TAG; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  env $ENVS GZ_NO_PIPELINE=1 GZ_ZIP_NO_OVERLAP=1 timeout 600 rocprofv3 --pmc $C --output-format csv -d $OUT/$C -o t -- python /root/repo/bench.py --steps 2 --warmup 1 --no-cpu --pin-codecs --warm-steps 0 "$@" > $OUT/bench_$C.json 2> $OUT/$C.err
done
python - $OUT <<'P' > /root/repo/gpurun_out/traffic_$TAG.txt
import collections, csv, glob, sys
out = sys.argv[1]
tot = collections.defaultdict(lambda: [0.0, 0.0])
for i, c in enumerate(("FETCH_SIZE", "WRITE_SIZE")):
    for path in glob.glob(out + "/" + c + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(path)):
            name = r["Kernel_Name"]; name = name[5:] if name.startswith("void ") else name
            name = name.split("(")[0].split("<")[0]
            if name.startswith("k_") and r["Counter_Name"] == c: tot[name][i] += float(r["Counter_Value"])
STEPS = 3
rows = sorted(tot.items(), key=lambda kv: -(2 * kv[1][0] + kv[1][1]))
print("per step (3 identical steps in the pass; traffic = 2 x FETCH_SIZE + WRITE_SIZE, KB -> GB):")
s = 0
for k, (f, w) in rows[:14]:
    print("%-24s fetch %6.2f GB  write %6.2f GB  total %6.2f" % (k, 2 * f * 1024 / 1e9 / STEPS, w * 1024 / 1e9 / STEPS, (2 * f + w) * 1024 / 1e9 / STEPS))
for k, (f, w) in rows: s += (2 * f + w) * 1024 / 1e9 / STEPS
print("all kernels: %.2f GB per step" % s)
P
find $OUT -name "*counter_collection.csv" -delete

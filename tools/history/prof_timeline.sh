set -x
OUT=/root/repo/gpurun_out/tl; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o tl -- python /root/repo/bench.py --steps 2 --warmup 1 --no-cpu > $OUT/bench.json 2> $OUT/trace.err
F=$(find $OUT -name "*kernel_trace.csv" | head -1)
python /root/repo/tools/timeline.py $F > $OUT/timeline.txt 2>&1
find $OUT -name "*kernel_trace.csv" -delete; du -sh $OUT

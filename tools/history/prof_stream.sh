# the streamed form (configs[4] at reduced scale): host phases, kernel sums and the long dispatches of one call
set -x
OUT=/root/repo/gpurun_out/strm; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
GZ_ZIP_TIMING=1 python /root/repo/bench.py --stream-reads 8000000 --steps 2 --warmup 1 --no-cpu --warm-steps 0 > $OUT/bench_plain.json 2> $OUT/plain.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o st -- python /root/repo/bench.py --stream-reads 8000000 --steps 2 --warmup 1 --no-cpu --warm-steps 0 > $OUT/bench.json 2> $OUT/trace.err
F=$(find $OUT -name "*kernel_trace.csv" | head -1)
python /root/repo/tools/timeline.py $F 3 > $OUT/timeline.txt 2>&1
S=$(find $OUT -name "*kernel_stats.csv" | head -1); head -25 $S > $OUT/kernel_stats_head.csv
find $OUT -name "*kernel_trace.csv" -delete; du -sh $OUT

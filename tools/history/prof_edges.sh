# the step's lead-in and tail in detail (every dispatch of the first / last 12 ms) + the host's phase times of an unprofiled run
set -x
OUT=/root/repo/gpurun_out/edges; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
GZ_ZIP_TIMING=1 python /root/repo/bench.py --steps 3 --warmup 1 --no-cpu --warm-steps 0 > $OUT/bench_plain.json 2> $OUT/plain.err
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o tl -- python /root/repo/bench.py --steps 2 --warmup 1 --no-cpu --warm-steps 0 > $OUT/bench.json 2> $OUT/trace.err
F=$(find $OUT -name "*kernel_trace.csv" | head -1)
python /root/repo/tools/timeline.py $F 0.01 12 > $OUT/timeline_edges.txt 2>&1
python /root/repo/tools/timeline.py $F > $OUT/timeline.txt 2>&1
find $OUT -name "*kernel_trace.csv" -delete; du -sh $OUT

# round 5, second session: rocprofv3 kernel statistics + timeline of the default command's workload with the final binary (one gpurun call)
set -x
OUT=/root/repo/gpurun_out/prof_r05b; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o r05b -- python /root/repo/bench.py --steps 3 --warmup 1 --no-cpu --warm-steps 0 > $OUT/bench_stats.json 2> $OUT/trace.err
F=$(find $OUT/trace -name "*kernel_trace.csv" | head -1)
python /root/repo/tools/timeline.py $F 0.3 > /root/repo/gpurun_out/r05b_timeline.txt 2>&1
S=$(find $OUT/trace -name "*kernel_stats.csv" | head -1); cp $S /root/repo/gpurun_out/r05b_kernel_stats.csv
find $OUT -name "*_kernel_trace.csv" -delete; du -sh $OUT
head -5 /root/repo/gpurun_out/r05b_kernel_stats.csv | cut -c1-200

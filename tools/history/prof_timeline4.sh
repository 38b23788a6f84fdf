# kernel timeline + stats of one bench command (round 4): bash tools/prof_timeline4.sh <tag> <bench args...>
set -x
TAG=$1; shift
OUT=/root/repo/gpurun_out/tl4_$TAG; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o tl -- python /root/repo/bench.py --steps 3 --warmup 1 --no-cpu --warm-steps 0 "$@" > $OUT/bench.json 2> $OUT/trace.err
F=$(find $OUT -name "*kernel_trace.csv" | head -1)
python /root/repo/tools/timeline.py $F 0.3 > $OUT/timeline.txt 2>&1
S=$(find $OUT -name "*kernel_stats.csv" | head -1); cp $S $OUT/kernel_stats.csv
find $OUT -name "*kernel_trace.csv" -delete; du -sh $OUT

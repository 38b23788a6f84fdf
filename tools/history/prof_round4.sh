# round 4: rocprofv3 kernel stats + HBM counters of the default bench command (one gpurun call). PMC passes: separate runs, serialised
# kernels (GZ_NO_PIPELINE: no persistent kernel), codecs pinned (no trial compressions), as tools/prof_round2.sh
set -x
OUT=/root/repo/gpurun_out/prof_r4; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o r4 -- python /root/repo/bench.py --steps 3 --warmup 1 --no-cpu --warm-steps 0 > $OUT/bench_stats.json 2> $OUT/trace.err
GZ_NO_PIPELINE=1 GZ_ZIP_NO_OVERLAP=1 timeout 1200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o r4 -- python /root/repo/bench.py --steps 2 --warmup 1 --no-cpu --pin-codecs --warm-steps 0 > $OUT/bench_fetch.json 2> $OUT/fetch.err
GZ_NO_PIPELINE=1 GZ_ZIP_NO_OVERLAP=1 timeout 1200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o r4 -- python /root/repo/bench.py --steps 2 --warmup 1 --no-cpu --pin-codecs --warm-steps 0 > $OUT/bench_write.json 2> $OUT/write.err
F=$(find $OUT/trace -name "*kernel_trace.csv" | head -1)
python /root/repo/tools/timeline.py $F > $OUT/timeline.txt 2>&1
find $OUT -name "*_kernel_trace.csv" -delete; find $OUT -size +20M -delete; du -sh $OUT

set -x
OUT=/root/repo/gpurun_out/prof2; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o r2 -- python /root/repo/bench.py --steps 3 --warmup 1 --no-cpu > $OUT/bench_stats.json 2> $OUT/trace.err
GZ_NO_PIPELINE=1 GZ_ZIP_NO_OVERLAP=1 timeout 1200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o r2 -- python /root/repo/bench.py --steps 2 --warmup 1 --no-cpu --pin-codecs > $OUT/bench_fetch.json 2> $OUT/fetch.err
GZ_NO_PIPELINE=1 GZ_ZIP_NO_OVERLAP=1 timeout 1200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o r2 -- python /root/repo/bench.py --steps 2 --warmup 1 --no-cpu --pin-codecs > $OUT/bench_write.json 2> $OUT/write.err
find $OUT -name "*.csv" | head -20; du -sh $OUT; find $OUT -name "*_kernel_trace.csv" -delete; find $OUT -size +20M -delete; du -sh $OUT

# round 6: rocprofv3 kernel stats + HBM counters of the default bench command (one gpurun call). PMC passes: separate runs, serialised
# kernels (GZ_NO_PIPELINE: no persistent kernel), codecs pinned (no trial compressions), as tools/history/prof_round2.sh
set -x
OUT=/root/repo/gpurun_out/prof_r06; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o r06 -- python /root/repo/bench.py --steps 3 --warmup 1 --no-cpu --warm-steps 0 > $OUT/bench_stats.json 2> $OUT/trace.err
GZ_NO_PIPELINE=1 GZ_ZIP_NO_OVERLAP=1 timeout 1200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o r06 -- python /root/repo/bench.py --steps 2 --warmup 1 --no-cpu --pin-codecs --warm-steps 0 > $OUT/bench_fetch.json 2> $OUT/fetch.err
GZ_NO_PIPELINE=1 GZ_ZIP_NO_OVERLAP=1 timeout 1200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o r06 -- python /root/repo/bench.py --steps 2 --warmup 1 --no-cpu --pin-codecs --warm-steps 0 > $OUT/bench_write.json 2> $OUT/write.err
F=$(find $OUT/trace -name "*kernel_trace.csv" | head -1)
python /root/repo/tools/timeline.py $F > $OUT/timeline.txt 2>&1
cd /root/repo
python tools/summarize_prof.py $OUT profiles r06 r06 '{"pairs": 1000000, "vb_bytes": 14720000, "qual": "div"}' > $OUT/summary.txt 2>&1
cp $OUT/timeline.txt profiles/r06_timeline.txt
# the other configurations' timelines and kernel stats
for cfg in "bam:--config bam" "bin:--qual bin" "stream:--stream-reads 8000000"; do
  tag=${cfg%%:*}; args=${cfg#*:}
  O2=/root/repo/gpurun_out/tl06_$tag; mkdir -p $O2; cd /tmp
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O2/trace -o tl -- python /root/repo/bench.py --steps 3 --warmup 1 --no-cpu --warm-steps 0 $args > $O2/bench.json 2> $O2/trace.err
  F=$(find $O2 -name "*kernel_trace.csv" | head -1)
  python /root/repo/tools/timeline.py $F 0.3 > /root/repo/profiles/r06_timeline_$tag.txt 2>&1
  S=$(find $O2 -name "*kernel_stats.csv" | head -1); cp $S /root/repo/profiles/r06_kernel_stats_$tag.csv
  find $O2 -name "*kernel_trace.csv" -delete
done
cd /root/repo
find $OUT -name "*_kernel_trace.csv" -delete; find $OUT -size +20M -delete; du -sh $OUT
mkdir -p gpurun_out/profiles_r6; cp profiles/r06_* gpurun_out/profiles_r6/

# timeline of one step of configs[2] (text and BAM records): rocprofv3 --kernel-trace of bench.py --config bam [--bam-binary]
set -x
OUT=/root/repo/gpurun_out/tlbam; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
for V in text bin; do
  FLAG=""; [ $V = bin ] && FLAG="--bam-binary"
  GZ_ZIP_TIMING=1 timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_$V -o tl -- python /root/repo/bench.py --config bam $FLAG --steps 2 --warmup 1 --no-cpu > $OUT/bench_$V.json 2> $OUT/trace_$V.err
  F=$(find $OUT/trace_$V -name "*kernel_trace.csv" | head -1)
  python /root/repo/tools/timeline.py $F 0.4 > $OUT/timeline_$V.txt 2>&1
  find $OUT/trace_$V -name "*kernel_trace.csv" -delete
  grep -i "zip\|seg\|merge\|finish" $OUT/trace_$V.err | tail -30 > $OUT/host_$V.txt
done
du -sh $OUT

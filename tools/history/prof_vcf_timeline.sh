# GPU: kernel timeline of one VCF VBlock from text (tools/vcf_model_probe.py, 2 steps)
set -x
OUT=/root/repo/gpurun_out/vcftl; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o tl -- python /root/repo/tools/vcf_model_probe.py ${1:-3000} > $OUT/out.txt 2> $OUT/log.txt
F=$(find $OUT -name "*kernel_trace.csv" | head -1)
python /root/repo/tools/timeline.py $F > $OUT/timeline.txt 2>&1
find $OUT -name "*kernel_trace.csv" -delete; du -sh $OUT

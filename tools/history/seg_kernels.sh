# per-kernel times of the seg-side kernels (QUAL gather, DOMQ) from a kernel trace of the BAM and binned-FASTQ steps
set -x
OUT=/root/repo/gpurun_out/segk; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
for V in bam bin; do
  FLAG="--config bam"; [ $V = bin ] && FLAG="--qual bin"
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$V -o tl -- python /root/repo/bench.py $FLAG --steps 3 --warmup 1 --no-cpu --warm-steps 0 > $OUT/bench_$V.json 2> $OUT/trace_$V.err
  S=$(find $OUT/trace_$V -name "*kernel_stats.csv" | head -1)
  grep -i "k_domq\|k_blob\|k_nl_\|k_lines\|k_col_insert\|k_icol\|k_tokenize\|k_fastq_rec\|k_acgt" $S | cut -d, -f1-4 > $OUT/stats_$V.txt
  find $OUT/trace_$V -name "*kernel_trace.csv" -delete
done

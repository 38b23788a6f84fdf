# every number of profiles/round2_* in one go (one gpurun call): benches, configs, probes, rocprofv3 stats + HBM counters
set -x
OUT=/root/repo/gpurun_out/fin; mkdir -p $OUT; cd /root/repo
timeout 300 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
timeout 200 python bench.py --vb-mb 16 --no-cpu > $OUT/bench_vb16.json 2>/dev/null
timeout 200 python bench.py --vb-mb 4 --no-cpu > $OUT/bench_vb4.json 2>/dev/null
timeout 200 python bench.py --qual bin --no-cpu > $OUT/bench_bin.json 2>/dev/null
timeout 300 python bench.py --stream-reads 8000000 --steps 2 --warmup 1 --no-cpu > $OUT/bench_stream.json 2>/dev/null
timeout 400 python bench.py --stream-reads 75000000 --steps 1 --warmup 1 --no-cpu > $OUT/bench_stream_wgs.json 2>/dev/null
timeout 400 python tools/config_bench.py bam > $OUT/config_bam.json 2> $OUT/config_bam.err
timeout 400 python tools/config_bench.py vcf > $OUT/config_vcf.json 2> $OUT/config_vcf.err
timeout 200 python tools/model_probe.py > $OUT/model_probe.txt 2>&1
bash tools/prof_round2.sh > $OUT/prof.log 2>&1
bash tools/prof_round2_sq.sh > $OUT/prof_sq.log 2>&1
bash tools/prof_timeline.sh > $OUT/tl.log 2>&1
find /root/repo/gpurun_out -size +25M -delete; du -sh /root/repo/gpurun_out

# round 4, second final pass (after the rounds / LDS counts / XCD-aware grid): suite + bench lines, kernel stats + HBM counters, timelines, the scatter calibration
bash tools/final_round4.sh > gpurun_out/fin4.log 2>&1; tail -12 gpurun_out/fin4.log
bash tools/prof_round4.sh > gpurun_out/prof_r4.log 2>&1
bash tools/prof_timeline4.sh default > /dev/null 2>&1
bash tools/prof_timeline4.sh bin --qual bin > /dev/null 2>&1
bash tools/prof_timeline4.sh stream --stream-reads 8000000 > /dev/null 2>&1
bash tools/prof_timeline4.sh bam --config bam > /dev/null 2>&1
bash tools/prof_timeline4.sh vcf --config vcf > /dev/null 2>&1
bash tools/ubench_scatter.sh > gpurun_out/scatter.log 2>&1

# a rank's share of configs[4] (75 M read pairs: 15 calls per file), both quality profiles
python bench.py --stream-reads 75000000 --steps 2 --warmup 1 --no-cpu --warm-steps 0 > gpurun_out/fin4/bench_stream75m.json 2> gpurun_out/fin4/bench_stream75m.err
python bench.py --stream-reads 75000000 --qual bin --steps 2 --warmup 1 --no-cpu --warm-steps 0 > gpurun_out/fin4/bench_stream75m_bin.json 2> gpurun_out/fin4/bench_stream75m_bin.err

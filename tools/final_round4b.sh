# round 4, second final pass (after the rounds / LDS counts / XCD-aware grid): suite + bench lines, kernel stats + HBM counters, timelines, the scatter calibration
bash tools/final_round4.sh > gpurun_out/fin4.log 2>&1; tail -12 gpurun_out/fin4.log
bash tools/prof_round4.sh > gpurun_out/prof_r4.log 2>&1
bash tools/prof_timeline4.sh default > /dev/null 2>&1
bash tools/prof_timeline4.sh bin --qual bin > /dev/null 2>&1
bash tools/prof_timeline4.sh stream --stream-reads 8000000 > /dev/null 2>&1
bash tools/prof_timeline4.sh bam --config bam > /dev/null 2>&1
bash tools/prof_timeline4.sh vcf --config vcf > /dev/null 2>&1
bash tools/ubench_scatter.sh > gpurun_out/scatter.log 2>&1
ls gpurun_out

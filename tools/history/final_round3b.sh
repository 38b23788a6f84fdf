# round 3, last pass on the GPU box (one gpurun call): the -m gpu suite, smoke, the bench lines, rocprofv3 stats + HBM counters + timeline
set -x
bash tools/final_round3.sh > gpurun_out/fin3.log 2>&1
tail -12 gpurun_out/fin3.log
bash tools/prof_round3.sh > gpurun_out/prof_r3.log 2>&1
tail -3 gpurun_out/prof_r3.log
bash tools/prof_edges.sh > gpurun_out/edges.log 2>&1

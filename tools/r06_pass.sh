#!/bin/sh
# one gpurun pass of round 6: the -m gpu suite, smoke(), the bench lines of every configuration (with the CPU legs), host-side phase timing
mkdir -p gpurun_out/r06
python -m pytest tests -m gpu -q > gpurun_out/r06/pytest_gpu.log 2>&1; tail -3 gpurun_out/r06/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06/smoke.log 2>&1; tail -1 gpurun_out/r06/smoke.log
python bench.py > gpurun_out/r06/bench_default.json 2> gpurun_out/r06/bench_default.err
python bench.py --qual bin > gpurun_out/r06/bench_bin.json 2> gpurun_out/r06/bench_bin.err
python bench.py --config bam > gpurun_out/r06/config_bam.json 2> gpurun_out/r06/config_bam.err
python bench.py --config bam --bam-binary > gpurun_out/r06/config_bam_records.json 2> gpurun_out/r06/config_bam_records.err
python bench.py --config vcf --steps 3 --warmup 1 --warm-steps 2 > gpurun_out/r06/config_vcf.json 2> gpurun_out/r06/config_vcf.err
python bench.py --stream-reads 8000000 --steps 3 --warmup 1 > gpurun_out/r06/bench_stream.json 2> gpurun_out/r06/bench_stream.err
GZ_ZIP_TIMING=1 python bench.py --no-cpu --steps 2 --warmup 1 --warm-steps 0 2>&1 | grep gz_zip | tail -4 > gpurun_out/r06/host_phases.txt
for f in default bin; do python -c "
import json
d=json.load(open('gpurun_out/r06/bench_$f.json')); print('$f', d['ms_per_step'], d['value'], d['bit_exact'], d['gpu_over_cpu'])"; done
for f in bam bam_records vcf; do python -c "
import json
d=json.load(open('gpurun_out/r06/config_$f.json')); print('$f', d['ms_per_step'], d['value'], d['bit_exact'], d.get('warm',{}).get('ms_per_step'), d['gpu_over_cpu'])"; done
python -c "
import json
d=json.load(open('gpurun_out/r06/bench_stream.json')); print('stream', d['ms_per_step'], d['value'], d['bit_exact'], d['gpu_over_cpu'])"
cat gpurun_out/r06/host_phases.txt

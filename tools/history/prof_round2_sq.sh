set -x
OUT=/root/repo/gpurun_out/prof3; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
GZ_NO_PIPELINE=1 GZ_ZIP_NO_OVERLAP=1 timeout 1200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES --output-format csv -d $OUT/sq -o r2 -- python /root/repo/bench.py --steps 1 --warmup 1 --no-cpu --pin-codecs > $OUT/bench_sq.json 2> $OUT/sq.err
ls -la $OUT/sq; tail -3 $OUT/sq.err

run() { python bench.py "$@" --no-cpu 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'])"; }
echo base; run --stream-reads 8000000 --steps 3 --warmup 1; run --stream-reads 8000000 --steps 3 --warmup 1; run --steps 6 --warmup 2; run --config bam --steps 4 --warmup 1
cp genozip_amd/libgenozip_amd.so /tmp/real.so; cp genozip_amd/variants/prio.so genozip_amd/libgenozip_amd.so
echo prio; run --stream-reads 8000000 --steps 3 --warmup 1; run --stream-reads 8000000 --steps 3 --warmup 1; run --steps 6 --warmup 2; run --config bam --steps 4 --warmup 1
cp /tmp/real.so genozip_amd/libgenozip_amd.so

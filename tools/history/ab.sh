# GPU: A/B of two builds of the library on one box (genozip_amd/libgz_head.so stands in for the library for the A runs)
cp genozip_amd/libgenozip_amd.so /tmp/lib_keep.so
for rep in 1 2; do
for which in head cur; do
  if [ $which = head ]; then cp genozip_amd/libgz_head.so genozip_amd/libgenozip_amd.so; else cp /tmp/lib_keep.so genozip_amd/libgenozip_amd.so; fi
  python tools/vcf_model_probe.py 3000 2>&1 | grep "^step" | tail -1
  for a in "--config bam" "--qual bin" "" "--stream-reads 8000000 --steps 3 --warmup 1"; do python bench.py $a --no-cpu --warm-steps 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1:], d['ms_per_step'])" $which $a; done
done; done
cp /tmp/lib_keep.so genozip_amd/libgenozip_amd.so
python -m pytest tests/test_gpu.py -x -q -k "golden or fastq_zip or vcf_zip or configs or sam_zip" 2>&1 | tail -2

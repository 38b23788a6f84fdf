# GPU: the eventful-batch thresholds of k_arith_model (GZ_MODEL_EVENTS_IN / _OUT) - variants built beside the library, each standing in
# for it for one run of the VCF probe and the BAM / FASTQ bench lines
cp genozip_amd/libgenozip_amd.so /tmp/lib_keep.so
for v in 24 40 64; do
  cp genozip_amd/libgz_ev_$v.so genozip_amd/libgenozip_amd.so
  echo "== EVENTS_IN $v"
  python tools/vcf_model_probe.py 3000 2>&1 | grep "phases\|^step" | tail -2
  python bench.py --config bam --no-cpu --warm-steps 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bam', d['ms_per_step'])"
  python bench.py --no-cpu --warm-steps 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fastq', d['ms_per_step'])"
  python bench.py --qual bin --no-cpu --warm-steps 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fastq bin', d['ms_per_step'])"
done
cp /tmp/lib_keep.so genozip_amd/libgenozip_amd.so
python bench.py --qual bin --no-cpu --warm-steps 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('baseline fastq bin', d['ms_per_step'])"

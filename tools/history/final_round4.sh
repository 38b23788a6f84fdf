# round 4, final pass on the GPU box (one gpurun call): the whole -m gpu suite, smoke, the bench lines that go into profiles/
set -x
OUT=gpurun_out/fin4; mkdir -p $OUT
python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log; tail -3 $OUT/pytest.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
python bench.py --config bam > $OUT/config_bam.json 2> $OUT/config_bam.err
python bench.py --config bam --qual div > $OUT/config_bam_div.json 2> $OUT/config_bam_div.err
python bench.py --config bam --bam-binary > $OUT/config_bam_records.json 2> $OUT/config_bam_records.err
python bench.py --qual bin > $OUT/bench_bin.json 2> $OUT/bench_bin.err
python bench.py --config vcf --steps 3 --warmup 1 --warm-steps 0 > $OUT/config_vcf.json 2> $OUT/config_vcf.err
python bench.py --stream-reads 8000000 --steps 3 --warmup 1 --no-cpu --warm-steps 0 > $OUT/bench_stream.json 2> $OUT/bench_stream.err
for f in bench_default bench_bin config_bam config_bam_div config_bam_records config_vcf bench_stream; do python - $OUT/$f.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1], d["ms_per_step"], d["value"], d["unit"], "bit_exact", d.get("bit_exact"), "cpu", (d.get("cpu_baseline") or {}).get("value"))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done

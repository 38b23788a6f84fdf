# after "QUAL coded ahead only when its streams are long": the zip tests on the GPU, then the configurations it touches
set -x
OUT=gpurun_out/r3c; mkdir -p $OUT
python -m pytest tests -m gpu -x -q -k "fastq_zip or sam_zip or vcf_zip or streamed or full_size_fastq" > $OUT/pytest_sub.log 2>&1; tail -4 $OUT/pytest_sub.log
python bench.py --steps 3 --warmup 1 --no-cpu --warm-steps 0 > $OUT/default.json 2> $OUT/default.err
python bench.py --qual bin --steps 3 --warmup 1 --no-cpu --warm-steps 0 > $OUT/bin.json 2> $OUT/bin.err
python bench.py --config bam --steps 3 --warmup 1 --no-cpu > $OUT/bam_text.json 2> $OUT/bam_text.err
python bench.py --config bam --bam-binary --steps 3 --warmup 1 --no-cpu > $OUT/bam_bin.json 2> $OUT/bam_bin.err
GZ_ZIP_EARLY_MIN=0 python bench.py --config bam --steps 3 --warmup 1 --no-cpu > $OUT/bam_text_early.json 2> $OUT/bam_text_early.err
GZ_ZIP_EARLY_MIN=0 python bench.py --qual bin --steps 3 --warmup 1 --no-cpu --warm-steps 0 > $OUT/bin_early.json 2> $OUT/bin_early.err
python - <<'PY'
import json
for f in ("default","bin","bin_early","bam_text","bam_text_early","bam_bin"):
    try:
        d=json.loads(open("gpurun_out/r3c/%s.json"%f).read().strip().splitlines()[-1])
        print(f, d["ms_per_step"], d["value"], d.get("bit_exact"))
    except Exception as e: print(f,"FAILED",e)
PY

set -x
OUT=gpurun_out/r3b; mkdir -p $OUT
python -m pytest tests -m gpu -x -q -k "bam_front or sam_zip or seg_columns or fastq_zip_driver or abi" > $OUT/pytest_sub.log 2>&1; tail -5 $OUT/pytest_sub.log
python bench.py --config bam --steps 3 --warmup 1 --no-cpu > $OUT/bam_text.json 2> $OUT/bam_text.err; tail -c 600 $OUT/bam_text.err
python bench.py --config bam --bam-binary --steps 3 --warmup 1 --no-cpu > $OUT/bam_bin.json 2> $OUT/bam_bin.err; tail -c 600 $OUT/bam_bin.err
python - <<'PY'
import json
for f in ("bam_text","bam_bin"):
    try:
        d=json.loads(open("gpurun_out/r3b/%s.json"%f).read().strip().splitlines()[-1])
        print(f, d["ms_per_step"], d["value"], d["text_mb_s"])
        print({k:v for k,v in d["roofline"]["kernel_ms_per_step_summed_over_concurrent_launches"].items()})
    except Exception as e: print(f,"FAILED",e)
PY

run() { echo "$@"; python bench.py "$@" --no-cpu 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'], d.get('bit_exact'))"; }
python -m pytest tests/test_gpu.py -x -q -k "codec or golden or arith or chain" 2>&1 | tail -2
run --qual bin --steps 6 --warmup 2
run --config bam --steps 4 --warmup 1
run --config bam --bam-binary --steps 4 --warmup 1
run --config vcf --steps 2 --warmup 1

run() { echo "$@"; python bench.py "$@" --no-cpu 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'], {k: v for k, v in d['roofline']['kernel_ms_per_step_summed_over_concurrent_launches'].items() if k in ('k_chain_expand','k_low_scatter','k_low_scan','k_ctx_scatter','k_arith_model','k_arith_chain','k_ctx_count','k_ctx_scan')})"; }
python -m pytest tests/test_gpu.py -x -q -k "codec or golden or arith or chain or wide" 2>&1 | tail -2
run --steps 6 --warmup 2
run --stream-reads 8000000 --steps 3 --warmup 1
run --stream-reads 8000000 --steps 3 --warmup 1
run --config bam --steps 4 --warmup 1

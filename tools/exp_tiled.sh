# A / B of k_arith_model_tiled (GZ_MODEL_TILED=0: the contexts' waves behind the sort through memory, as before) in one gpurun call
mkdir -p gpurun_out/tiled
P='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d["roofline"]["kernel_ms_per_step_summed_over_concurrent_launches"]; print(sys.argv[1:], d["ms_per_step"], d["value"], (d["roofline"].get("critical_path") or {}).get("ns_per_symbol"), {x: k.get(x) for x in ("k_arith_chain", "k_arith_model", "k_arith_model_tiled", "k_ctx_scatter", "k_chain_expand", "k_low_scatter")})'
for rep in ${REPS:-1}; do
for t in 1 0; do
  i=0
  for a in "" "--stream-reads 8000000 --steps 3 --warmup 1" "--qual bin" "--config bam" ; do
    i=$((i+1))
    GZ_MODEL_TILED=$t timeout 100 python bench.py $a --no-cpu --warm-steps 0 > gpurun_out/tiled/b_${t}_${i}_$rep.json 2> gpurun_out/tiled/b_${t}_${i}_$rep.err; echo "rc $?"
    python -c "$P" tiled=$t $a < gpurun_out/tiled/b_${t}_${i}_$rep.json
  done
done; done

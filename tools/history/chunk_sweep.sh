# GZ_ARITH_CHUNKS (position chunks of the model / chain pipeline) swept over the default step and the streamed form
OUT=gpurun_out/chunks; mkdir -p $OUT
for C in 16 32 64 128; do
  GZ_ARITH_CHUNKS=$C python bench.py --stream-reads 8000000 --steps 3 --warmup 1 --no-cpu --warm-steps 0 > $OUT/stream_$C.json 2> $OUT/stream_$C.err
done
for C in 16 24 32 48; do
  GZ_ARITH_CHUNKS=$C python bench.py --steps 5 --warmup 2 --no-cpu --warm-steps 0 > $OUT/default_$C.json 2> $OUT/default_$C.err
done
for C in 16 48; do
  GZ_ARITH_CHUNKS=$C python bench.py --config bam --steps 5 --warmup 2 --no-cpu --warm-steps 0 > $OUT/bam_$C.json 2> $OUT/bam_$C.err
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/chunks/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f, d["ms_per_step"], d["value"], "bit_exact", d.get("bit_exact"), (d["roofline"].get("critical_path") or {}).get("longest_launch_ms"))
    except Exception as e:
        print(f, "FAILED", e, open(f.replace(".json", ".err")).read()[-300:])
PY

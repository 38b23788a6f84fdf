for cfg in "--config bam" "--config bam --bam-binary" "--qual bin" "" "--config vcf --steps 2 --warmup 1"; do
  for np in 0 1; do
    tag=$(echo "$cfg" | tr -d ' -')_np$np
    if [ $np = 1 ]; then export GZ_ZIP_NO_PREDICTION=1; else unset GZ_ZIP_NO_PREDICTION; fi
    python bench.py $cfg --no-cpu > gpurun_out/r5_pred_$tag.json 2> gpurun_out/r5_pred_$tag.err
    python -c "
import json,sys
d=json.load(open('gpurun_out/r5_pred_$tag.json'))
print('$tag', d['ms_per_step'], d['value'], d['config'].get('codec_prediction','')[:30])"
  done
done

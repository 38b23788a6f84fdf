# engine clock while the default step runs: samples of pp_dpm_sclk / rocm-smi during bench.py
OUT=gpurun_out/clk; mkdir -p $OUT
ls /sys/class/drm/ > $OUT/drm.txt 2>&1
for c in /sys/class/drm/card*/device/pp_dpm_sclk; do echo $c; cat $c; done > $OUT/sclk_idle.txt 2>&1
( for i in $(seq 1 60); do for c in /sys/class/drm/card*/device/pp_dpm_sclk; do grep "\*" $c; done; sleep 0.1; done > $OUT/sclk_run.txt 2>&1 ) &
python bench.py --steps 40 --warmup 2 --no-cpu --warm-steps 0 > $OUT/bench.json 2> $OUT/bench.err
wait
rocm-smi --showclocks > $OUT/smi.txt 2>&1
sort $OUT/sclk_run.txt | uniq -c
cat $OUT/sclk_idle.txt | head -20

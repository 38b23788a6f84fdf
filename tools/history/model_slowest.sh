# which model wave is the long pole of the trial / finish phases (BAM from text, binned FASTQ): the -DGZ_MODEL_DEBUG build (genozip_amd/libgz_dbg.so) in the library's place
cp genozip_amd/libgenozip_amd.so /tmp/lib_keep.so; cp genozip_amd/libgz_dbg.so genozip_amd/libgenozip_amd.so
OUT=gpurun_out/slow; mkdir -p $OUT
GZ_DEBUG_PIPE=1 GZ_ZIP_TIMING=1 python bench.py --config bam --steps 1 --warmup 1 --no-cpu > $OUT/bam.json 2> $OUT/bam.err
GZ_DEBUG_PIPE=1 GZ_ZIP_TIMING=1 python bench.py --qual bin --steps 1 --warmup 1 --no-cpu --warm-steps 0 > $OUT/bin.json 2> $OUT/bin.err
cp /tmp/lib_keep.so genozip_amd/libgenozip_amd.so
for f in bam bin; do grep "\[model\]\|\[phases\]\|\[gz_zip\|\[pipe\] bg" $OUT/$f.err | tail -40 > $OUT/$f.txt; done

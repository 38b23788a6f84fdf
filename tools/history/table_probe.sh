#!/bin/bash
# GPU: k_rans_table's phases, timed inside the kernel (a -DGZ_TABLE_DEBUG build stands in for the library for this run only)
set -e
mkdir -p gpurun_out/tab
cp genozip_amd/libgenozip_amd.so /tmp/lib_keep.so
cp genozip_amd/libgenozip_amd_dbg.so genozip_amd/libgenozip_amd.so
python bench.py --steps 2 --warmup 1 --no-cpu --warm-steps 0 > gpurun_out/tab/bench.json 2> gpurun_out/tab/table.txt || true
cp /tmp/lib_keep.so genozip_amd/libgenozip_amd.so
grep "\[table\]" gpurun_out/tab/table.txt | head -40
tail -3 gpurun_out/tab/table.txt

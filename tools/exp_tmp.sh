cp genozip_amd/libgenozip_amd.so /tmp/real.so
bash tools/prof_timeline4.sh s_real --stream-reads 8000000 > /dev/null 2>&1
cp genozip_amd/variants/sortedrec.so genozip_amd/libgenozip_amd.so
bash tools/prof_timeline4.sh s_sorted --stream-reads 8000000 > /dev/null 2>&1
cp /tmp/real.so genozip_amd/libgenozip_amd.so
for t in s_real s_sorted; do echo == $t; head -c 300 gpurun_out/tl4_$t/bench.json; echo; tail -2 gpurun_out/tl4_$t/trace.err; grep -E "k_arith_model|k_arith_chain|k_ctx_scatter|k_chain_expand" gpurun_out/tl4_$t/kernel_stats.csv | cut -d, -f1-4 | cut -c1-40,100-; done

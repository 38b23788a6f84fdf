# SQ counters of the hot kernels (where do a kernel's wave cycles go: issuing, waiting for an instruction's operands, parked at s_waitcnt):
# separate --pmc passes with the kernels serialised (GZ_NO_PIPELINE: no persistent chain), per configuration.   usage: prof_sq.sh <tag> <bench args ...>
set -x
TAG=$1; shift
OUT=/root/repo/gpurun_out/sq_$TAG; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU"
P2="SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT"
P3="GRBM_GUI_ACTIVE SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_INSTS_BRANCH SQ_WAVES_EQ_64 SQ_LDS_IDX_ACTIVE"
i=1
for P in "$P1" "$P2" "$P3"; do
  GZ_NO_PIPELINE=1 GZ_ZIP_NO_OVERLAP=1 timeout 900 rocprofv3 --pmc $P --output-format csv -d $OUT/p$i -o sq -- python /root/repo/bench.py --steps 1 --warmup 1 --no-cpu --pin-codecs --warm-steps 0 "$@" > $OUT/bench_p$i.json 2> $OUT/p$i.err
  i=$((i+1))
done
python /root/repo/tools/summarize_sq.py $OUT > /root/repo/gpurun_out/sq_$TAG.txt 2>&1
find $OUT -name "*counter_collection.csv" -size +30M -delete; du -sh $OUT

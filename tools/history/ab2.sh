# GPU: A/B of builds of the library on one box: sh tools/ab2.sh "<variant names, 'cur' = the shipped library>" ["bench args" ...]
VARS=$1; shift
cp genozip_amd/libgenozip_amd.so /tmp/lib_keep.so
for rep in 1 2; do
for which in $VARS; do
  if [ $which = cur ]; then cp /tmp/lib_keep.so genozip_amd/libgenozip_amd.so; else cp genozip_amd/libgz_$which.so genozip_amd/libgenozip_amd.so; fi
  for a in "$@"; do python bench.py $a --no-cpu --warm-steps 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1:], d['ms_per_step'])" $which $a; done
done; done
cp /tmp/lib_keep.so genozip_amd/libgenozip_amd.so

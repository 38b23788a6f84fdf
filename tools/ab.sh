#!/bin/sh
# GPU box: A / B of library variants inside ONE call (same box, same clocks):   tools/ab.sh "<variants>" <reps> <bench args...>
# prints ms per step, value and the chain's ns per symbol for each; the product library is put back at the end
P='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d["ms_per_step"], d["value"], (d["roofline"].get("critical_path") or {}).get("ns_per_symbol"), d.get("bit_exact"))'
vars=$1; reps=$2; shift 2
cp genozip_amd/libgenozip_amd.so /tmp/product.so
for rep in $(seq $reps); do
  for v in $vars; do
    if [ $v = product ]; then cp /tmp/product.so genozip_amd/libgenozip_amd.so; else cp genozip_amd/variants/$v.so genozip_amd/libgenozip_amd.so; fi
    timeout 400 python bench.py "$@" 2>/dev/null | python -c "$P" $v
  done
done
cp /tmp/product.so genozip_amd/libgenozip_amd.so

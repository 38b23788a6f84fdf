# an experimental build of the library beside the real one: bash tools/build_variant.sh <name> <extra hipcc flags...>
# -> genozip_amd/variants/<name>.so (travels with gpurun; use it there with: cp genozip_amd/variants/<name>.so genozip_amd/libgenozip_amd.so)
N=$1; shift
mkdir -p /root/repo/genozip_amd/variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -Wno-unused-value "$@" -I /root/repo/genozip_amd/csrc -x hip /root/repo/genozip_amd/csrc/gz_host.cpp -o /root/repo/genozip_amd/variants/$N.so

# builds genozip_amd/libgz_<name>.so with extra compiler flags: sh tools/build_variant.sh <name> [-DFLAG ...]
N=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -Wno-unused-value "$@" -I genozip_amd/csrc -x hip genozip_amd/csrc/gz_host.cpp -o genozip_amd/libgz_$N.so

#!/bin/sh
# a build of the library with another generated chain loop / other -D switches:   tools/build_variant.sh <name> [<chain header>] [-D...]
# -> genozip_amd/variants/<name>.so (travels to the GPU box, not committed); tools/ab.sh copies them over the product one at a time
set -e
cd "$(dirname "$0")/.."
name=$1; shift
hdr=""
if [ -n "$1" ] && [ "${1#-}" = "$1" ]; then hdr="-DGZ_CHAIN_ASM_HDR=\"$1\""; shift; fi
mkdir -p genozip_amd/variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -w -I genozip_amd/csrc $hdr "$@" -x hip genozip_amd/csrc/gz_host.cpp -o genozip_amd/variants/$name.so
echo built genozip_amd/variants/$name.so

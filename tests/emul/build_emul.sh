#!/bin/sh
# TEST INFRASTRUCTURE: compiles the unmodified product sources against the CPU stand-in for <hip/hip_runtime.h>
set -e
HERE=$(cd "$(dirname "$0")" && pwd)
ROOT=$(cd "$HERE/../.." && pwd)
g++ -O2 -g -std=c++20 -fPIC -shared -ffp-contract=off -Wno-unused-function -I "$HERE" -I "$ROOT/genozip_amd/csrc" \
    -x c++ "$ROOT/genozip_amd/csrc/gz_host.cpp" -o "$HERE/libgenozip_amd_emul.so.tmp$$"
mv -f "$HERE/libgenozip_amd_emul.so.tmp$$" "$HERE/libgenozip_amd_emul.so"

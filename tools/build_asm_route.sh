#!/bin/sh
# The library built through assembly text, so that tools/align_asm.py can stand between the compiler and the assembler:
#   tools/build_asm_route.sh <out.so> [align|plain] [extra compiler flags]
# device code: hipcc -S -> (align_asm.py) -> llvm-mc / lld -> clang-offload-bundler; host code: hipcc --cuda-host-only with that bundle.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$1; MODE=${2:-align}; shift; [ $# -gt 0 ] && shift
T=$(mktemp -d); trap 'rm -rf $T' EXIT
LLVM=/opt/rocm/lib/llvm/bin
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-value -I $ROOT/genozip_amd/csrc $*"
/opt/rocm/bin/hipcc $FLAGS -x hip $ROOT/genozip_amd/csrc/gz_host.cpp --cuda-device-only -S -o $T/dev.s
if [ "$MODE" = align ]; then python3 $ROOT/tools/align_asm.py $T/dev.s $T/dev_al.s > $T/align.log; tail -1 $T/align.log; else cp $T/dev.s $T/dev_al.s; fi
$LLVM/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c $T/dev_al.s -o $T/dev.o
$LLVM/lld -flavor gnu -m elf64_amdgpu --no-undefined -shared -o $T/dev.co $T/dev.o
$LLVM/clang-offload-bundler -type=o -bundle-align=4096 -targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950 -input=/dev/null -input=$T/dev.co -output=$T/dev.hipfb
/opt/rocm/bin/hipcc $FLAGS -shared -x hip $ROOT/genozip_amd/csrc/gz_host.cpp --cuda-host-only -Xclang -fcuda-include-gpubinary -Xclang $T/dev.hipfb -o $OUT.tmp$$
mv -f $OUT.tmp$$ $OUT
[ -f $T/align.log ] && cp $T/align.log ${OUT%.so}.align.log || true
echo built $OUT

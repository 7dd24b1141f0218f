#!/bin/bash
# Build a variant of the library whose KERNEL file is compiled with extra flags (instrumented or experimental builds for same-box A/B runs):
#   bash scripts/build_variant.sh <tag> <extra hipcc flags...>   ->  ab/libsluamd_<tag>.so   (ab/ is git-ignored and travels with gpurun)
# The host objects are the in-tree ones (run `make -C superlu_dist_amd/csrc` first).
set -e
tag=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
src=$root/superlu_dist_amd/csrc
mkdir -p $root/ab/$tag
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$root/include -I$src -munsafe-fp-atomics -Wno-unused-value -Wno-pass-failed "$@" \
    -c $src/sluamd_kernels.hip -o $root/ab/$tag/sluamd_kernels.o
objs=$(ls $root/superlu_dist_amd/build/*.o | grep -v sluamd_kernels.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared $root/ab/$tag/sluamd_kernels.o $objs -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib \
    -Wl,--version-script=$src/sluamd.map -o $root/ab/libsluamd_$tag.so
echo "built ab/libsluamd_$tag.so"

#!/bin/bash
# Builds an A/B variant of liblig_hip.so with extra compiler flags into tools/ab/liblig_hip_<name>.so (git-ignored; it
# travels to the GPU box with the snapshot).  Use with LIG_HIP_LIB=tools/ab/liblig_hip_<name>.so python bench.py ...
#   tools/build_variant.sh dpp "-DLIG_TILE_DPP"
set -e
name=$1; flags=$2
root=$(cd "$(dirname "$0")/.." && pwd)
tmp=/tmp/lig_variant_$name
rm -rf $tmp && mkdir -p $tmp/ligero-prover_amd $tmp/include $root/tools/ab
cp -r $root/ligero-prover_amd/csrc $tmp/ligero-prover_amd/ && cp $root/include/*.h $root/include/*.hpp $tmp/include/
rm -f $tmp/ligero-prover_amd/csrc/*.o
make -C $tmp/ligero-prover_amd/csrc -j8 CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function $flags" > /dev/null
cp $tmp/ligero-prover_amd/liblig_hip.so $root/tools/ab/liblig_hip_$name.so
echo built tools/ab/liblig_hip_$name.so

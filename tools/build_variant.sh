#!/bin/bash
# usage: tools/build_variant.sh TAG "-DFLAG=..."  -> retinal_oct_image_segmentation_via_deep_learning_amd/liboct_hip_TAG.so
# An A/B build of the library with extra compiler flags (same-box comparisons: OCT_HIP_LIB=<that file> python bench.py ...).
set -e
TAG=$1; EXTRA=$2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SRC=$ROOT/retinal_oct_image_segmentation_via_deep_learning_amd/csrc
OBJ=/tmp/oct_variant_$TAG
mkdir -p $OBJ
for f in $SRC/*.hip $SRC/runtime.cpp; do
  b=$(basename $f)
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-result $EXTRA -x hip -c $f -o $OBJ/${b%.*}.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/retinal_oct_image_segmentation_via_deep_learning_amd/liboct_hip_$TAG.so $OBJ/*.o
echo built liboct_hip_$TAG.so

#!/bin/bash
# usage: tools/spills.sh FILE.hip ["-Dflags"]  -- per kernel: VGPRs and scratch bytes (spills) from the device assembly
SRC=$(dirname "$0")/../retinal_oct_image_segmentation_via_deep_learning_amd/csrc
F=${1:-igemm2.hip}; EXTRA=$2
OUT=/tmp/spills_${F%.*}_$$.s
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -Wno-unused-result $EXTRA --cuda-device-only -S -o $OUT $SRC/$F || exit 1
python3 - $OUT <<'PY'
import re, sys
t = open(sys.argv[1]).read()
n = bad = 0
for m in re.finditer(r"\.amdhsa_kernel (\S+).*?; NumVgprs: (\d+).*?; ScratchSize: (\d+)", t, re.S):
    n += 1
    if int(m.group(3)):
        bad += 1
        print("scratch", m.group(3), "vgprs", m.group(2), m.group(1))
print(n, "kernels,", bad, "with scratch")
PY
echo $OUT

#!/bin/bash
# Cout<=16 kernel (16x16x4 MFMA) tile sweep on conv1aa/conv1b geometry (batch 16 -> 32 images at 224x512)
cd "$(dirname "$0")/.."
echo "== 32x32x2 kernel (PWC_CONV16=0)"
PWC_CONV16=0 PWC_BENCH_GEOM=32,224,512 python tools/bench_conv.py p1:16:16:1 2>&1 | grep -v amdgpu.ids
for t in 1,4 1,16 2,4 2,8 2,16 4,4 4,16; do
  echo "== 16x16x4 kernel nt,ck = $t"
  PWC_CONV16_TILE=$t PWC_BENCH_GEOM=32,224,512 python tools/bench_conv.py p1:16:16:1 2>&1 | grep -v amdgpu.ids
done
echo "== batch 1 (2 images)"
for t in 1,4 1,16 2,16; do
  echo "== 16x16x4 kernel nt,ck = $t"
  PWC_CONV16_TILE=$t PWC_BENCH_GEOM=2,224,512 python tools/bench_conv.py p1:16:16:1 2>&1 | grep -v amdgpu.ids
done

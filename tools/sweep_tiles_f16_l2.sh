#!/bin/bash
# fp16 conv tile sweep (PWC_CONV16F_MT / _NT / _RING) for the level-2 layers at batch 16 (run on the GPU box)
cd "$(dirname "$0")/.."
export PWC_BENCH_F16_ONLY=1
L="c2_0:117:128:1 c2_1:245:128:1 c2_2:373:96:1 c2_3:469:64:1 c2_4:533:32:1 dc1:565:128:1 dc2:128:128:2 dc3:128:128:4 dc4:128:96:8 dc5:96:64:16 dc6:64:32:1"
echo "== tile auto"; python tools/bench_conv_f16.py $L 2>&1 | grep -v amdgpu.ids
for mt in 1 2 3 4; do for nt in 2 4; do for r in 2 3; do
  echo "== tile $mt,$nt,$r"
  PWC_CONV16F_MT=$mt PWC_CONV16F_NT=$nt PWC_CONV16F_RING=$r python tools/bench_conv_f16.py $L 2>&1 | grep -v amdgpu.ids
done; done; done

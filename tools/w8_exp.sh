#!/bin/bash
# MFMA shape experiment on the 8-wave fp16 conv kernel (auto tile rule)
export PWC_BENCH_F16_ONLY=1
for sh in 0 1 0 1; do
  echo "=== PWC_CONV16F_SHAPE=$sh"
  export PWC_CONV16F_SHAPE=$sh
  python tools/bench_conv_f16.py dc_conv1 conv2_0 conv2_2 conv2_3 conv2_4 c2_1:245:128:1 dc4:128:96:8 dc6:64:32:1
  PWC_BENCH_GEOM=16,56,128 python tools/bench_conv_f16.py c3_0:149:128:1 c3_1:277:128:1 c3_2:405:96:1 c3_3:501:64:1
done

#!/bin/bash
# 8-wave fp16 conv kernel vs the 5-wave one on the level-2 / level-3 / context layers
export PWC_BENCH_F16_ONLY=1
for w8 in 0 1 2 4; do
  echo "=== PWC_CONV16F_W8=$w8"
  export PWC_CONV16F_W8=$w8
  python tools/bench_conv_f16.py dc_conv1 conv2_0 conv2_2 conv2_3 conv2_4 dc_conv2 dc_conv3 c2_1:245:128:1 dc4:128:96:8 dc6:64:32:1
  PWC_BENCH_GEOM=16,56,128 python tools/bench_conv_f16.py c3_0:149:128:1 c3_1:277:128:1 c3_2:405:96:1 c3_3:501:64:1 c3_4:565:32:1
  PWC_BENCH_GEOM=32,112,256 python tools/bench_conv_f16.py c2aa:32:32:1
  PWC_BENCH_GEOM=32,224,512 python tools/bench_conv_f16.py c1aa:16:16:1
done

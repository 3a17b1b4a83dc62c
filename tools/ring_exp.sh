#!/bin/bash
# ring-depth experiment on small fp16 layers
export PWC_BENCH_F16_ONLY=1
for ring in 0 4 5; do
  echo "=== PWC_CONV16F_RING=$ring"
  export PWC_CONV16F_RING=$ring
  [ "$ring" = "0" ] && unset PWC_CONV16F_RING
  PWC_BENCH_GEOM=16,7,16 python tools/bench_conv_f16.py c6_0:81:128:1 c6_2:337:96:1 c6_4:529:32:1
  PWC_BENCH_GEOM=16,14,32 python tools/bench_conv_f16.py c5_0:213:128:1 c5_4:661:32:1
  PWC_BENCH_GEOM=16,28,64 python tools/bench_conv_f16.py c4_0:181:128:1 c4_2:437:96:1 c4_4:629:32:1
  PWC_BENCH_GEOM=16,56,128 python tools/bench_conv_f16.py c3_0:149:128:1 c3_1:277:128:1 c3_4:597:32:1
  PWC_BENCH_GEOM=32,224,512 python tools/bench_conv_f16.py c1aa:16:16:1
  PWC_BENCH_GEOM=32,112,256 python tools/bench_conv_f16.py c2aa:32:32:1
  PWC_BENCH_GEOM=32,56,128 python tools/bench_conv_f16.py c3aa:64:64:1
  PWC_BENCH_GEOM=32,28,64 python tools/bench_conv_f16.py c4aa:96:96:1
done

#!/bin/bash
# split-K sweep: level-6/5/4 dense-block layers at batch 1 and 16 (run on the GPU box)
cd "$(dirname "$0")/.."
for geom in "1,7,16" "1,14,32" "1,28,64" "1,56,128" "16,7,16" "16,14,32" "16,28,64"; do
  for k in 0 -1 2 4 8 16 32; do
    if [ "$k" = "-1" ]; then unset PWC_CONV_SPLIT; else export PWC_CONV_SPLIT=$k; fi
    echo "== geom $geom split ${PWC_CONV_SPLIT:-auto}"
    PWC_BENCH_GEOM=$geom PWC_BENCH_WS=1 python tools/bench_conv.py c0:81:128:1 c2:337:96:1 c4:497:32:1 hd:529:2:1 2>&1 | grep -v amdgpu.ids
  done
done

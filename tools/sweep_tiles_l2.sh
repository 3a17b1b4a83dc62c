#!/bin/bash
# conv tile sweep (PWC_CONV_TILE=mt,nt,two) for the level-2 layers at batch 16 (run on the GPU box)
cd "$(dirname "$0")/.."
for tile in auto 1,1,0 1,1,1 2,1,0 2,1,1 3,1,0 3,1,1 4,1,0 4,1,1 1,2,0 1,2,1 2,2,0 2,2,1 3,2,1 4,2,1 1,4,0 1,4,1 2,4,0 2,4,1; do
  if [ "$tile" = "auto" ]; then unset PWC_CONV_TILE; else export PWC_CONV_TILE=$tile; fi
  echo "== tile $tile"
  python tools/bench_conv.py c2_0:117:128:1 c2_1:245:128:1 c2_2:373:96:1 c2_3:469:64:1 c2_4:533:32:1 dc2:128:128:2 dc3:128:128:4 dc4:128:96:8 dc5:96:64:16 dc6:64:32:1 2>&1 | grep -v amdgpu.ids
done

#!/bin/bash
# conv tile sweep (PWC_CONV_TILE=mt,nt,two) for the level-3 / level-4 dense blocks and the pyramid layers at batch 16
# (run on the GPU box; forced tiles that do not exist for a layer fall back to the model's choice)
cd "$(dirname "$0")/.."
run() {  # geom, cases...
  local geom=$1; shift
  for tile in auto 1,1,0 1,1,1 2,1,0 2,1,1 3,1,1 4,1,1 1,2,0 1,2,1 2,2,0 2,2,1 4,2,1 1,4,0 1,4,1 2,4,0 2,4,1 4,4,0; do
    if [ "$tile" = "auto" ]; then unset PWC_CONV_TILE; else export PWC_CONV_TILE=$tile; fi
    echo "== geom $geom tile $tile"
    PWC_BENCH_GEOM=$geom python tools/bench_conv.py "$@" 2>&1 | grep -v amdgpu.ids
  done
}
run 16,56,128 c3_0:149:128:1 c3_1:277:128:1 c3_2:405:96:1 c3_3:501:64:1 c3_4:565:32:1
run 16,28,64 c4_0:181:128:1 c4_1:309:128:1 c4_2:437:96:1
run 32,224,512 p1:16:16:1
run 32,112,256 p2:32:32:1
run 32,56,128 p3:64:64:1
run 32,28,64 p4:96:96:1

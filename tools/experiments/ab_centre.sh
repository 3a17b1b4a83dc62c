#!/bin/bash
# A/B of the fused warp+correlation kernel's window-origin code: variant libraries (tools/variant_build.sh) against the in-tree one,
# interleaved, on the forward's own operands (tools/bench_corr_pipe.py time plan).  usage: ab_centre.sh <variant name>...
D=$PWD/opticalflow_amd/csrc/build/var
O=gpurun_out/s3_ab_centre3.txt
: > $O
echo "== bit-equality check (in-tree lib)" >> $O
timeout -k 10 200 python tools/bench_corr_pipe.py check >> $O 2>&1 || { echo "CHECK FAILED" >> $O; cat $O; exit 1; }
for i in 1 2 3; do
  for v in "$@" intree; do
    echo "== round $i $v" >> $O
    if [ $v = intree ]; then L=""; else L=$D/libpwc_$v.so; fi
    PWC_HIP_LIB=$L PWC_BENCH_LEVELS=2,3 timeout -k 10 120 python tools/bench_corr_pipe.py time plan 2>&1 | grep " new" >> $O
  done
done
cat $O

#!/bin/bash
# timing-only experiments on the 8-wave Winograd kernel: rebuilds the library with -DPWC_WINO_EXP=<mask> (results invalid),
# 1 no U DMA, 2 no raw DMA, 4 no transform, 8 no barrier / vmcnt wait.  Restores the normal build at the end.
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
cd "$ROOT/opticalflow_amd/csrc"
for e in ${@:-8 7 15}; do
  rm -f build/pwc_conv_wino.o; make EXTRA=-DPWC_WINO_EXP=$e > /dev/null 2>&1
  echo "== PWC_WINO_EXP=$e"; python3 "$ROOT/tools/bench_wino.py" time 2>&1 | grep -E "dc_conv1|conv2_4|conv2_0"
done
rm -f build/pwc_conv_wino.o; make > /dev/null 2>&1

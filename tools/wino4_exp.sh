#!/bin/bash
# timing-only experiments on the F(4x4,3x3) kernel: rebuilds the library with the -D flags in PWC_EXPS (one build per word) and times
# four layers.  -DPWC_W4_EXP=<mask> (results invalid): 1 = every U (filter) fetch reads chunk 0 (always cache-resident), 2 = every raw
# (input) fetch reads chunk 0, 4 = no LDS-DMA in the loop, 8 = no input transforms, 16 = no barrier / wait, 32 = no patch-row reads.
# -DPWC_W4_BLOCK=0|1: transform schedule (spread: four operations per MFMA / blocks of twelve).  Restores the normal build at the end.
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
cd "$ROOT/opticalflow_amd/csrc"
for e in ${PWC_EXPS:--DPWC_W4_EXP=0 -DPWC_W4_EXP=3 -DPWC_W4_EXP=7 -DPWC_W4_EXP=19}; do
  rm -f build/pwc_conv_wino4.o; make EXTRA="$e" > /dev/null 2>&1
  echo "== $e"; python3 "$ROOT/tools/bench_wino4.py" layers 2>&1 | grep -E "dc_conv1|conv2_1|conv2_4|conv2_3|conv2_2" | cut -c1-120
done
rm -f build/pwc_conv_wino4.o; make > /dev/null 2>&1

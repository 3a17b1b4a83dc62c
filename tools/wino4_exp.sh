#!/bin/bash
# timing-only experiments on the F(4x4,3x3) kernel: builds VARIANT libraries (tools/variant_build.sh -> csrc/build/var/libpwc_w4exp<n>.so,
# selected with PWC_HIP_LIB) with the -D flags in PWC_EXPS, one per word, and times the large layers with each.  The shipped
# libpwc_hip.so is never touched (the earlier form rebuilt it in place and an interrupted run left a wrong-result library as
# the default one).  -DPWC_W4_EXP=<mask> (results invalid): 1 = every U (filter) fetch reads chunk 0 (always cache-resident), 2 = every
# raw (input) fetch reads chunk 0, 4 = no LDS-DMA in the loop, 8 = no input transforms, 16 = no barrier / wait, 32 = no patch-row reads.
# -DPWC_W4_BLOCK=0|1: transform schedule (spread: four operations per MFMA / blocks of twelve).
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
i=0
for e in ${PWC_EXPS:--DPWC_W4_EXP=0 -DPWC_W4_EXP=3 -DPWC_W4_EXP=7 -DPWC_W4_EXP=19}; do
  i=$((i + 1))
  lib=$("$ROOT/tools/variant_build.sh" "w4exp$i" pwc_conv_wino4.hip "$e" | tail -1)
  echo "== $e"
  PWC_HIP_LIB="$ROOT/opticalflow_amd/csrc/$lib" python3 "$ROOT/tools/bench_wino4.py" layers 2>&1 | grep -E "dc_conv1|conv2_1|conv2_4|conv2_3|conv2_2" | cut -c1-120
done

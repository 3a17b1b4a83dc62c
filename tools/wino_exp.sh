#!/bin/bash
# timing-only experiments on the Winograd kernel (PWC_WINO_EXP bits: 1 no U DMA, 2 no raw DMA, 4 no transform, 8 no barrier, 16 no vmcnt wait): results invalid
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
for e in 0 1 2 4 8 16 3 7 31; do echo "== PWC_WINO_EXP=$e"; PWC_WINO_EXP=$e python3 "$ROOT/tools/bench_wino.py" time 2>&1 | grep -E "dc_conv1|conv2_4|conv2_0"; done

# fused window kernel variants (tools/variant_build.sh <name> pwc_corr_pipe.hip "<flags>"): level 2, the forward's own flow; each variant
# twice, interleaved with the shipped build, on ONE box (boxes and runs differ by +-2 us)
V=$PWD/opticalflow_amd/csrc/build/var
run() { PWC_BENCH_LEVELS=2 PWC_HIP_LIB=$1 timeout -k 10 100 python tools/bench_corr_pipe.py time plan 2>/dev/null | grep new | sed 's/.*fused/fused/'; }
for rep in 1 2 3; do
  echo "== shipped $(run $PWD/opticalflow_amd/libpwc_hip.so)"
  for v in ${PWC_VARIANTS}; do echo "== $v $(run $V/libpwc_$v.so)"; done
done

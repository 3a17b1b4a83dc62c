# fused window kernel variants (tools/variant_build.sh <name> pwc_corr_pipe.hip "<flags>"): level 2, the forward's own flow
V=$PWD/opticalflow_amd/csrc/build/var
echo "== shipped $(PWC_BENCH_LEVELS=2 timeout -k 10 100 python tools/bench_corr_pipe.py time plan 2>/dev/null | grep new)"
for v in ${PWC_VARIANTS}; do echo "== $v $(PWC_BENCH_LEVELS=2 PWC_HIP_LIB=$V/libpwc_$v.so timeout -k 10 100 python tools/bench_corr_pipe.py time plan 2>/dev/null | grep new)"; done

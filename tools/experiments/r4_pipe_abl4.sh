V=$PWD/opticalflow_amd/csrc/build/var
for v in ${PWC_VARIANTS}; do echo "== $v roll=$PWC_CORR_ROLL $(PWC_BENCH_LEVELS=2 PWC_HIP_LIB=$V/libpwc_$v.so timeout -k 10 100 python tools/bench_corr_pipe.py time 2>/dev/null | grep new)"; done

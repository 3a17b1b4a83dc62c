V=$PWD/opticalflow_amd/csrc/build/var
for v in ${PWC_VARIANTS:-pe17 pe33 pe49 pe16 pe32}; do echo "== $v $(PWC_BENCH_LEVELS=2 PWC_HIP_LIB=$V/libpwc_$v.so timeout -k 10 100 python tools/bench_corr_pipe.py time 2>/dev/null | grep new)"; done

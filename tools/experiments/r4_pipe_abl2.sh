# bits: 1 no fma, 2 no stores (out-of-range offsets), 4 no DMA after prologue, 8 drainer idle, 16 no stage writes, 32 reads only, 64 fmas only
V=$PWD/opticalflow_amd/csrc/build/var
for v in ${PWC_VARIANTS:-pe46 pe78 pe14}; do echo "== $v $(PWC_BENCH_LEVELS=2 PWC_HIP_LIB=$V/libpwc_$v.so timeout -k 10 100 python tools/bench_corr_pipe.py time 2>/dev/null | grep new)"; done

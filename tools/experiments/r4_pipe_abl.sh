# corr81_pipe_kernel: check, timing, ablations (timing only; -DPWC_PIPE_EXP bits: 1 no fma, 2 stores out of range, 4 no LDS-DMA after the prologue, 8 drainer idle)
V=$PWD/opticalflow_amd/csrc/build/var
timeout -k 10 200 python tools/bench_corr_pipe.py all
for v in ${PWC_VARIANTS:-pe1 pe2 pe8 pe12}; do echo "== $v $(PWC_BENCH_LEVELS=2 PWC_HIP_LIB=$V/libpwc_$v.so timeout -k 10 100 python tools/bench_corr_pipe.py time 2>/dev/null | grep new)"; done

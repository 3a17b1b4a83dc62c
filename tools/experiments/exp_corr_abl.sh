# corr81_dma_kernel ablations (timing only; -DPWC_CORR_EXP bits: 1 no fma, 2 cache-resident fetches, 4 no LDS-DMA in the loop, 8 no stores)
V=$PWD/opticalflow_amd/csrc/build/var
echo "base: $(python tools/bench_corr.py 2>/dev/null | grep 'corr ')"
for v in ${PWC_VARIANTS:-ce1 ce2 ce3 ce4 ce8 ce9 ce12}; do echo "$v: $(PWC_HIP_LIB=$V/libpwc_$v.so python tools/bench_corr.py 2>/dev/null | grep 'corr ')"; done
echo "fused base: $(PWC_BENCH_LEVELS=2 python tools/bench_warpcorr.py 2>/dev/null | cut -c1-60)"
for v in ${PWC_VARIANTS:-ce1 ce2 ce3 ce4 ce8 ce9 ce12}; do echo "fused $v: $(PWC_HIP_LIB=$V/libpwc_$v.so PWC_BENCH_LEVELS=2 python tools/bench_warpcorr.py 2>/dev/null | cut -c1-60)"; done

# round 4 baseline: the shipped correlation kernels and their ablations under ROTATING (cold) operand sets
V=$PWD/opticalflow_amd/csrc/build/var
echo "base: $(PWC_BENCH_LEVELS=2 python tools/bench_warpcorr.py 2>/dev/null)"
for v in ce1 ce8 ce9; do echo "$v: $(PWC_HIP_LIB=$V/libpwc_$v.so PWC_BENCH_LEVELS=2 python tools/bench_warpcorr.py 2>/dev/null)"; done
python tools/bench_corr.py 2>/dev/null

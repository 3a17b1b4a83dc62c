V=$PWD/opticalflow_amd/csrc/build/var
for rep in 1 2; do
 echo "== new";  python tools/bench_level_corr_f16.py 16 2>/dev/null | grep level | sed 's/| one kernel.*//'
 echo "== base"; PWC_HIP_LIB=$V/libpwc_f16base.so python tools/bench_level_corr_f16.py 16 2>/dev/null | grep level | sed 's/| one kernel.*//'
done

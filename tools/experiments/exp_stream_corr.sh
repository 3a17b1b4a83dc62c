V=$PWD/opticalflow_amd/csrc/build/var
echo "== stream3x3 predict_flow2 (565->2 @112x256 B16)"
echo base; python tools/bench_conv.py flow2:565:2:1
for v in s1 s2 s3; do echo $v; PWC_HIP_LIB=$V/libpwc_$v.so python tools/bench_conv.py flow2:565:2:1; done
echo "== plain corr"
echo base; python tools/bench_corr.py | grep corr
for v in cr4 ccu1 ccu3; do echo $v; PWC_HIP_LIB=$V/libpwc_$v.so python tools/bench_corr.py | grep corr; done
echo "== fused"
echo base; PWC_BENCH_LEVELS=2 python tools/bench_warpcorr.py
for v in wcseq cr4 ccu1; do echo $v; PWC_HIP_LIB=$V/libpwc_$v.so PWC_BENCH_LEVELS=2 python tools/bench_warpcorr.py; done

# stream3x3 (predict_flow2 565->2 @112x256 B16): tile shape (PWC_STREAM_CFG = TH,KS) x ring depth (variant builds, tools/variant_build.sh)
V=$PWD/opticalflow_amd/csrc/build/var
for cfg in 81 42 41 44; do
  echo "cfg $cfg ring 3: $(PWC_STREAM_CFG=$cfg python tools/bench_conv.py flow2:565:2:1 2>/dev/null)"
  for r in r4 r5; do
    echo "cfg $cfg ring $r: $(PWC_STREAM_CFG=$cfg PWC_HIP_LIB=$V/libpwc_$r.so python tools/bench_conv.py flow2:565:2:1 2>&1 | tail -1)"
  done
done

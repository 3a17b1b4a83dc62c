# stream3x3 ablations (timing only): s1 = consumers skip the arithmetic, s2 = every fetch reads chunk 0 (cache-resident), s3 = both
V=$PWD/opticalflow_amd/csrc/build/var
echo "base: $(python tools/bench_conv.py flow2:565:2:1 2>/dev/null)"
for v in ${PWC_VARIANTS:-s1 s2 s3}; do echo "$v: $(PWC_HIP_LIB=$V/libpwc_$v.so python tools/bench_conv.py flow2:565:2:1 2>&1 | tail -1)"; done

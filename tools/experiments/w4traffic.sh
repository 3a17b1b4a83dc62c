cd /tmp && export TMPDIR=/tmp
R=/root/repo
python3 $R/tools/bench_wino4.py layers 2>&1 | grep -E "dc_conv1|conv2_1|conv2_0|conv3_1|sum" | cut -c1-130
for c in FETCH_SIZE; do
  rocprofv3 --kernel-trace --pmc $c -d $R/gpurun_out/q -o p --output-format csv -- python3 $R/tools/bench_wino4.py pmc > /dev/null 2>&1
  python3 $R/tools/pmc_avg.py $R/gpurun_out/q conv3x3_wino4 $c; rm -rf $R/gpurun_out/q
done
python3 -m pytest $R/tests/test_gpu_parity.py -m gpu -q -k "winograd4" 2>&1 | tail -2

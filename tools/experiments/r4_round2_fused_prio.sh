# round-2 fused warp+correlation kernel (levels 3-5 of the forward): producer waves with / without issue priority, interleaved, one box
V=$PWD/opticalflow_amd/csrc/build/var
run() { PWC_HIP_LIB=$1 timeout -k 10 100 python tools/bench_warpcorr.py 2>/dev/null | grep level | sed -E 's/level ([0-9]).*fused +([0-9.]+) us.*/L\1 \2/' | tr '\n' ' '; }
for rep in 1 2 3; do
  echo "== prio2: $(run $PWD/opticalflow_amd/libpwc_hip.so)"
  echo "== prio0: $(run $V/libpwc_cp0.so)"
done

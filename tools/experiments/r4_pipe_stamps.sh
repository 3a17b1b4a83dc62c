V=$PWD/opticalflow_amd/csrc/build/var
echo "== full (stamped)"; PWC_HIP_LIB=$V/libpwc_pst.so timeout -k 10 100 python tools/experiments/pipe_stamps.py 2>/dev/null
echo "== arithmetic only: no DMA after prologue, no stores (stamped)"; PWC_HIP_LIB=$V/libpwc_pst1.so timeout -k 10 100 python tools/experiments/pipe_stamps.py 2>/dev/null

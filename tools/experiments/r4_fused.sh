timeout -k 10 300 python tools/bench_corr_pipe.py check noise
PWC_BENCH_LEVELS=2 timeout -k 10 300 python tools/bench_corr_pipe.py time plan

cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -x -q -k "pwconv or adapters or replk" > gpurun_out/t_pw.log 2>&1 || exit 1
timeout -k 10 120 python tools/bench_pw3.py > gpurun_out/pw3_new.log 2>&1 || exit 1
timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no_cpu_baseline > gpurun_out/bench_b.log 2>&1

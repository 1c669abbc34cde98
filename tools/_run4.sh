cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -x -q -k "upsample2x or conv2d_nhwc" > gpurun_out/t_up.log 2>&1 || exit 1
timeout -k 10 400 python -m pytest tests/test_e2e_gpu.py -x -q > gpurun_out/t_e2e.log 2>&1 || exit 1
timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no_cpu_baseline > gpurun_out/bench_c.log 2>&1

cd $GRAFT_REPO_ROOT
timeout -k 10 500 python -m pytest tests/test_kernels_gpu.py -x -q -k "bn_channel or fused_bn" > gpurun_out/t_bn.log 2>&1 && \
timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no_cpu_baseline > gpurun_out/bench_a.log 2>&1

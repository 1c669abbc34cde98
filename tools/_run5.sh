cd $GRAFT_REPO_ROOT
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -DDW_PROF -shared -o ppea-depth_amd/libppea_dwprof.so ppea-depth_amd/csrc/dwconv_mfma.hip && \
timeout -k 10 200 python tools/dwconv_phases.py > gpurun_out/dwphases.log 2>&1

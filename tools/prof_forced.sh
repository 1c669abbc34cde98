#!/bin/bash
# Kernel trace + stats of the forced-collective single-rank bench (every SyncBN / gradient collective goes through
# ProcessGroupNCCL with world = 1).  GPU box only:  gpurun -- 'bash tools/prof_forced.sh r03'
set -o pipefail
R=${1:-r03}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/prof_fc
mkdir -p $OUT
export PPEA_FORCE_COLLECTIVES=1
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_fc -- python3 bench.py --steps 20 --warmup 3 --no_cpu_baseline \
    > $OUT/${R}_forced_collectives_under_rocprof.json 2> $OUT/rocprof.err || exit 1
KT=$(find /tmp/prof_fc -name "*kernel_trace.csv" | head -1)
ST=$(find /tmp/prof_fc -name "*kernel_stats.csv" | head -1)
cp "$ST" $OUT/${R}_forced_collectives_kernel_stats.csv
python3 tools/step_profile.py "$KT" 20 70 > $OUT/${R}_forced_collectives_step_profile.txt || exit 1
echo done

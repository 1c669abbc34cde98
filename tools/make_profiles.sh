#!/bin/bash
# Regenerates the round's evidence under gpurun_out/prof_final (copy what is to be judged into profiles/).
# GPU box only:  gpurun --timeout 1200 -- 'bash tools/make_profiles.sh r02'
set -o pipefail
R=${1:-r02}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/prof_final
mkdir -p $OUT
if [ -z "$SKIP_BENCH" ]; then
rm -f $OUT/*
# 1. the default bench line (no profiler)
python3 bench.py > $OUT/${R}_bench_n1.json 2> $OUT/bench.err || exit 1
# 2. kernel trace + stats of the same command (20 timed steps), per-step breakdown
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_kt -- python3 bench.py --steps 20 --warmup 3 --no_cpu_baseline \
    > $OUT/${R}_bench_n1_under_rocprof.json 2> $OUT/rocprof.err || exit 1
KT=$(find /tmp/prof_kt -name "*kernel_trace.csv" | head -1)
ST=$(find /tmp/prof_kt -name "*kernel_stats.csv" | head -1)
cp "$ST" $OUT/${R}_bench_n1_kernel_stats.csv
python3 tools/step_profile.py "$KT" 20 70 > $OUT/${R}_bench_n1_step_profile.txt || exit 1
fi
# 3. hardware counters: ONE counter per pass (MI355X_MICROARCH.md: separate --pmc passes), never combined with tracing
i=0; dirs=""
for pmc in FETCH_SIZE WRITE_SIZE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES \
           SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS; do
    i=$((i + 1))
    timeout -k 10 240 rocprofv3 --pmc $pmc --output-format csv -d /tmp/prof_pmc$i -- python3 tools/pmc_target.py > /dev/null 2>> $OUT/pmc.err \
        || { echo "pmc pass $pmc failed" | tee -a $OUT/pmc_failed.txt; continue; }
    dirs="$dirs /tmp/prof_pmc$i"
    echo "pmc pass $pmc done"
done
python3 tools/pmc_summary.py $OUT/${R}_pmc_kernels.csv $dirs || exit 1
# 4. per-shape tables
python3 tools/bench_pw4.py --lib > $OUT/${R}_pwconv_shapes.txt 2>/dev/null || exit 1
echo "pwconv table done"
python3 tools/bench_conv.py > $OUT/${R}_conv_layers_vs_library.txt 2>/dev/null || exit 1
echo "conv table done"
# 5. wall-time attribution by ablation (tools/null_ablation.py)
for t in none pwconv pwgrad bn_ nhwc_bn conv_ dwconv_lk tapsum nhwc_up2cat; do
    PPEA_NULL=$t timeout -k 10 150 python3 tools/null_ablation.py > /tmp/null_$t.log 2> /tmp/null_$t.err
    echo "$t $(grep -o '"ms_per_step": [0-9.]*' /tmp/null_$t.log)" | tee -a $OUT/${R}_wall_attribution.txt
done
echo done

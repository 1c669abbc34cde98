#!/bin/bash
# Everything the round's DESIGN.md / README cite, regenerated on one box (GPU box only, ~12 min):
#   gpurun --timeout 1200 -- 'bash tools/make_evidence.sh r03'       then copy gpurun_out/prof_final/* into profiles/
set -o pipefail
R=${1:-r03}
cd "$GRAFT_REPO_ROOT" || exit 1
# (two gpurun calls fit the 1200 s limit: `bash tools/make_profiles.sh r04`, then `SKIP_PROFILES=1 bash tools/make_evidence.sh r04`)
if [ -z "$SKIP_PROFILES" ]; then bash tools/make_profiles.sh $R || exit 1; fi
OUT=gpurun_out/prof_final
mkdir -p $OUT
bash tools/prof_forced.sh $R || exit 1
cp gpurun_out/prof_fc/${R}_forced_collectives_* $OUT/
PPEA_FORCE_COLLECTIVES=1 python3 bench.py --no_cpu_baseline > $OUT/${R}_forced_collectives_n1.json 2> $OUT/fc.err || exit 1
python3 bench.py --no_cpu_baseline --rep_size l --batch 8 > $OUT/${R}_bench_n1_config_l.json 2>> $OUT/cfg.err || exit 1
python3 bench.py --no_cpu_baseline --dc --height 192 --width 512 --batch 4 > $OUT/${R}_bench_n1_config_dc192.json 2>> $OUT/cfg.err || exit 1
python3 bench.py --no_cpu_baseline --dc --height 512 --width 1024 --batch 4 > $OUT/${R}_bench_n1_config_dc512.json 2>> $OUT/cfg.err || exit 1
python3 bench.py --no_cpu_baseline --input_pipeline > $OUT/${R}_bench_n1_input_pipeline.json 2>> $OUT/cfg.err || exit 1
python3 tools/render_parity.py > $OUT/${R}_bf16_render_parity.txt 2>> $OUT/cfg.err || exit 1
python3 tools/render_parity.py e2e_render_l > $OUT/${R}_bf16_render_parity_l.txt 2>> $OUT/cfg.err || exit 1
python3 tools/render_parity.py e2e_render_dc > $OUT/${R}_bf16_render_parity_dc.txt 2>> $OUT/cfg.err || exit 1
python3 tools/aten_census.py > $OUT/${R}_aten_census.txt 2>> $OUT/cfg.err || exit 1
python3 tools/bench_costvol.py 2>/dev/null | grep -v amdgpu > $OUT/${R}_cost_volume.txt
# the step as a pure function (eager == eager == replay, with the second-consumer alias handed to the forked adapters):
(python3 tools/debug_repro.py --dtype f32 --alias-fork 2>&1; python3 tools/debug_repro.py --dtype bf16 --H 192 --W 640 --alias-fork 2>&1) \
    | grep "collectives_on\|differ\|loss 0" > $OUT/${R}_step_purity.txt
python3 tools/graph_edges_dup.py --captures 1 --dtype f32 2>&1 | grep -v "pretrained\|amdgpu" > $OUT/${R}_captured_graph_edges.txt
python3 tools/dwconv_phases.py 2>/dev/null | grep -v amdgpu > $OUT/${R}_dwconv_phases.txt
(cd tools && python3 bench_dwbn.py 2>/dev/null | grep -v amdgpu > ../$OUT/${R}_dwconv_fused_bn.txt)
(echo "# row-band kernels only (PPEA_DW_BM=0)"; PPEA_DW_BM=0 python3 tools/bench_dwconv.py --dtype bf16 2>/dev/null | grep -v amdgpu;
 echo "# default dispatch (batch-major variant on the 24 / 12 / 6-row maps, and for the 48-row data gradient)"; python3 tools/bench_dwconv.py --dtype bf16 2>/dev/null | grep -v amdgpu) > $OUT/${R}_dwconv_shapes.txt
# the captured step's own timeline (timestamps recorded into the graph), current topology and the round-3 one
python3 tools/step_timeline.py --replays 6 2>/dev/null > $OUT/${R}_step_timeline.txt
PPEA_POSE_SIDE=0 PPEA_MONO_SIDE=0 python3 tools/step_timeline.py --replays 6 2>/dev/null > $OUT/${R}_step_timeline_before.txt
echo evidence done

cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/q
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof1 -- python3 bench.py --steps 6 --warmup 3 > gpurun_out/q/bench.log 2>&1
f=$(find /tmp/prof1 -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv, sys, gzip
rows = list(csv.DictReader(open(sys.argv[1])))
rows = rows[-4516 * 5:]
with gzip.open("gpurun_out/q/trace_tail.csv.gz", "wt") as f:
    w = csv.writer(f)
    w.writerow(["Queue_Id", "Kernel_Name", "Start_Timestamp", "End_Timestamp", "Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z", "Workgroup_Size_X", "LDS_Block_Size", "VGPR_Count"])
    for r in rows:
        w.writerow([r["Queue_Id"], r["Kernel_Name"][:200], r["Start_Timestamp"], r["End_Timestamp"], r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"], r["Workgroup_Size_X"], r["LDS_Block_Size"], r["VGPR_Count"]])
PY

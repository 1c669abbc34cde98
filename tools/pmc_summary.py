#!/usr/bin/env python3
"""rocprofv3 --pmc counter CSVs -> one summary CSV (kernel, grid, counter, mean per dispatch, dispatches).

    python tools/pmc_summary.py OUT.csv DIR [DIR ...]      # DIRs = rocprofv3 -d outputs of separate --pmc passes

Counters are per dispatch (rocprofv3 sums over XCDs / SEs); FETCH_SIZE and WRITE_SIZE are in KB as rocprofv3 reports
them -- bench.py applies the gfx950 correction of MI355X_MICROARCH.md (FETCH_SIZE x2 for 16-byte-per-lane loads)."""
import collections
import csv
import glob
import os
import sys


def main():
    out, dirs = sys.argv[1], sys.argv[2:]
    acc = collections.defaultdict(lambda: [0.0, 0])
    for d in dirs:
        for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            with open(path) as f:
                for r in csv.DictReader(f):
                    name = r["Kernel_Name"]
                    if "ppea" not in name and "anonymous namespace" not in name:
                        continue
                    short = name.replace("(anonymous namespace)::", "").replace("void ", "")
                    key = (short.split("(")[0], r.get("Grid_Size", ""), r["Counter_Name"])
                    acc[key][0] += float(r["Counter_Value"])
                    acc[key][1] += 1
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "grid", "counter", "mean_per_dispatch", "dispatches"])
        for (k, g, c), (s, n) in sorted(acc.items()):
            w.writerow([k, g, c, f"{s / n:.1f}", n])
    print(f"{out}: {len(acc)} rows")


if __name__ == "__main__":
    main()

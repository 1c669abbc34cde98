#!/usr/bin/env python3
"""Wall-time attribution by ablation: run bench.py's timed graph with one kernel family replaced by no-ops
(PPEA_NULL=prefix[,prefix...] of the C-ABI names after 'ppea_') -- the results are garbage, the step time tells how
much of the step's wall time hangs on that family (kernel time under a profiler does not: branches overlap).
A TOOL: the product has no such switch."""
import os, runpy, sys
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(root, "ppea-depth_amd"))
from ppeadepth import _abi
tags = [t for t in os.environ.get("PPEA_NULL", "").split(",") if t]
n = 0
for name in _abi.SIGNATURES:
    if name.endswith("_bytes") or name.endswith("_partials") or name in ("ppea_abi_version", "ppea_nhwc_bn_slabs",
                                                                          "ppea_nhwc_bias_elu_slabs"):
        continue
    if any(name.startswith("ppea_" + t) for t in tags):
        setattr(_abi.lib, name, lambda *a: 0)
        n += 1
print(f"[null_ablation] {tags}: {n} entry points replaced by no-ops", file=sys.stderr, flush=True)
sys.argv = ["bench.py", "--steps", "8", "--warmup", "2", "--no_cpu_baseline"]
runpy.run_path(os.path.join(root, "bench.py"), run_name="__main__")

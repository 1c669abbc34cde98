"""Debug: TrainEngine fp32 steps in sequence (as tests/test_e2e_gpu.py runs them) with the second-consumer alias on."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "ppea-depth_amd"))
import test_e2e_gpu as T
dev = torch.device("cuda:0")
import conftest
cache = {}
def golden(name):
    if name not in cache:
        cache[name] = conftest.load_golden(name)
    return cache[name]
seq = [("e2e_small", False), ("e2e_small", True), ("e2e_full", False), ("e2e_full", True), ("e2e_dc", True)]
if len(sys.argv) > 1:
    seq = [(a.split(":")[0], a.split(":")[1] == "1") for a in sys.argv[1:]]
for name, graph in seq:
    res = T._engine_step(name, golden, dev, bf16=False, graph=graph, **T.CONFIG_OF.get(name, {}))
    errs = T._errors(*res)
    bad = {k: v for k, v in errs.items() if v > T.TOL_F32[k.split(":")[0]]}
    print(name, graph, "bad:", len(bad))
    for k, v in sorted(bad.items(), key=lambda kv: -kv[1])[:12]:
        print("     %-90s %.4g" % (k, v))

#!/usr/bin/env python3
"""Registers, LDS, occupancy and spills of every __global__ function in ppea-depth_amd/csrc, as the compiler reports them
(hipcc -Rpass-analysis=kernel-resource-usage; cross-compiles for gfx950, no GPU needed):

    python3 tools/kernel_resources.py > profiles/rNN_kernel_resources.txt
"""
import concurrent.futures as cf
import glob
import os
import re
import subprocess
import tempfile

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
CSRC = os.path.join(ROOT, "ppea-depth_amd", "csrc")
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-slp-vectorize", "--cuda-device-only", "-c",
         "-Rpass-analysis=kernel-resource-usage"]          # the Makefile's flags + device only + the remark
KEYS = ("TotalSGPRs", "VGPRs", "AGPRs", "ScratchSize [bytes/lane]", "Occupancy [waves/SIMD]", "SGPRs Spill", "VGPRs Spill",
        "LDS Size [bytes/block]")
PAT = re.compile(r"remark:\s+(Function Name|" + "|".join(re.escape(k) for k in KEYS) + r"): (\S+)")


def remarks(src):
    with tempfile.TemporaryDirectory() as tmp:
        r = subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + [src, "-o", os.path.join(tmp, "o.o")], cwd=CSRC,
                           capture_output=True, text=True)
    rows, cur = [], None
    for line in r.stderr.splitlines():
        m = PAT.search(line)
        if not m:
            continue
        if m.group(1) == "Function Name":
            cur = {"file": os.path.basename(src), "name": m.group(2)}
            rows.append(cur)
        elif cur is not None:
            cur[m.group(1)] = m.group(2)
    return rows


def main():
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    with cf.ThreadPoolExecutor(6) as ex:
        rows = [r for rs in ex.map(remarks, srcs) for r in rs]
    dem = subprocess.run(["c++filt"], input="\n".join(r["name"] for r in rows), capture_output=True, text=True).stdout.splitlines()
    print("# Per-kernel resources of every __global__ function in ppea-depth_amd/csrc (tools/kernel_resources.py: the Makefile's")
    print("# flags + -Rpass-analysis=kernel-resource-usage).  VGPR + AGPR <= 512 per lane; occupancy in waves per SIMD as the register")
    print("# count allows (LDS may lower it); scratch / spills > 0 = register pressure the compiler could not place.")
    print(f"{'file':18s} {'VGPR':>5s} {'AGPR':>5s} {'SGPR':>5s} {'occ':>4s} {'scratch':>8s} {'vspill':>7s} {'sspill':>7s} {'LDS':>7s}  kernel")
    spilled = scratch = 0
    for r, d in zip(rows, dem):
        d = re.sub(r"\(anonymous namespace\)::", "", d)
        d = re.sub(r"^void ", "", d)
        d = re.sub(r"\(.*$", "", d)
        g = lambda k: r.get(k, "?")                                                          # noqa: E731
        spilled += g("SGPRs Spill") not in ("0", "?")
        scratch += (g("VGPRs Spill") not in ("0", "?")) or (g("ScratchSize [bytes/lane]") not in ("0", "?"))
        print(f"{r['file']:18s} {g('VGPRs'):>5s} {g('AGPRs'):>5s} {g('TotalSGPRs'):>5s} {g('Occupancy [waves/SIMD]'):>4s} "
              f"{g('ScratchSize [bytes/lane]'):>8s} {g('VGPRs Spill'):>7s} {g('SGPRs Spill'):>7s} {g('LDS Size [bytes/block]'):>7s}  {d}")
    print(f"# {len(rows)} kernels; {scratch} with VGPR spills or scratch memory; {spilled} park SGPRs in VGPR lanes (no memory traffic)")


if __name__ == "__main__":
    main()

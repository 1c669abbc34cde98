#!/usr/bin/env python3
"""Capture-race investigation: the second-consumer alias of a BatchNorm output handed to a
FORKED adapter gave wrong encoder gradients in replayed hipGraphs.

    python tools/graph_edges_dup.py [--captures 3] [--dtype f32|bf16] [--out gpurun_out/dup_dot]

Runs the e2e_small engine step in the known-good topology (alias for in-line adapters only) and in the suspect one (alias
for the forked adapters too).  For every capture the step is replayed from the restored state and judged against the
reference golden (tests/test_e2e_gpu.py tolerances) and against an eager engine step.  The captured hipGraph itself is read
back through the HIP graph API (hipGraphGetNodes / GetEdges / KernelNodeGetParams; torch keeps the hipGraph_t with
keep_graph=True): kernel names, dependency edges and every kernel's pointer arguments.  Two analyses:

 (1) edges: for each `bn_bwd_channel*_dup` launch, is every other kernel that touches its `dyb` buffer an ancestor or a
     descendant?  (a kernel that shares the buffer and is UNORDERED with the launch is the race)
 (2) buffer sharing: every pair of UNORDERED kernel nodes that share a device address in their arguments, as a set of
     (kernel, kernel) names; the pairs that exist only in the suspect topology are printed.
"""
import argparse
import ctypes as ct
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "ppea-depth_amd"))

import torch  # noqa: E402

_GRAPHS = []
_RealGraph = torch.cuda.CUDAGraph


def _kept_graph(*a, **k):
    g = _RealGraph(keep_graph=True)
    _GRAPHS.append(g)
    return g


class Dim3(ct.Structure):
    _fields_ = [("x", ct.c_uint), ("y", ct.c_uint), ("z", ct.c_uint)]


class KernelNodeParams(ct.Structure):
    _fields_ = [("blockDim", Dim3), ("extra", ct.c_void_p), ("func", ct.c_void_p), ("gridDim", Dim3),
                ("kernelParams", ct.c_void_p), ("sharedMemBytes", ct.c_uint)]


class IoVec(ct.Structure):
    _fields_ = [("base", ct.c_void_p), ("len", ct.c_size_t)]


_libc = ct.CDLL("libc.so.6", use_errno=True)
_libc.process_vm_readv.restype = ct.c_ssize_t
_libc.process_vm_readv.argtypes = [ct.c_int, ct.POINTER(IoVec), ct.c_ulong, ct.POINTER(IoVec), ct.c_ulong, ct.c_ulong]


def safe_read(addr, n):
    """n bytes at addr of this process, or None if the range is not readable (never faults)."""
    if not addr:
        return None
    buf = ct.create_string_buffer(n)
    loc = IoVec(ct.cast(buf, ct.c_void_p), n)
    rem = IoVec(ct.c_void_p(addr), n)
    got = _libc.process_vm_readv(os.getpid(), ct.byref(loc), 1, ct.byref(rem), 1, 0)
    return buf.raw if got == n else None


def read_graph(raw, log):
    """-> (names: list, kinds: list, edges: list[(i, j)], args: list[set of device addresses])."""
    hip = ct.CDLL("libamdhip64.so")
    hip.hipKernelNameRefByPtr.restype = ct.c_char_p
    hip.hipKernelNameRefByPtr.argtypes = [ct.c_void_p, ct.c_void_p]
    hip.hipKernelNameRef.restype = ct.c_char_p
    hip.hipKernelNameRef.argtypes = [ct.c_void_p]
    graph = ct.c_void_p(raw)
    n = ct.c_size_t(0)
    assert hip.hipGraphGetNodes(graph, None, ct.byref(n)) == 0
    nodes = (ct.c_void_p * n.value)()
    assert hip.hipGraphGetNodes(graph, nodes, ct.byref(n)) == 0
    ne = ct.c_size_t(0)
    assert hip.hipGraphGetEdges(graph, None, None, ct.byref(ne)) == 0
    fr, to = (ct.c_void_p * ne.value)(), (ct.c_void_p * ne.value)()
    assert hip.hipGraphGetEdges(graph, fr, to, ct.byref(ne)) == 0
    index = {nodes[i]: i for i in range(n.value)}
    edges = [(index[fr[i]], index[to[i]]) for i in range(ne.value)]
    names, kinds, args, first = [], [], [], []
    lo, hi = 1 << 40, 1 << 48          # device heap addresses on this platform sit far above host heap / stack
    for i in range(n.value):
        t = ct.c_int(-1)
        hip.hipGraphNodeGetType(ct.c_void_p(nodes[i]), ct.byref(t))
        kinds.append(t.value)
        found, words = set(), []
        name = {1: "<memcpy>", 2: "<memset>", 3: "<host>", 5: "<empty>", 6: "<waitEvent>", 7: "<eventRecord>"}.get(t.value, f"<type {t.value}>")
        if t.value == 0:
            p = KernelNodeParams()
            if hip.hipGraphKernelNodeGetParams(ct.c_void_p(nodes[i]), ct.byref(p)) == 0:
                nm = hip.hipKernelNameRefByPtr(p.func, None)
                if not nm and safe_read(p.func, 64) is not None:
                    nm = hip.hipKernelNameRef(p.func)
                name = nm.decode(errors="replace") if nm else f"<kernel {p.func:#x}>"
                if p.kernelParams:
                    tab = safe_read(p.kernelParams, 8 * 24)
                    for k in range(24 if tab else 0):
                        a = int.from_bytes(tab[8 * k:8 * k + 8], "little")
                        v = safe_read(a, 8) if a else None
                        if v is None:
                            break
                        words.append(int.from_bytes(v, "little"))
                if p.extra:
                    ex = safe_read(p.extra, 40)
                    if ex:
                        w = [int.from_bytes(ex[8 * k:8 * k + 8], "little") for k in range(5)]
                        if w[0] == 1 and w[2] == 2:            # HIP_LAUNCH_PARAM_BUFFER_POINTER / _SIZE
                            sz = safe_read(w[3], 8)
                            size = min(int.from_bytes(sz, "little"), 512) if sz else 0
                            buf = safe_read(w[1], size) if size else None
                            if buf:
                                words += [int.from_bytes(buf[k:k + 8], "little") for k in range(0, size - 7, 8)]
                found = {w for w in words if lo <= w < hi and w % 4 == 0}
        names.append(name)
        args.append(found)
        first.append(words[:3])
    log(f"   graph: {n.value} nodes ({sum(1 for k in kinds if k == 0)} kernels), {ne.value} edges")
    return names, kinds, edges, args, first


def ancestors_bitsets(n, edges):
    preds = [[] for _ in range(n)]
    succs = [[] for _ in range(n)]
    indeg = [0] * n
    for a, b in edges:
        preds[b].append(a)
        succs[a].append(b)
        indeg[b] += 1
    order, stack = [], [i for i in range(n) if indeg[i] == 0]
    while stack:
        v = stack.pop()
        order.append(v)
        for w in succs[v]:
            indeg[w] -= 1
            if indeg[w] == 0:
                stack.append(w)
    anc = [0] * n
    for v in order:
        m = 0
        for p in preds[v]:
            m |= anc[p] | (1 << p)
        anc[v] = m
    return anc, preds


def short(name):
    name = re.sub(r"void |\(anonymous namespace\)::|at::native::|ppea::", "", name)
    return name[:100]


def analyse(names, kinds, edges, args, first, log):
    n = len(names)
    anc, preds = ancestors_bitsets(n, edges)
    users = {}
    for i, s in enumerate(args):
        for a in s:
            users.setdefault(a, []).append(i)

    def ordered(i, j):
        return bool((anc[j] >> i) & 1) or bool((anc[i] >> j) & 1)

    # bn_bwd_channel(dy, dyb, ...) / bn_bwd_channel_next(dy2, dy2b, ...): a second-consumer launch has a non-null 2nd pointer
    dup = [i for i, nm in enumerate(names) if "bn_bwd_channel" in nm and len(first[i]) > 1 and first[i][1] != 0]
    log(f"   second-consumer launches: {len(dup)}")
    races = 0
    for d in dup[:8]:
        log(f"   node {d}: {short(names[d])}")
        for p in preds[d]:
            log(f"        <- {p}: {short(names[p])}")
    for d in dup:
        for a in (first[d][1],):                     # the alias gradient's buffer
            for u in users.get(a, []):
                if u != d and not ordered(u, d):
                    races += 1
                    if races <= 30:
                        log(f"   UNORDERED with dup launch {d} ({short(names[d])}) on buffer {a:#x}: node {u} {short(names[u])}")
    log(f"   kernels sharing a buffer with a second-consumer launch without an ordering edge: {races}")
    pairs = {}
    for a, us in users.items():
        if len(us) < 2 or len(us) > 64:
            continue
        for x in range(len(us)):
            for y in range(x + 1, len(us)):
                i, j = us[x], us[y]
                if not ordered(i, j):
                    key = tuple(sorted((short(names[i]), short(names[j]))))
                    pairs[key] = pairs.get(key, 0) + 1
    log(f"   unordered kernel pairs that share a device address: {sum(pairs.values())} ({len(pairs)} distinct name pairs)")
    return pairs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--captures", type=int, default=3)
    ap.add_argument("--dtype", default="f32")
    ap.add_argument("--name", default="e2e_small")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "dup_dot"))
    args = ap.parse_args()
    os.makedirs(args.out, exist_ok=True)
    lines = []

    def log(s):
        print(s, flush=True)
        lines.append(s)

    import conftest
    import test_e2e_gpu as T
    from ppeadepth.networks import replknet_adapter as rka
    dev = torch.device("cuda:0")
    cache = {}

    def golden(name):
        if name not in cache:
            cache[name] = conftest.load_golden(name)
        return cache[name]

    bf16 = args.dtype == "bf16"
    cfg = T.CONFIG_OF.get(args.name, {})
    ref = T._engine_step(args.name, golden, dev, bf16=bf16, graph=False, **cfg)
    ref_grads = ref[-1]
    log(f"dtype {args.dtype}; fixture {args.name}; eager reference taken")
    torch.cuda.CUDAGraph = _kept_graph
    pair_sets = {}
    for fork in (False, True):
        rka.BN_DUP_FORKED = fork                 # the suspect topology: alias handed to the forked adapters too
        for i in range(args.captures if fork else 1):
            _GRAPHS.clear()
            res = T._engine_step(args.name, golden, dev, bf16=bf16, graph=True, **cfg)
            grads = res[-1]
            errs = T._errors(*res)
            bad_gold = {k: v for k, v in errs.items() if v > T.TOL_F32[k.split(":")[0]]} if not bf16 else {}
            nbad = sum(1 for k, v in grads.items()
                       if float((v - ref_grads[k]).abs().max()) > 1e-3 * float(ref_grads[k].abs().max()) + 1e-12)
            log(f"[alias to forked adapters = {fork}] capture {i}: {len(bad_gold)} golden keys out of tolerance, "
                f"{nbad} of {len(grads)} gradients > 1e-3 from the eager step")
            for k, v in sorted(bad_gold.items(), key=lambda kv: -kv[1])[:6]:
                log(f"       {k} {v:.4g}")
            if i == 0:
                names, kinds, edges, kargs, first = read_graph(_GRAPHS[-1].raw_cuda_graph(), log)
                pair_sets[fork] = analyse(names, kinds, edges, kargs, first, log)
                with open(os.path.join(args.out, f"graph_{args.dtype}_{'fork' if fork else 'inline'}.txt"), "w") as f:
                    for j, nm in enumerate(names):
                        f.write(f"N {j} {kinds[j]} {short(nm)} | {' '.join(hex(a) for a in sorted(kargs[j]))}\n")
                    for a, b in edges:
                        f.write(f"E {a} {b}\n")
    new = {k: v for k, v in pair_sets[True].items() if k not in pair_sets[False]}
    log(f"unordered buffer-sharing name pairs only in the suspect topology: {len(new)}")
    for k, v in sorted(new.items(), key=lambda kv: -kv[1])[:40]:
        log(f"   {v:4d} x  {k[0]}   ||   {k[1]}")
    with open(os.path.join(args.out, f"summary_{args.dtype}.txt"), "w") as f:
        f.write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()

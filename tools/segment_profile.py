"""Which part of the model does the step time go to?  GPU box only.

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/seg -- python3 tools/segment_profile.py run
    python tools/segment_profile.py parse gpurun_out/seg/**/*kernel_trace.csv

`run` executes eager training steps with MARKER kernels (a float64 fill whose size encodes an id) launched
from module forward / backward hooks on the step's stream; `parse` replays the kernel trace in time order,
keeps a stack of open segments and charges every kernel to the innermost open one.
"""
import collections
import csv
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ppea-depth_amd"))
sys.path.insert(0, ROOT)
os.environ.setdefault("MIOPEN_FIND_MODE", "NORMAL")

UNIT = 8192          # marker id i -> fill of i * UNIT doubles
N_CAL = 120


def labels_for(model):
    """(module, label) for the modules we segment by; label = path of coarse names."""
    import torch
    from ppeadepth.networks import replknet_adapter as rka
    out = []
    for name, m in model.named_modules():
        top = name.split(".")[0]
        if name in ("encoder", "mono_encoder", "depth", "mono_depth", "pose_encoder", "pose"):
            out.append((m, name))
        elif isinstance(m, rka.RepLKBlock):
            out.append((m, "lkblock"))
        elif isinstance(m, rka.ConvFFN):
            out.append((m, "ffn"))
        elif isinstance(m, (rka.B_Adapter, rka.Adapter)):
            out.append((m, "adapter"))
        elif isinstance(m, rka.ReparamLargeKernelConv):
            out.append((m, "lk_dw"))
        elif top in ("depth", "mono_depth") and type(m).__name__ in ("ConvBlock", "Conv3x3") and \
                name.count(".") <= 2 and not name.endswith(".conv"):
            out.append((m, "dec_conv:" + top))
        elif top == "pose_encoder" and type(m).__name__ in ("BasicBlock", "Bottleneck"):
            out.append((m, "pose_block"))
        elif top == "pose" and isinstance(m, torch.nn.Conv2d):
            out.append((m, "pose_dec_conv"))
        elif top in ("encoder", "mono_encoder") and (name.endswith(".stem") or ".transitions." in name and
                                                     name.count(".") == 2):
            out.append((m, "stem/transition"))
    return out


def run():
    import torch
    from ppeadepth import dist as pdist
    from ppeadepth import networks, options, rng
    from ppeadepth.trainer import Trainer
    from ppeadepth import synthetic as synth

    dev = torch.device("cuda", 0)
    H, W, B = 192, 640, int(os.environ.get("SEG_BATCH", "12"))
    opt = options.default_options(height=H, width=W, batch_size=B, rep_size="b")
    torch.manual_seed(0)
    model = networks.RepDepth(opt)
    synth.fill_state_dict(model)
    model.to(dev).train()
    trainer = Trainer(opt, model, dev, amp_dtype=torch.bfloat16)
    engine = pdist.TrainEngine(trainer, bf16_params=True)
    rng.set_mode("device")
    networks.repdepth.TWO_STREAMS = False            # one stream: markers and kernels in one time order
    networks.replknet_adapter.ADAPTER_STREAMS = False
    inputs = {k: v.to(dev) for k, v in synth.make_inputs(B, H, W, seed=1234, smooth=True).items()}
    random.seed(1000)
    for _ in range(3):
        engine.step(dict(inputs))
    torch.cuda.synchronize()

    names = ["<cal>"]                     # id -> label; even = open, odd = close

    def marker(i):
        torch.full((i * UNIT,), 0.0, dtype=torch.float64, device=dev)

    def new_segment(label):
        names.append(label)
        return len(names) - 1

    enabled = [False]

    def hook_pair(seg_open, seg_close):
        def o(*a):
            if enabled[0]:
                marker(seg_open)

        def c(*a):
            if enabled[0]:
                marker(seg_close)
        return o, c

    seg_ids = {}
    for m, label in labels_for(model):
        if label not in seg_ids:
            seg_ids[label] = (new_segment("+" + label + " fwd"), new_segment("-" + label + " fwd"),
                              new_segment("+" + label + " bwd"), new_segment("-" + label + " bwd"))
        fo, fc, bo, bc = seg_ids[label]
        o, c = hook_pair(fo, fc)
        m.register_forward_pre_hook(o)
        m.register_forward_hook(c)
        o, c = hook_pair(bo, bc)
        m.register_full_backward_pre_hook(o)
        m.register_full_backward_hook(c)
    phases = {p: (new_segment("+" + p), new_segment("-" + p)) for p in ("forward+loss", "backward", "optimizer")}
    assert len(names) < N_CAL

    with open(os.environ.get("SEG_NAMES", os.path.join(ROOT, "gpurun_out", "seg_names.txt")), "w") as f:
        f.write("\n".join(names))

    stream = engine.stream
    stream.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(stream):
        for _ in range(2):               # hooks installed: one untimed step, then the profiled ones
            engine._step_body(dict(inputs))
        torch.cuda.synchronize()
        enabled[0] = True
        marker(N_CAL)                     # start-of-profile sentinel
        for _ in range(int(os.environ.get("SEG_STEPS", "2"))):
            marker(phases["forward+loss"][0])
            outputs, losses = trainer.process_batch(dict(inputs), is_train=True)
            marker(phases["forward+loss"][1])
            for p in engine.params:
                p.grad = None
            marker(phases["backward"][0])
            losses["loss"].backward()
            marker(phases["backward"][1])
            marker(phases["optimizer"][0])
            engine._optimizer_phase()
            marker(phases["optimizer"][1])
        marker(N_CAL + 1)                 # end sentinel
        torch.cuda.synchronize()
    print("segments:", len(names))


def parse(path, names_path):
    names = open(names_path).read().split("\n")
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            gx = int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0)
            wx = int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 1)) or 1)
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], gx, wx))
    rows.sort()
    is_marker = lambda n: "FillFunctor<double>" in n
    marks = [r for r in rows if is_marker(r[2])]
    if not marks:
        print("no markers found"); return
    # grid of a fill of n doubles = ceil(n / per_block) * wg: calibrate per_block from the two sentinels
    # (ids N_CAL, N_CAL + 1 are the first / last marker of the profiled region)
    per = None
    for cand in (256, 512, 1024, 2048, 4096, 8192):
        if marks[-1][3] // marks[-1][4] == -(-(N_CAL + 1) * UNIT // cand):
            hits = [m for m in marks if m[3] // m[4] == -(-N_CAL * UNIT // cand)]
            if hits:
                per = cand
                break
    if per is None:
        print("cannot calibrate marker sizes", marks[-1][3:], marks[0][3:]); return
    ident = lambda r: (r[3] // r[4]) * per // UNIT
    start = max(i for i, r in enumerate(rows) if is_marker(r[2]) and ident(r) == N_CAL)
    end = max(i for i, r in enumerate(rows) if is_marker(r[2]) and ident(r) == N_CAL + 1)
    steps = sum(1 for r in rows[start + 1:end] if is_marker(r[2]) and names[ident(r)] == "+optimizer")
    stack, agg = [], collections.defaultdict(lambda: [0.0, 0])
    kagg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    for r in rows[start + 1:end]:
        if is_marker(r[2]):
            nm = names[ident(r)]
            if nm[0] == "+":
                stack.append(nm[1:])
            elif stack and stack[-1] == nm[1:]:
                stack.pop()
            elif nm[1:] in stack:           # unbalanced close (a backward hook that never opened): unwind
                while stack and stack.pop() != nm[1:]:
                    pass
            continue
        # path: phase / top module / innermost class label
        tops = [s for s in stack if s.split(" ")[0] in ("encoder", "mono_encoder", "depth", "mono_depth",
                                                        "pose_encoder", "pose")]
        inner = [s for s in stack if s.split(" ")[0] in ("lkblock", "ffn", "adapter", "lk_dw", "stem/transition",
                                                         "pose_block", "pose_dec_conv")
                 or s.startswith("dec_conv")]
        phase = stack[0] if stack else "?"
        key = (phase, tops[-1].split(" ")[0] if tops else "(trainer)",
               "/".join(s.split(" ")[0] for s in inner) if inner else "-")
        agg[key][0] += r[1] - r[0]; agg[key][1] += 1
        kagg[key][r[2]][0] += r[1] - r[0]; kagg[key][r[2]][1] += 1
    tot = sum(v[0] for v in agg.values())
    print(f"{steps} steps, kernel time {tot / 1e6 / steps:.1f} ms/step, "
          f"{sum(v[1] for v in agg.values()) // steps} launches/step")
    for key, (t, c) in sorted(agg.items(), key=lambda x: -x[1][0]):
        print(f"{t / 1e6 / steps:8.2f} ms {100 * t / tot:5.1f}% {c // steps:6d} launches  {' | '.join(key)}")
        if t / tot > 0.02:
            for n, (kt, kc) in sorted(kagg[key].items(), key=lambda x: -x[1][0])[:8]:
                print(f"            {kt / 1e6 / steps:7.2f} ms {kc // steps:5d}x avg {kt / kc / 1e3:7.1f} us  {n[:100]}")


def trim(path, out):
    """Keep (start, end, name, grid, wg) of the kernels after the last start sentinel-sized marker burst."""
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            gx = int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0)
            wx = int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 1)) or 1)
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], gx, wx))
    rows.sort()
    idx = [i for i, r in enumerate(rows) if "FillFunctor<double>" in r[2]]
    rows = rows[idx[0]:]
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Start_Timestamp", "End_Timestamp", "Kernel_Name", "Grid_Size_X", "Workgroup_Size_X"])
        w.writerows(rows)


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run()
    elif sys.argv[1] == "trim":
        trim(sys.argv[2], sys.argv[3])
    else:
        parse(sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else os.path.join(ROOT, "gpurun_out", "seg_names.txt"))

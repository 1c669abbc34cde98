#!/usr/bin/env python3
"""Per-layer timing of the dense convolutions of the bf16 step (B=12, 31B): this build's implicit-GEMM kernels
(forward, data gradient, weight gradient) against the library convs (MIOpen, channels_last bf16) on the same shapes.
    python tools/bench_conv.py [filter]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ppea-depth_amd"))
os.environ.setdefault("MIOPEN_FIND_MODE", "NORMAL")
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from ppeadepth import ops  # noqa: E402

B = 12
LAYERS = [  # name, N, Cin, H, W, Cout, k, stride, pad, reflect, act, bias, count per step
    ("dec 1024->512 @6x20", B, 1024, 6, 20, 512, 3, 1, 1, True, "elu", True, 2),
    ("dec 1024->512 @12x40", B, 1024, 12, 40, 512, 3, 1, 1, True, "elu", True, 2),
    ("dec 512->256 @12x40", B, 512, 12, 40, 256, 3, 1, 1, True, "elu", True, 2),
    ("dec 512->256 @24x80", B, 512, 24, 80, 256, 3, 1, 1, True, "elu", True, 2),
    ("dec 256->128 @24x80", B, 256, 24, 80, 128, 3, 1, 1, True, "elu", True, 2),
    ("dec 256->128 @48x160", B, 256, 48, 160, 128, 3, 1, 1, True, "elu", True, 2),
    ("dec 128->64 @48x160", B, 128, 48, 160, 64, 3, 1, 1, True, "elu", True, 2),
    ("dec 64->64 @96x320", B, 64, 96, 320, 64, 3, 1, 1, True, "elu", True, 2),
    ("dec 64->32 @96x320", B, 64, 96, 320, 32, 3, 1, 1, True, "elu", True, 2),
    ("dec 32->32 @192x640", B, 32, 192, 640, 32, 3, 1, 1, True, "elu", True, 2),
    ("disp 32->1 @192x640", B, 32, 192, 640, 1, 3, 1, 1, True, "sigmoid", True, 2),
    ("reduce 224->128 @48x160", B, 224, 48, 160, 128, 3, 1, 1, False, "relu", True, 1),
    ("stem 3->128 s2 @192x640", B, 8, 192, 640, 128, 3, 2, 1, False, "none", False, 3),
    ("pose conv1 7x7 s2", 2 * B, 8, 192, 640, 64, 7, 2, 3, False, "none", False, 1),
    ("res 64->64 @48x160", 2 * B, 64, 48, 160, 64, 3, 1, 1, False, "none", False, 4),
    ("res 64->128 s2", 2 * B, 64, 48, 160, 128, 3, 2, 1, False, "none", False, 1),
    ("res 128->128 @24x80", 2 * B, 128, 24, 80, 128, 3, 1, 1, False, "none", False, 3),
    ("res 128->256 s2", 2 * B, 128, 24, 80, 256, 3, 2, 1, False, "none", False, 1),
    ("res 256->256 @12x40", 2 * B, 256, 12, 40, 256, 3, 1, 1, False, "none", False, 3),
    ("res 256->512 s2", 2 * B, 256, 12, 40, 512, 3, 2, 1, False, "none", False, 1),
    ("res 512->512 @6x20", 2 * B, 512, 6, 20, 512, 3, 1, 1, False, "none", False, 3),
    ("posedec 256->256 @6x20", 2 * B, 256, 6, 20, 256, 3, 1, 1, False, "relu", True, 2),
]


def timeit(fn, n=20, reps=3):
    """Median of `reps` event-bracketed batches of n calls: one host hiccup in a batch must not become the layer's figure."""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    out = []
    for _ in range(reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(n):
            fn()
        e.record()
        e.synchronize()
        out.append(s.elapsed_time(e) / n * 1e3)
    return sorted(out)[reps // 2]


def main():
    flt = sys.argv[1] if len(sys.argv) > 1 else ""
    dev = torch.device("cuda:0")
    tot = {"ours": 0.0, "lib": 0.0}
    print(f"{'layer':28s} {'GF':>6s} | {'fwd us':>8s} {'TF/s':>6s} {'lib':>8s} | {'dgrad':>8s} {'lib':>8s} | {'wgrad':>8s} {'lib':>8s}")
    for (name, N, Cin, H, W, Cout, k, stride, pad, reflect, act, has_bias, cnt) in LAYERS:
        if flt not in name:
            continue
        x = torch.randn(N, Cin, H, W, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
        w = (torch.randn(Cout, Cin, k, k, device=dev) / (Cin * k * k) ** 0.5).bfloat16().requires_grad_(True)
        b = torch.zeros(Cout, device=dev).bfloat16().requires_grad_(True) if has_bias else None
        Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
        gflop = 2.0 * N * Ho * Wo * Cout * Cin * k * k / 1e9 * (3 / 8 if Cin == 8 and k == 3 else (6 / 8 if Cin == 8 else 1))
        image_fed = Cin == 8
        # ---- this build ----------------------------------------------------------------------------
        xr = x.clone().requires_grad_(not image_fed)
        y = ops.conv2d_nhwc(xr, w, b, stride, pad, reflect, act)
        go = torch.randn_like(y)
        t_f = timeit(lambda: ops.conv2d_nhwc(x, w.detach(), None if b is None else b.detach(), stride, pad, reflect, act))
        C8 = max(8, (Cout + 7) // 8 * 8)
        dz = torch.randn(N, C8, Ho, Wo, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
        wt = ops._conv_packed(w.detach().clone() if Cout % 8 == 0 else torch.cat([w.detach(), w.new_zeros(C8 - Cout, Cin, k, k)]), True)
        t_d = 0.0
        if not image_fed:
            if reflect:
                t_d = timeit(lambda: ops.conv_nhwc_raw(dz, wt, None, Cin, k, k, 1, k - 1, False, stride, Ho + k - 1, Wo + k - 1, 0, False))
            else:
                t_d = timeit(lambda: ops.conv_nhwc_raw(dz, wt, None, Cin, k, k, 1, k - 1 - pad, False, stride, H, W, 0, False))
        ws = torch.empty(ops._abi.lib.ppea_conv_wgrad_workspace_bytes(N, Cin, C8, k, k, stride, Ho, Wo) // 4, device=dev)
        dw = torch.empty(C8, Cin, k, k, device=dev, dtype=torch.bfloat16)
        if image_fed:
            # the image-fed layers (stem[0], pose conv1) take the row-packed kernels in the step (ops._ConvImage): time THOSE
            cin_real = 3 if k == 3 else 6
            wsi = torch.empty(ops._abi.lib.ppea_conv_image_wgrad_workspace_bytes(N, Cout, k, Ho, Wo) // 4, device=dev)
            dwi = torch.empty(Cout, cin_real, k, k, device=dev, dtype=torch.bfloat16)
            t_w = timeit(lambda: ops.call("ppea_conv_image_wgrad_bf16", ops._raw(dz), ops._raw(x), ops.ptr(dwi), 1, ops.ptr(wsi), N, H, W,
                                          cin_real, Cout, k, 2, pad, Ho, Wo, ops.stream_ptr()))
        else:
            t_w = timeit(lambda: ops.call("ppea_conv_wgrad_nhwc_bf16", ops._raw(dz), ops._raw(x), ops.ptr(dw), 1, ops.ptr(ws), N, H, W,
                                          Cin, C8, k, k, stride, pad, int(reflect), Ho, Wo, ops.stream_ptr()))
        # ---- library (what the step used before): pad kernel + conv + bias/act kernels are NOT counted, conv only ---
        xl = (F.pad(x, (1, 1, 1, 1), mode="reflect") if reflect else x).contiguous(memory_format=torch.channels_last)
        wl = w.detach().contiguous(memory_format=torch.channels_last)
        p = 0 if reflect else pad
        l_f = timeit(lambda: F.conv2d(xl, wl, None, stride, p))
        yl = F.conv2d(xl, wl, None, stride, p)
        gl = torch.randn_like(yl)
        l_d = 0.0 if image_fed else timeit(lambda: torch.ops.aten.convolution_backward(
            gl, xl, wl, None, [stride, stride], [p, p], [1, 1], False, [0, 0], 1, [True, False, False]))
        l_w = timeit(lambda: torch.ops.aten.convolution_backward(
            gl, xl, wl, None, [stride, stride], [p, p], [1, 1], False, [0, 0], 1, [False, True, False]))
        print(f"{name:28s} {gflop:6.1f} | {t_f:8.1f} {gflop / t_f * 1e3:6.0f} {l_f:8.1f} | {t_d:8.1f} {l_d:8.1f} | {t_w:8.1f} {l_w:8.1f}   x{cnt}",
              flush=True)
        tot["ours"] += cnt * (t_f + t_d + t_w)
        tot["lib"] += cnt * (l_f + l_d + l_w)
    print(f"per step (fwd + dgrad + wgrad, counts applied): ours {tot['ours'] / 1e3:.2f} ms   library {tot['lib'] / 1e3:.2f} ms")


if __name__ == "__main__":
    main()

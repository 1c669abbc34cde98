#!/usr/bin/env python3
"""Checksum of the decoder's high-resolution 3x3 convs (forward and data gradient) at the bench size -- run with and
without PPEA_CONV_NO_RESIDENT=1: the persistent resident-weight kernel must give the per-tile kernel's bytes."""
import hashlib, os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ppea-depth_amd"))
from ppeadepth import ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(5)
for (N, Cin, H, W, Cout, reflect, act) in [(12, 32, 192, 640, 32, True, "elu"), (12, 32, 192, 640, 1, True, "sigmoid"),
                                           (12, 64, 96, 320, 64, True, "elu"), (12, 64, 96, 320, 32, True, "elu")]:
    x = torch.randn(N, Cin, H, W, generator=g).bfloat16().to(dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5).bfloat16().to(dev).requires_grad_(True)
    b = (0.2 * torch.randn(Cout, generator=g)).bfloat16().to(dev).requires_grad_(True)
    go = torch.randn(N, Cout, H, W, generator=g).bfloat16().to(dev).contiguous(memory_format=torch.channels_last)
    for rep in range(3):
        x.grad = None
        y = ops.conv2d_nhwc(x, w, b, 1, 1, reflect, act, False)
        y.backward(go)
        torch.cuda.synchronize()
        hy = hashlib.md5(y.detach().contiguous(memory_format=torch.channels_last).view(torch.int16).cpu().numpy().tobytes()).hexdigest()[:10]
        hx = hashlib.md5(x.grad.contiguous(memory_format=torch.channels_last).view(torch.int16).cpu().numpy().tobytes()).hexdigest()[:10]
        print(f"{Cin}->{Cout} @{H}x{W} rep {rep}: y {hy}  dx {hx}", flush=True)

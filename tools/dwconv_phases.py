"""Where do the cycles of the MFMA depthwise kernel go?  Loads libppea_dwprof.so (dwconv_mfma.hip built with
-DDW_PROF: s_memtime deltas per wave) and prints the per-phase averages for the bench shapes.  GPU box only.
  hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -DDW_PROF -shared \
        -o ppea-depth_amd/libppea_dwprof.so ppea-depth_amd/csrc/dwconv_mfma.hip"""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = ctypes.CDLL(os.path.join(ROOT, "ppea-depth_amd", "libppea_dwprof.so"))
vp, ci = ctypes.c_void_p, ctypes.c_int
lib.ppea_dwconv_lk_packed_bytes.restype = ctypes.c_long
dev = torch.device("cuda:0")
for (K, C, H, W) in [(31, 128, 48, 160), (29, 256, 24, 80), (27, 512, 12, 40), (13, 1024, 6, 20)]:
    N = 12
    x = torch.randn(N, C, H, W, device=dev).bfloat16()
    wb = (torch.randn(C, 1, K, K, device=dev) / K).float().contiguous()
    ws = (torch.randn(C, 1, 5, 5, device=dev) / 5).float().contiguous()
    pb = torch.empty(lib.ppea_dwconv_lk_packed_bytes(C, K), dtype=torch.uint8, device=dev)
    ps = torch.empty(lib.ppea_dwconv_lk_packed_bytes(C, 5), dtype=torch.uint8, device=dev)
    lib.ppea_dwconv_lk_pack_bf16(vp(wb.data_ptr()), vp(pb.data_ptr()), ci(C), ci(K), ci(0), None)
    lib.ppea_dwconv_lk_pack_bf16(vp(ws.data_ptr()), vp(ps.data_ptr()), ci(C), ci(5), ci(0), None)
    yb, ys = torch.empty_like(x), torch.empty_like(x)
    for _ in range(3):
        err = lib.ppea_dwconv_lk_fwd_bf16p(vp(x.data_ptr()), vp(pb.data_ptr()), vp(ps.data_ptr()), vp(yb.data_ptr()),
                                           vp(ys.data_ptr()), ci(N), ci(C), ci(H), ci(W), ci(K), ci(5), None)
        assert err == 0, err
    torch.cuda.synchronize()
    buf = np.zeros((4096, 8), dtype=np.uint64)
    assert lib.ppea_debug_dwconv_prof(buf.ctypes.data_as(vp)) == 0
    act = buf[buf[:, 5] > 0].astype(np.float64)
    m = act.mean(0)
    names = ["setup (filter image + fragments)", "staging per wave (all items)", "mac streams (bm: + epilogues)", "epilogues (bm: MFMA rows alone)",
             "items total", "kernel total", "  of setup: fragment construction"]
    print(f"k{K} [{N},{C},{H},{W}]: {len(act)} waves; shader-clock cycles per wave (s_memtime):")
    for i, nm in enumerate(names):
        print(f"   {nm:36s} {m[i]:10.0f} cycles   ({100 * m[i] / m[5]:5.1f} % of the wave)")

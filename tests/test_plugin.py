"""Boundary b3: the LARGE_KERNEL_CONV_IMPL plug-in `depthwise_conv2d_implicit_gemm`, loaded the way
the reference loads it (networks/replknet_adapter.py:151-168) from inside a process whose `ppeadepth` package is NOT
this build's (under `python -m ppeadepth.train` it is the reference's).  Every case runs in a child process so the
planted foreign `ppeadepth` never leaks into the test session."""
import os
import subprocess
import sys
import textwrap

import pytest

from conftest import PKG, ROOT

# replknet_adapter.py:151-168 restated literally (the branch taken when LARGE_KERNEL_CONV_IMPL is set), preceded by
# what `python -m ppeadepth.train` leaves in sys.modules: a package called `ppeadepth` that is not ours.
PROLOGUE = textwrap.dedent("""
    import os, sys, types
    foreign = types.ModuleType("ppeadepth")
    foreign.__file__ = "/somewhere/else/ppeadepth/__init__.py"
    foreign.__path__ = ["/somewhere/else/ppeadepth"]
    sys.modules["ppeadepth"] = foreign
    import torch, torch.nn as nn

    def get_conv2d(in_channels, out_channels, kernel_size, stride, padding, dilation, groups, bias):
        if type(kernel_size) is int:
            use_large_impl = kernel_size > 5
        else:
            assert len(kernel_size) == 2 and kernel_size[0] == kernel_size[1]
            use_large_impl = kernel_size[0] > 5
        has_large_impl = 'LARGE_KERNEL_CONV_IMPL' in os.environ
        if (has_large_impl and in_channels == out_channels and out_channels == groups and use_large_impl
                and stride == 1 and padding == kernel_size // 2 and dilation == 1):
            sys.path.append(os.environ['LARGE_KERNEL_CONV_IMPL'])
            from depthwise_conv2d_implicit_gemm import DepthWiseConv2dImplicitGEMM
            return DepthWiseConv2dImplicitGEMM(in_channels, kernel_size, bias=bias)
        return nn.Conv2d(in_channels, out_channels, kernel_size, stride, padding, dilation, groups, bias)

    def fuse_bn(conv, bn):                       # rka.py:199-208
        std = (bn.running_var + bn.eps).sqrt()
        return conv.weight * (bn.weight / std).reshape(-1, 1, 1, 1), bn.bias - bn.running_mean * bn.weight / std
""")


def _child(body, timeout=600):
    env = dict(os.environ, LARGE_KERNEL_CONV_IMPL=PKG)
    env.pop("PYTHONPATH", None)
    r = subprocess.run([sys.executable, "-c", PROLOGUE + textwrap.dedent(body)], env=env, cwd="/tmp",
                       capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    return r.stdout


def test_plugin_imports_next_to_a_foreign_ppeadepth_package():
    out = _child("""
        for bias in (False, True):
            m = get_conv2d(8, 8, 31, 1, 15, 1, 8, bias)
            assert type(m).__name__ == "DepthWiseConv2dImplicitGEMM" and isinstance(m, nn.Conv2d)
            assert tuple(m.weight.shape) == (8, 1, 31, 31) and (m.bias is not None) == bias
            assert list(m.state_dict().keys()) == (["weight", "bias"] if bias else ["weight"])
            # attributes merge_kernel (rka.py:250-261) and deep_fuse_BN (rka.py:563-580) read
            assert (m.in_channels, m.out_channels, m.groups) == (8, 8, 8)
            assert m.kernel_size == (31, 31) and m.stride == (1, 1) and m.padding == (15, 15) and m.dilation == (1, 1)
        assert isinstance(get_conv2d(8, 8, 5, 1, 2, 1, 8, False), nn.Conv2d)        # small kernels stay nn.Conv2d
        # deep_fuse_BN's reads + the conv it builds from them (tuple kernel_size/stride -> plain nn.Conv2d upstream)
        seq = nn.Sequential(get_conv2d(8, 8, 13, 1, 6, 1, 8, False), nn.BatchNorm2d(8))
        conv, bn = seq[0], seq[1]
        assert hasattr(conv, "kernel_size") and hasattr(conv, "weight")
        k, b = fuse_bn(conv, bn)
        fused = get_conv2d(conv.in_channels, conv.out_channels, kernel_size=conv.kernel_size, stride=conv.stride,
                           padding=conv.padding, dilation=conv.dilation, groups=conv.groups, bias=True)
        fused.weight.data, fused.bias.data = k, b
        assert tuple(fused.weight.shape) == (8, 1, 13, 13) and fused.padding == (6, 6)
        # the process's `ppeadepth` is still the foreign one; the kernels live under a private package name
        assert sys.modules["ppeadepth"] is foreign
        assert "_ppea_depth_amd_kernels.ops" in sys.modules and "ppeadepth.ops" not in sys.modules
        import depthwise_conv2d_implicit_gemm as plug
        # no CPU fallback behind the plug-in: a host tensor is refused loudly
        try:
            plug.DepthWiseConv2dImplicitGEMM(4, 7)(torch.zeros(1, 4, 8, 8))
        except Exception as e:
            assert type(e).__name__ == "PpeaKernelError", repr(e)
        else:
            raise AssertionError("CPU tensor was accepted")
        import ppea_kernels
        assert ppea_kernels.ops is plug.ops and hasattr(ppea_kernels.layers, "SSIM")
        print("ok")
    """)
    assert out.strip().endswith("ok")


@pytest.mark.skipif(not os.path.isdir("/root/reference/ppeadepth"), reason="reference tree only in the build container")
def test_plugin_through_the_unmodified_reference_get_conv2d():
    """The reference's own `get_conv2d` / `ReparamLargeKernelConv` with LARGE_KERNEL_CONV_IMPL set (CPU: construction,
    state_dict keys, merge_kernel; the forward needs a GPU and is covered by the -m gpu case below)."""
    env = dict(os.environ, LARGE_KERNEL_CONV_IMPL=PKG)
    code = textwrap.dedent(f"""
        import sys
        sys.path.insert(0, {ROOT!r})
        from oracle import ref_harness as rh
        rh.install_stubs()
        import ppeadepth
        assert ppeadepth.__file__.startswith("/root/reference/")
        from ppeadepth.networks import replknet_adapter as rka
        blk = rka.ReparamLargeKernelConv(8, 8, 13, 1, 8, small_kernel=5)
        assert type(blk.lkb_origin.conv).__name__ == "DepthWiseConv2dImplicitGEMM"
        assert type(blk.small_conv.conv).__name__ == "Conv2d"
        assert "lkb_origin.conv.weight" in blk.state_dict()
        blk.eval(); blk.merge_kernel()
        assert tuple(blk.lkb_reparam.weight.shape) == (8, 1, 13, 13) and blk.lkb_reparam.bias is not None
        print("ok")
    """)
    r = subprocess.run([sys.executable, "-c", code], env=env, cwd="/tmp", capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-4000:]
    assert r.stdout.strip().endswith("ok")


@pytest.mark.gpu
def test_plugin_forward_and_gradients_on_gpu_vs_oracle():
    """forward, x.grad, weight.grad, bias.grad of the plug-in module (fp32, and bf16 activations under the MFMA
    kernel) against the CPU oracle, in a process whose `ppeadepth` is foreign."""
    out = _child(f"""
        sys.path.insert(0, {ROOT!r})
        from oracle import ref_ops as R
        dev = torch.device("cuda:0")
        def rel(a, b):
            return float((a.double() - b.double()).abs().max() / (b.double().abs().max() + 1e-12))
        for K, (H, W) in ((31, (24, 40)), (13, (6, 20)), (7, (9, 11))):
            C, N = 6, 2
            g = torch.Generator().manual_seed(K)
            m = get_conv2d(C, C, K, 1, K // 2, 1, C, True)
            with torch.no_grad():
                m.weight.copy_(torch.randn(C, 1, K, K, generator=g) / K)
                m.bias.copy_(torch.randn(C, generator=g))
            w, b = m.weight.detach().clone(), m.bias.detach().clone()
            x = torch.randn(N, C, H, W, generator=g)
            gy = torch.randn(N, C, H, W, generator=g)
            xr, wr, br = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
            yr = R.dwconv(xr, wr) + br.view(1, -1, 1, 1)
            (yr * gy).sum().backward()
            m = m.to(dev)
            xd = x.to(dev).requires_grad_(True)
            y = m(xd)
            (y * gy.to(dev)).sum().backward()
            assert rel(y.cpu(), yr) < 2e-5, (K, rel(y.cpu(), yr))
            assert rel(xd.grad.cpu(), xr.grad) < 2e-4
            assert rel(m.weight.grad.cpu(), wr.grad) < 2e-4
            assert rel(m.bias.grad.cpu(), br.grad) < 2e-4
            if K in (31, 13):                                   # bf16 activations: the MFMA kernel
                xb = x.bfloat16().to(dev).requires_grad_(True)
                yb = m(xb)
                assert yb.dtype == torch.bfloat16
                ref = R.dwconv(x.bfloat16().float(), w.bfloat16().float()) + b.view(1, -1, 1, 1)
                assert rel(yb.float().cpu(), ref) < 2 ** -6
                yb.backward(gy.bfloat16().to(dev))
                xr2 = x.bfloat16().float().requires_grad_(True)
                (R.dwconv(xr2, w.bfloat16().float()) * gy.bfloat16().float()).sum().backward()
                assert rel(xb.grad.float().cpu(), xr2.grad) < 2 ** -6
        import ppea_kernels                                      # INTEGRATION.md section 2: reference-named layers
        a, t = torch.rand(2, 3, 24, 40, generator=g), torch.rand(2, 3, 24, 40, generator=g)
        s = ppea_kernels.layers.SSIM()(a.to(dev), t.to(dev))
        assert rel(s.cpu(), R.ssim(a, t)) < 2e-5
        assert sys.modules["ppeadepth"] is foreign
        print("ok")
    """)
    assert out.strip().endswith("ok")

"""`ppeadepth.layers` -- same public names as the reference module (layers.py), backed by the
HIP kernels of libppea_depth.so.  Signatures, argument meaning and return shapes follow the
reference (file:line cited per item, paths relative to /root/reference/ppeadepth/).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops


FUSE_BIAS_ELU = True


def disp_to_depth(disp, min_depth, max_depth):
    """layers.py:14-23."""
    min_disp = 1.0 / max_depth
    max_disp = 1.0 / min_depth
    scaled_disp = min_disp + (max_disp - min_disp) * disp
    return scaled_disp, 1.0 / scaled_disp


def rot_from_axisangle(vec):
    """layers.py:61-100 (Rodrigues; vec [B,1,3] -> [B,4,4]).  Built with one stack instead of nine
    scatter writes: ~10 small kernels instead of ~40."""
    angle = torch.norm(vec, 2, 2, True)
    axis = vec / (angle + 1e-7)
    ca, sa = torch.cos(angle)[:, 0, 0], torch.sin(angle)[:, 0, 0]
    C = 1 - ca
    x, y, z = axis[:, 0, 0], axis[:, 0, 1], axis[:, 0, 2]
    xC, yC, zC = x * C, y * C, z * C
    zero, one = torch.zeros_like(ca), torch.ones_like(ca)
    rows = [x * xC + ca, x * yC - z * sa, z * xC + y * sa, zero,
            x * yC + z * sa, y * yC + ca, y * zC - x * sa, zero,
            z * xC - y * sa, y * zC + x * sa, z * zC + ca, zero,
            zero, zero, zero, one]
    return torch.stack(rows, 1).reshape(-1, 4, 4)


def get_translation_matrix(translation_vector):
    """layers.py:45-58."""
    t = translation_vector.contiguous().view(-1, 3)
    T = torch.eye(4, device=t.device, dtype=t.dtype).repeat(t.shape[0], 1, 1)
    T = T.clone()
    T[:, :3, 3] = t
    return T


POSE_MATRIX_KERNEL = __import__("os").environ.get("PPEA_POSE_KERNEL", "1") == "1"


def transformation_from_parameters(axisangle, translation, invert=False):
    """layers.py:26-42.  On the device: one launch (csrc/geometry.hip pose_matrix_*) -- and fp32 throughout: under bf16
    autocast the composite below would run its two 4x4 matmuls in bf16."""
    if POSE_MATRIX_KERNEL and axisangle.is_cuda and axisangle.dim() == 3 and axisangle.shape[1] == 1:
        from . import ops
        return ops.pose_matrix(axisangle, translation, invert)
    R = rot_from_axisangle(axisangle)
    t = translation.clone()
    if invert:
        R = R.transpose(1, 2)
        t = t * -1
    T = get_translation_matrix(t)
    return torch.matmul(R, T) if invert else torch.matmul(T, R)


def upsample(x):
    """layers.py:204-207.  Nearest-neighbour copy: run it in the tensor's own dtype (autocast would widen a
    bf16 activation to fp32 here and every consumer down to the next conv would move twice the bytes)."""
    if x.is_cuda and x.dtype == torch.bfloat16:
        with torch.autocast("cuda", enabled=False):
            return F.interpolate(x, scale_factor=2, mode="nearest")
    return F.interpolate(x, scale_factor=2, mode="nearest")


def upsample_cat(x, skip=None):
    """depth_decoder_v2.py:231-236: `upsample(x)` followed, where the level has a skip connection, by
    `torch.cat([x, skip], 1)` -- on channels_last HIP tensors one fused pass (and one fused pass backward)."""
    if x.is_cuda:
        from . import ops
        if skip is not None and skip.dtype != x.dtype:
            skip = skip.to(x.dtype)
        if ops.up2cat_supported(x, skip):
            return ops.upsample2x_cat(x, skip)
    x = upsample(x)
    return x if skip is None else torch.cat([x, skip], 1)


class Conv3x3(nn.Module):
    """layers.py:119-135 (reflection- or zero-padded 3x3 conv)."""

    def __init__(self, in_channels, out_channels, use_refl=True):
        super().__init__()
        self.use_refl = use_refl
        self.pad = nn.ReflectionPad2d(1) if use_refl else nn.ZeroPad2d(1)
        self.conv = ops.Conv2d(int(in_channels), int(out_channels), 3)

    def forward(self, x, act="none"):
        if self.use_refl and x.is_cuda:
            y = ops.conv_module(self.conv, x, act, reflect=True)     # bf16 step: pad + conv + bias + act in one kernel
            if y is not None:
                return y
            y = self.conv(ops.reflect_pad1(x))         # gather-style pad kernels (no atomics in backward)
        else:
            y = self.conv(self.pad(x))
        return {"none": lambda t: t, "sigmoid": torch.sigmoid, "elu": F.elu}[act](y)


class ConvBlock(nn.Module):
    """layers.py:103-116 (Conv3x3 + ELU)."""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.conv = Conv3x3(in_channels, out_channels)
        self.nonlin = nn.ELU(inplace=True)

    def forward(self, x):
        c = self.conv
        if x.is_cuda and c.use_refl:
            y = ops.conv_module(c.conv, x, "elu", reflect=True)
            if y is not None:
                return y
        if FUSE_BIAS_ELU and x.is_cuda and c.use_refl and c.conv.bias is not None:
            # conv without bias, then bias + ELU in one pass whose backward also yields the bias gradient
            z = ops.conv2d(ops.reflect_pad1(x), c.conv.weight, None)
            if z.dtype in (torch.float32, torch.bfloat16):
                return ops.bias_elu(z, c.conv.bias)
            return self.nonlin(z + c.conv.bias.view(1, -1, 1, 1))
        return self.nonlin(self.conv(x))


class BackprojectDepth(nn.Module):
    """layers.py:138-168: depth [B,1,H,W], inv_K [B,4,4] -> homogeneous points [B,4,HW].
    Kept for API compatibility; the training step uses the fused `backproject_project` kernel and
    never materialises the point cloud."""

    def __init__(self, batch_size, height, width):
        super().__init__()
        self.batch_size, self.height, self.width = batch_size, height, width
        ys, xs = torch.meshgrid(torch.arange(height, dtype=torch.float32),
                                torch.arange(width, dtype=torch.float32), indexing="ij")
        pix = torch.stack([xs.reshape(-1), ys.reshape(-1), torch.ones(height * width)], 0)
        self.register_buffer("pix_coords", pix[None].repeat(batch_size, 1, 1), persistent=False)
        self.register_buffer("ones", torch.ones(batch_size, 1, height * width), persistent=False)

    def forward(self, depth, inv_K):
        cam_points = torch.matmul(inv_K[:, :3, :3], self.pix_coords)
        cam_points = depth.view(self.batch_size, 1, -1) * cam_points
        return torch.cat([cam_points, self.ones], 1)


class Project3D(nn.Module):
    """layers.py:171-199: points [B,4,HW], K, T -> sampling grid [B,H,W,2]."""

    def __init__(self, batch_size, height, width, dc=False, eps=1e-7):
        super().__init__()
        self.batch_size, self.height, self.width, self.eps, self.dc = batch_size, height, width, eps, dc

    def forward(self, points, K, T):
        P = torch.matmul(K, T)[:, :3, :]
        cam_points = torch.matmul(P, points)
        pix = cam_points[:, :2, :] / (cam_points[:, 2, :].unsqueeze(1) + self.eps)
        pix = pix.view(self.batch_size, 2, self.height, self.width).permute(0, 2, 3, 1)
        scale = pix.new_tensor([self.width - 1, self.height - 1])
        pix = (pix / scale - 0.5) * 2
        if self.dc:
            return pix, cam_points[:, 2, :].unsqueeze(1).view(self.batch_size, 1, self.height, self.width)
        return pix


def backproject_project(depth, inv_K, K, T, eps=1e-7):
    """BackprojectDepth followed by Project3D as ONE kernel (trainer.py:904-907)."""
    return ops.backproject_project(depth, inv_K, K, T, eps)


def get_smooth_loss(disp, img):
    """layers.py:210-223."""
    return ops.smooth_loss(disp, img)


class SSIM(nn.Module):
    """layers.py:226-257: per-pixel, per-channel SSIM loss map clamp((1 - SSIM) / 2, 0, 1).
    Returned as [B,C,H,W] like the reference; `Trainer.compute_reprojection_loss` uses the fused
    `ops.ssim_l1` instead (SSIM + L1 + channel mean in one pass)."""

    def forward(self, x, y):
        B, C, H, W = x.shape
        # alpha = 1 and one channel at a time turns the fused kernel into plain SSIM
        out = ops.ssim_l1(x.reshape(B * C, 1, H, W), y.reshape(B * C, 1, H, W), 1.0)
        return out.reshape(B, C, H, W)


def compute_depth_errors(gt, pred):
    """layers.py:260-278."""
    thresh = torch.max((gt / pred), (pred / gt))
    a1 = (thresh < 1.25).float().mean()
    a2 = (thresh < 1.25 ** 2).float().mean()
    a3 = (thresh < 1.25 ** 3).float().mean()
    rmse = torch.sqrt(((gt - pred) ** 2).mean())
    rmse_log = torch.sqrt(((torch.log(gt) - torch.log(pred)) ** 2).mean())
    abs_rel = torch.mean(torch.abs(gt - pred) / gt)
    sq_rel = torch.mean((gt - pred) ** 2 / gt)
    return abs_rel, sq_rel, rmse, rmse_log, a1, a2, a3

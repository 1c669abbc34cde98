"""TEST INFRASTRUCTURE ONLY -- CPU restatement (the parity oracle) of the PPEA-Depth
hot-path operators, SURVEY.md section 8(a) rows A1-A29.

Plain PyTorch fp32 on the CPU, written from the algorithm each reference function
implements (file:line cited per function; paths are relative to
/root/reference/ppeadepth).  Pinned against golden vectors produced by the
reference itself (`oracle/gen_golden.py` -> `tests/golden/*.npz`,
`tests/test_oracle_golden.py`).  Only tests/, `__graft_entry__.smoke()` and
bench.py's `cpu_baseline` leg may import this module; the product package
(`ppea-depth_amd/`) never does.
"""
import math

import torch
import torch.nn.functional as F

# ---------------------------------------------------------------------------
# A17 / A15: scalar geometry helpers
# ---------------------------------------------------------------------------


def disp_to_depth(disp, min_depth, max_depth):
    """layers.py:14-23 -- sigmoid output -> (scaled disparity, depth)."""
    lo, hi = 1.0 / max_depth, 1.0 / min_depth
    sd = lo + (hi - lo) * disp
    return sd, 1.0 / sd


def rot_from_axisangle(vec):
    """layers.py:61-100 -- Rodrigues, vec [B,1,3] -> [B,4,4]; angle = |v|, axis = v/(|v|+1e-7)."""
    B = vec.shape[0]
    angle = vec.norm(p=2, dim=2, keepdim=True)            # [B,1,1]
    axis = vec / (angle + 1e-7)
    ca, sa = torch.cos(angle)[:, 0, 0], torch.sin(angle)[:, 0, 0]
    C = 1 - ca
    x, y, z = axis[:, 0, 0], axis[:, 0, 1], axis[:, 0, 2]
    R = vec.new_zeros(B, 4, 4)
    # NB the reference forms x*(x*C), x*(y*C), z*(x*C) etc. in this association order.
    xC, yC, zC = x * C, y * C, z * C
    R[:, 0, 0] = x * xC + ca
    R[:, 0, 1] = x * yC - z * sa
    R[:, 0, 2] = z * xC + y * sa
    R[:, 1, 0] = x * yC + z * sa
    R[:, 1, 1] = y * yC + ca
    R[:, 1, 2] = y * zC - x * sa
    R[:, 2, 0] = z * xC - y * sa
    R[:, 2, 1] = y * zC + x * sa
    R[:, 2, 2] = z * zC + ca
    R[:, 3, 3] = 1
    return R


def transformation_from_parameters(axisangle, translation, invert=False):
    """layers.py:26-58 -- T*R, or (invert) R^T * T(-t)."""
    R = rot_from_axisangle(axisangle)
    t = translation.reshape(-1, 3)
    T = torch.eye(4, dtype=R.dtype).repeat(R.shape[0], 1, 1)
    if invert:
        T[:, :3, 3] = -t
        return R.transpose(1, 2) @ T
    T[:, :3, 3] = t
    return T @ R


# ---------------------------------------------------------------------------
# A18 / A19 / A20: backproject, project, warp
# ---------------------------------------------------------------------------


def backproject(depth, inv_K):
    """layers.py:138-168 -- depth [B,1,H,W], inv_K [B,4,4] -> homogeneous points [B,4,HW].
    Pixel grid is (x, y, 1) in row-major order (meshgrid indexing='xy')."""
    B, _, H, W = depth.shape
    ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float32),
                            torch.arange(W, dtype=torch.float32), indexing="ij")
    pix = torch.stack([xs.reshape(-1), ys.reshape(-1), torch.ones(H * W)], 0)   # [3,HW]
    rays = inv_K[:, :3, :3] @ pix[None].expand(B, 3, H * W)
    pts = depth.reshape(B, 1, H * W) * rays
    return torch.cat([pts, torch.ones(B, 1, H * W)], 1)


def project3d(points, K, T, H, W, eps=1e-7):
    """layers.py:171-199 -- points [B,4,HW] -> sampling grid [B,H,W,2] in [-1,1]."""
    B = points.shape[0]
    P = (K @ T)[:, :3, :]
    cam = P @ points
    xy = cam[:, :2, :] / (cam[:, 2:3, :] + eps)
    xy = xy.reshape(B, 2, H, W).permute(0, 2, 3, 1).clone()
    xy[..., 0] = xy[..., 0] / (W - 1)
    xy[..., 1] = xy[..., 1] / (H - 1)
    return (xy - 0.5) * 2


def grid_sample_border(src, grid):
    """trainer.py:911-914 -- bilinear, padding_mode='border', align_corners=True."""
    return F.grid_sample(src, grid, mode="bilinear", padding_mode="border", align_corners=True)


def grid_sample_zeros(src, grid):
    """replk_matching_adapter.py:299 -- bilinear, zeros padding, align_corners=True."""
    return F.grid_sample(src, grid, mode="bilinear", padding_mode="zeros", align_corners=True)


def grid_sample_manual(src, grid, border):
    """Gather-free restatement of ATen's grid_sampler_2d (bilinear, align_corners=True)
    used to cross-check F.grid_sample above: ix = (x+1)/2*(W-1); border mode clips the
    coordinate to [0, W-1] before taking floor; zeros mode drops out-of-range corners."""
    B, C, H, W = src.shape
    ix = (grid[..., 0] + 1) * 0.5 * (W - 1)
    iy = (grid[..., 1] + 1) * 0.5 * (H - 1)
    if border:
        ix = ix.clamp(0, W - 1)
        iy = iy.clamp(0, H - 1)
    x0, y0 = torch.floor(ix), torch.floor(iy)
    fx, fy = ix - x0, iy - y0
    out = src.new_zeros(B, C, *grid.shape[1:3])
    flat = src.reshape(B, C, H * W)
    for dy, wy in ((0, 1 - fy), (1, fy)):
        for dx, wx in ((0, 1 - fx), (1, fx)):
            xi, yi = x0 + dx, y0 + dy
            ok = (xi >= 0) & (xi <= W - 1) & (yi >= 0) & (yi <= H - 1)
            idx = (yi.clamp(0, H - 1) * W + xi.clamp(0, W - 1)).long().reshape(B, 1, -1)
            v = torch.gather(flat, 2, idx.expand(B, C, -1)).reshape(out.shape)
            out = out + v * (wx * wy * ok)[:, None]
    return out


# ---------------------------------------------------------------------------
# A21 / A22 / A23: photometric terms
# ---------------------------------------------------------------------------


def ssim(x, y):
    """layers.py:226-257 -- reflection pad 1, 3x3 means, clamp((1 - n/d)/2, 0, 1)."""
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    xp = F.pad(x, (1, 1, 1, 1), mode="reflect")
    yp = F.pad(y, (1, 1, 1, 1), mode="reflect")

    def box(t):
        return F.avg_pool2d(t, 3, 1)

    mx, my = box(xp), box(yp)
    sx = box(xp * xp) - mx * mx
    sy = box(yp * yp) - my * my
    sxy = box(xp * yp) - mx * my
    n = (2 * mx * my + C1) * (2 * sxy + C2)
    d = (mx * mx + my * my + C1) * (sx + sy + C2)
    return torch.clamp((1 - n / d) / 2, 0, 1)


def reprojection_loss(pred, target, no_ssim=False):
    """trainer.py:995-1007 -- 0.85*mean_C(SSIM(pred,target)) + 0.15*mean_C|target-pred|."""
    l1 = (target - pred).abs().mean(1, keepdim=True)
    if no_ssim:
        return l1
    return 0.85 * ssim(pred, target).mean(1, keepdim=True) + 0.15 * l1


def smooth_loss(disp, img):
    """layers.py:210-223 -- edge-aware first-order smoothness."""
    dx = (disp[..., :, :-1] - disp[..., :, 1:]).abs()
    dy = (disp[..., :-1, :] - disp[..., 1:, :]).abs()
    ix = (img[..., :, :-1] - img[..., :, 1:]).abs().mean(1, keepdim=True)
    iy = (img[..., :-1, :] - img[..., 1:, :]).abs().mean(1, keepdim=True)
    return (dx * torch.exp(-ix)).mean() + (dy * torch.exp(-iy)).mean()


def normalised_smooth_loss(disp, img):
    """trainer.py:1147-1149 -- disp / (mean_HW(disp) + 1e-7) then smooth_loss."""
    m = disp.mean(2, keepdim=True).mean(3, keepdim=True)
    return smooth_loss(disp / (m + 1e-7), img)


def automask(reproj, identity):
    """trainer.py:1009-1027 -- argmin over cat([reproj, identity]) (first minimum wins).
    Returns (int64 index [B,1,H,W], float mask idx==0)."""
    idx = torch.argmin(torch.cat([reproj, identity], 1), dim=1, keepdim=True)
    return idx, (idx == 0).float()


def select_reprojection(reproj_pair, color_m1, color_p1):
    """trainer.py:1076-1083 -- per-pixel min over the two source frames (+ int64 frame
    index), then the `selec_reproj` overwrite: where one warped frame is (near) black
    (sum_C < 0.1) take the other frame's loss; where both are, 0."""
    val, idx = torch.min(reproj_pair, dim=1, keepdim=True)
    m1 = (color_m1.sum(1, keepdim=True) < 0.1)
    p1 = (color_p1.sum(1, keepdim=True) < 0.1)
    val = torch.where(m1, reproj_pair[:, 1:2], val)
    val = torch.where(p1, reproj_pair[:, 0:1], val)
    val = torch.where(m1 & p1, torch.zeros_like(val), val)
    return val, idx


def matching_mask(lowest_cost, mono_depth):
    """trainer.py:859-869 -- where cost-volume depth and teacher depth agree within 2x."""
    md = 1.0 / lowest_cost.unsqueeze(1)
    m = ((md - mono_depth) / mono_depth) < 1.0
    m = m & (((mono_depth - md) / md) < 1.0)
    return m[:, 0]


# ---------------------------------------------------------------------------
# A1 / A2: large-kernel depthwise conv + batch norm
# ---------------------------------------------------------------------------


def dwconv(x, w):
    """get_conv2d replknet_adapter.py:151-168 -- depthwise k x k, stride 1, pad k//2."""
    return F.conv2d(x, w, None, 1, w.shape[-1] // 2, 1, x.shape[1])


def dwconv_loops(x, w):
    """Tap-by-tap restatement of the same depthwise correlation (no conv primitive):
    y[n,c,i,j] = sum_{u,v} w[c,0,u,v] * x[n,c,i+u-p,j+v-p], zero outside."""
    k = w.shape[-1]
    p = k // 2
    H, W = x.shape[-2:]
    xp = F.pad(x.double(), (p, p, p, p))
    y = torch.zeros_like(x, dtype=torch.float64)
    for u in range(k):
        for v in range(k):
            y += w[:, 0, u, v].double()[None, :, None, None] * xp[:, :, u:u + H, v:v + W]
    return y.float()


def bn_train(x, gamma, beta, eps=1e-5):
    """nn.BatchNorm2d / SyncBatchNorm forward in training mode (get_bn
    replknet_adapter.py:176-180): per-channel mean and *biased* variance over (N,H,W).
    Returns (y, mean, biased_var)."""
    mean = x.mean((0, 2, 3))
    var = x.var((0, 2, 3), unbiased=False)
    y = (x - mean[None, :, None, None]) * torch.rsqrt(var + eps)[None, :, None, None]
    return y * gamma[None, :, None, None] + beta[None, :, None, None], mean, var


def bn_running_update(running_mean, running_var, mean, biased_var, count, momentum=0.1):
    """running stats use the *unbiased* variance (torch semantics)."""
    unb = biased_var * (count / max(count - 1, 1))
    return ((1 - momentum) * running_mean + momentum * mean,
            (1 - momentum) * running_var + momentum * unb)


def reparam_lk(x, w_big, g_big, b_big, w_small, g_small, b_small):
    """ReparamLargeKernelConv.forward replknet_adapter.py:232-239 (train mode):
    BN_a(DW_k(x)) + BN_b(DW_5(x))."""
    ya, _, _ = bn_train(dwconv(x, w_big), g_big, b_big)
    yb, _, _ = bn_train(dwconv(x, w_small), g_small, b_small)
    return ya + yb


# ---------------------------------------------------------------------------
# A3 / A4: adapters
# ---------------------------------------------------------------------------


def b_adapter(x, w1, b1, w2, b2):
    """B_Adapter adpt_test=4, replknet_adapter.py:59-62, 87-109:
    Linear_{C/4->C}(GELU(Conv3x3_{C->C/4}(x)))."""
    B, C, H, W = x.shape
    h = F.conv2d(x, w1, b1, padding=1)
    h = F.gelu(h.flatten(2).transpose(1, 2))
    return F.linear(h, w2, b2).transpose(1, 2).reshape(B, -1, H, W)


def mlp_adapter(x, w1, b1, w2, b2):
    """Adapter replknet_adapter.py:20-47: Linear(GELU(Linear(x^T)))."""
    B, C, H, W = x.shape
    h = F.gelu(F.linear(x.flatten(2).transpose(1, 2), w1, b1))
    return F.linear(h, w2, b2).transpose(1, 2).reshape(B, -1, H, W)


# ---------------------------------------------------------------------------
# A8 / A9 / A10: plane-sweep cost volume
# ---------------------------------------------------------------------------


def depth_bins_log(min_depth, max_depth, num_bins=96):
    """compute_depth_bins 'log' branch, replk_matching_adapter.py:146-151:
    exp(log(min) + log(max/min) * i / n), i = 0..n-1 (max excluded), fp32 arithmetic."""
    mn = torch.as_tensor(min_depth, dtype=torch.float32).reshape(())
    mx = torch.as_tensor(max_depth, dtype=torch.float32).reshape(())
    base = torch.log(mn)
    it = torch.log(mx / mn)
    return torch.exp(torch.stack([base + it * i / num_bins for i in range(num_bins)]))


def cost_volume(cur, lookup, poses, K, inv_K, bins):
    """match_features replk_matching_adapter.py:261-340 for one lookup frame set.
    cur [B,C,h,w]; lookup [B,F,C,h,w]; poses [B,F,4,4]; K, inv_K [B,4,4] (scale 2);
    bins [D].  Returns (cost [B,D,h,w], missing mask [B,D,h,w])."""
    B, C, h, w = cur.shape
    D = bins.shape[0]
    costs, masks = [], []
    for b in range(B):
        vol = torch.zeros(D, h, w)
        cnt = torch.zeros(D, h, w)
        planes = bins.reshape(D, 1, 1, 1).expand(D, 1, h, w)
        pts = backproject(planes, inv_K[b:b + 1].expand(D, 4, 4))
        for f in range(lookup.shape[1]):
            pose = poses[b, f]
            if float(pose.sum()) == 0.0:       # zeroed pose == missing frame (:294)
                continue
            grid = project3d(pts, K[b:b + 1].expand(D, 4, 4), pose[None].expand(D, 4, 4), h, w)
            warped = grid_sample_zeros(lookup[b, f][None].expand(D, C, h, w), grid)
            xv = (grid[..., 0] / 2 + 0.5) * (w - 1)
            yv = (grid[..., 1] / 2 + 0.5) * (h - 1)
            edge = ((xv >= 2.0) & (xv <= w - 2) & (yv >= 2.0) & (yv <= h - 2)).float()
            inner = torch.zeros(D, h, w)
            inner[:, 2:-2, 2:-2] = 1.0
            diff = (warped - cur[b:b + 1]).abs().mean(1) * (edge * inner)
            vol = vol + diff
            cnt = cnt + (diff > 0).float()
        vol = vol / (cnt + 1e-7)
        miss = (vol == 0).float()
        vol = vol * (1 - miss) + vol.max(0)[0][None] * miss       # set_missing_to_max
        costs.append(vol)
        masks.append(miss)
    return torch.stack(costs), torch.stack(masks)


def cost_volume_reduce(cost, missing, bins):
    """replk_matching_adapter.py:446-456, 372-387: confidence (all bins observed),
    argmin over bins after 0 -> 100 (int64 indices, first minimum), 1/depth of the
    winning bin, and the confidence-masked volume fed to reduce_conv."""
    D = cost.shape[1]
    conf = (((cost * (1 - missing)) > 0).sum(1) == D).float()
    viz = cost.clone()
    viz[viz == 0] = 100
    _, idx = torch.min(viz, 1)
    lowest = 1.0 / bins[idx.reshape(-1)].reshape(idx.shape)
    return conf, idx, lowest, cost * conf.unsqueeze(1)


# ---------------------------------------------------------------------------
# A29: adaptive depth-bin tracker
# ---------------------------------------------------------------------------


class DepthBinTracker:
    """DepthBins, trainer.py:41-69 -- EMA(0.99) of 0.9*mean per-image min and
    1.1*mean per-image max of the teacher depth; lower bound opt.min_depth."""

    def __init__(self, opt_min_depth):
        self.min_depth = torch.tensor(0.1)
        self.max_depth = torch.tensor(10.0)
        self.opt_min_depth = opt_min_depth
        self.updated = False

    def update(self, mono_depth):
        self.updated = True
        d = mono_depth.detach()
        mn = d.amin((-1, -2)).mean()
        mx = d.amax((-1, -2)).mean()
        mn = torch.clamp(mn * 0.9, min=self.opt_min_depth)
        mx = mx * 1.1
        self.max_depth = self.max_depth * 0.99 + mx * 0.01
        self.min_depth = self.min_depth * 0.99 + mn * 0.01

    def compute(self):
        return self.min_depth.float(), self.max_depth.float()


# ---------------------------------------------------------------------------
# misc
# ---------------------------------------------------------------------------


def drop_path_mask(batch, drop_prob, generator=None):
    """timm DropPath (scale_by_keep): bernoulli(keep)/keep per sample."""
    keep = 1.0 - drop_prob
    m = torch.empty(batch, 1, 1, 1).bernoulli_(keep, generator=generator)
    return m / keep if keep > 0 else m


def gelu_exact(x):
    return 0.5 * x * (1.0 + torch.erf(x / math.sqrt(2.0)))


# ---------------------------------------------------------------------------------------------
# depth metrics: evaluate_depth.py:35-54 and the per-image protocol of Trainer.val (trainer.py:780-835)
# ---------------------------------------------------------------------------------------------
def compute_errors(gt, pred):
    """gt, pred: 1-D float tensors of valid pixels -> (abs_rel, sq_rel, rmse, rmse_log, a1, a2, a3)."""
    gt, pred = gt.double(), pred.double()
    thresh = torch.maximum(gt / pred, pred / gt)
    a = [(thresh < 1.25 ** k).double().mean() for k in (1, 2, 3)]
    rmse = ((gt - pred) ** 2).mean().sqrt()
    rmse_log = ((gt.log() - pred.log()) ** 2).mean().sqrt()
    abs_rel = ((gt - pred).abs() / gt).mean()
    sq_rel = ((gt - pred) ** 2 / gt).mean()
    return torch.stack([abs_rel, sq_rel, rmse, rmse_log] + a)


def evaluate_image(pred_disp, gt_depth, min_val=1e-3, max_val=80.0):
    """pred_disp [h,w], gt_depth [H,W] (0 = no return), eval_split "eigen": bilinear resize of the disparity to the
    ground-truth size (cv2.resize INTER_LINEAR), Garg/Eigen crop, median scaling, clamp."""
    H, W = gt_depth.shape
    pd = 1 / F.interpolate(pred_disp[None, None].float(), (H, W), mode="bilinear", align_corners=False)[0, 0]
    mask = (gt_depth > min_val) & (gt_depth < max_val)
    crop = torch.zeros_like(mask)
    y0, y1, x0, x1 = int(0.40810811 * H), int(0.99189189 * H), int(0.03594771 * W), int(0.96405229 * W)
    crop[y0:y1, x0:x1] = True
    mask = mask & crop
    pd, gt = pd[mask], gt_depth[mask]
    pd = pd * (gt.median() if gt.numel() % 2 else _np_median(gt)) / (pd.median() if pd.numel() % 2 else _np_median(pd))
    return compute_errors(gt, pd.clamp(min_val, max_val))


def _np_median(t):
    """numpy's median of an even-length array is the mean of the two middle values (torch takes the lower one)."""
    s = t.sort()[0]
    n = s.numel()
    return (s[n // 2 - 1] + s[n // 2]) / 2

"""Multi-frame matching encoder (reference: networks/replk_matching_adapter.py).

`RepLKMatchingAdapter.forward(current_image, lookup_images, poses, K, invK, min_depth_bin,
max_depth_bin) -> (features, lowest_cost, confidence_mask)` as in the reference (rkm.py:389-476).
Execution differences: the 96-plane sweep is ONE fused kernel for the whole batch (no per-item
Python loop, no [96,C,h,w] repeat, no BackprojectDepth/Project3D modules rebuilt per call); depth
bins are built with one vectorised expression; the argmin -> 1/depth lookup stays on the device.
"""
import numpy as np
import torch
import torch.nn as nn

from .. import ops
from .replknet_adapter import create_RepLKNet31B_Adapter, create_RepLKNet31L_Adapter


class RepLKMatchingAdapter(nn.Module):
    def __init__(self, rep_size, use_checkpoint, trans_adpt, input_adpt, adpt_test, g_blk, g_ffn, ratio,
                 min_depth_bin=0.1, max_depth_bin=20.0, num_depth_bins=96, adaptive_bins=False,
                 depth_binning="linear"):
        super().__init__()
        if rep_size == "b":
            path, factory = "./pretrained/RepLKNet-31B_ImageNet-1K_224.pth", create_RepLKNet31B_Adapter
            self.num_ch_enc = np.array([128, 256, 512, 1024])
        elif rep_size == "l":
            path, factory = "./pretrained/RepLKNet-31L_ImageNet-22K.pth", create_RepLKNet31L_Adapter
            self.num_ch_enc = np.array([192, 384, 768, 1536])
        else:
            raise NotImplementedError
        self.replk = factory(drop_path_rate=0.3, num_classes=None, out_indices=(0, 1, 2, 3),
                             use_checkpoint=use_checkpoint, small_kernel_merged=False, pretrained=path,
                             use_sync_bn=True, g_blk=g_blk, g_ffn=g_ffn, ratio=ratio, trans_adpt=trans_adpt,
                             input_adpt=input_adpt, adpt_test=adpt_test)
        self.trans_adpt, self.input_adpt = trans_adpt, input_adpt
        self.adaptive_bins, self.depth_binning = adaptive_bins, depth_binning
        self.set_missing_to_max = True
        self.num_depth_bins = num_depth_bins
        self.depth_bins = None
        self.reduce_conv = nn.Sequential(
            ops.Conv2d(int(self.num_ch_enc[0]) + num_depth_bins, int(self.num_ch_enc[0]), 3, 1, 1),
            nn.ReLU(inplace=True))

    # -- rkm.py:134-161 -------------------------------------------------------------------------
    def compute_depth_bins(self, min_depth_bin, max_depth_bin, device=None):
        n = self.num_depth_bins
        mn = torch.as_tensor(min_depth_bin, dtype=torch.float32, device=device).reshape(())
        mx = torch.as_tensor(max_depth_bin, dtype=torch.float32, device=device).reshape(())
        i = torch.arange(n, dtype=torch.float32, device=mn.device)
        if self.depth_binning == "log":
            # exp(log(min) + log(max/min) * i / n): same association as the reference's list build
            bins = torch.exp(torch.log(mn) + torch.log(mx / mn) * i / n)
        elif self.depth_binning == "linear":
            bins = mn + (mx - mn) * i / (n - 1)
        elif self.depth_binning == "inverse":
            inv = 1 / mx + (1 / mn - 1 / mx) * i / (n - 1)
            bins = (1 / inv).flip(0)
        else:
            raise NotImplementedError
        self.depth_bins = bins
        return bins

    # -- rkm.py:261-340 -------------------------------------------------------------------------
    def match_features(self, current_feats, lookup_feats, relative_poses, K, invK):
        """current_feats [B,C,h,w]; lookup_feats [B,F,C,h,w]; relative_poses [B,F,4,4] -> cost volume
        and missing mask, both [B,D,h,w] (after the missing -> max fill, as the reference returns)."""
        if lookup_feats.shape[1] != 1:
            raise NotImplementedError("hot path uses one lookup frame (num_matching_frames=1)")
        raw = ops.cost_volume(current_feats, lookup_feats[:, 0], relative_poses[:, 0], K, invK,
                              self.depth_bins)
        missing = (raw == 0).float()
        filled = raw * (1 - missing) + raw.max(1, keepdim=True)[0] * missing
        return filled, missing

    def feature_extraction(self, image, return_all_feats=False):
        """rkm.py:342-369: stem + stage 0."""
        self.replk.plan_drop_masks(image, stages=(0,))
        x = self.replk.stem_forward(image)
        x = self.replk.stages[0](x)
        return x, [self.replk.stages[0].norm(x)]

    def indices_to_disparity(self, indices):
        return 1 / self.depth_bins[indices]

    def compute_confidence_mask(self, cost_volume, num_bins_threshold=None):
        if num_bins_threshold is None:
            num_bins_threshold = self.num_depth_bins
        return ((cost_volume > 0).sum(1) == num_bins_threshold).float()

    # -- rkm.py:389-476 -------------------------------------------------------------------------
    def forward(self, current_image, lookup_images, poses, K, invK, min_depth_bin=None, max_depth_bin=None):
        self.compute_depth_bins(min_depth_bin, max_depth_bin, current_image.device)
        current_feats, self.features = self.feature_extraction(current_image)
        with torch.no_grad():
            B, Fr, ch, H, W = lookup_images.shape
            if Fr != 1:
                raise NotImplementedError("hot path uses one lookup frame (num_matching_frames=1)")
            lookup_feats, _ = self.feature_extraction(lookup_images.reshape(B * Fr, ch, H, W))
            if callable(poses):
                poses = poses()
            raw = ops.cost_volume(self.features[-1], lookup_feats, poses[:, 0], K, invK,
                                  self.depth_bins)
            cost_volume, confidence_mask, argmin, lowest_cost = ops.cost_volume_reduce(raw, self.depth_bins)
        self.argmin = argmin
        feat = self.features[-1]
        cat = torch.cat([feat, cost_volume.to(feat.dtype)], 1)
        x = ops.conv_module(self.reduce_conv[0], cat, "relu", out_nchw=True) if cat.is_cuda else None   # bias + ReLU fused
        if x is None:
            x = self.reduce_conv(cat)
        self.replk.plan_drop_masks(x, stages=range(1, self.replk.num_stages))
        x = self.replk.transition_forward(0, x)
        for s in range(1, self.replk.num_stages):
            x = self.replk.stages[s](x)
            if s in self.replk.out_indices:
                self.features.append(self.replk.stages[s].norm(x))
            if s < self.replk.num_stages - 1:
                x = self.replk.transition_forward(s, x)
        return self.features, lowest_cost, confidence_mask

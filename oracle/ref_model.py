"""TEST INFRASTRUCTURE ONLY -- functional CPU restatement of the whole PPEA-Depth
training step: RepLKNet-31B/L +
adapters, matching encoder with plane-sweep cost volume, depth / pose decoders and
`Trainer.process_batch`.  Driven by a plain `state_dict` whose keys are the
reference's, so the very same tensors can be loaded into the product
model and into this oracle.  Pinned by tests/golden/e2e_*.npz (produced by the
reference's unmodified process_batch, oracle/gen_golden.py).

Used only by tests/, smoke() and bench.py's cpu_baseline leg.
File:line citations are into /root/reference/ppeadepth/.
"""
import random

import torch
import torch.nn.functional as F

from . import ref_ops as R

LK_SIZES = (31, 29, 27, 13)          # networks/replknet_adapter.py:630-638
LAYERS = (2, 2, 18, 2)
CHANNELS = {"b": (128, 256, 512, 1024), "l": (192, 384, 768, 1536)}
DROP_PATH_RATE = 0.3                  # replk_matching_adapter.py:62, repdepth.py:95


def trainable(name: str, opt) -> bool:
    """Freeze rule by parameter-name substring (repdepth.py:47-50, 121-124), Stage-1."""
    top = name.split(".")[0]
    if getattr(opt, "dc", False) and top in ("depth", "mono_depth"):
        return "adpt" in name or "adapter" in name        # dc_ft_init, repdepth.py:255-262
    if top == "encoder":
        return any(s in name for s in ("adpt", "adapter", "reduce", "bn"))
    if top == "mono_encoder":
        return any(s in name for s in ("adpt", "adapter", "bn"))
    if top == "pose_encoder" and ".fc." in name:
        return False                  # resnet_encoder.py:390-392
    return True


class RefRepDepth:
    """Functional restatement of networks.RepDepth (repdepth.py) in training mode."""

    def __init__(self, state_dict, opt):
        self.opt = opt
        self.sd = state_dict
        self.ch = CHANNELS[opt.rep_size]
        n = sum(LAYERS)
        # dpr: torch.linspace(0, rate, sum(layers)) (replknet_adapter.py:425)
        self.dpr = [x.item() for x in torch.linspace(0, DROP_PATH_RATE, n)]
        self.bn_updates = 1     # 2 inside checkpointed segments (reentrant recompute)
        self.training = True    # False: model.eval() -- BN on running statistics, DropPath off (Trainer.val)

    # ----- primitives -------------------------------------------------------
    def _bn(self, x, p, twice=False):
        sd = self.sd
        if not self.training:
            return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"],
                                sd[p + ".bias"], False, 0.1, 1e-5)
        y = F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"],
                         sd[p + ".bias"], True, 0.1, 1e-5)
        if twice and torch.is_grad_enabled() and self.opt.use_checkpoint:
            # reentrant checkpoint re-runs the block in backward: a second, identical
            # running-stat update (replknet_adapter.py:364-368, 515-519)
            with torch.no_grad():
                F.batch_norm(x.detach(), sd[p + ".running_mean"], sd[p + ".running_var"], None,
                             None, True, 0.1, 1e-5)
        if p + ".num_batches_tracked" in sd:
            sd[p + ".num_batches_tracked"] += 2 if (twice and torch.is_grad_enabled()
                                                    and self.opt.use_checkpoint) else 1
        return y

    def _conv_bn(self, x, p, stride=1, pad=0, groups=1, relu=False, ck=False):
        y = F.conv2d(x, self.sd[p + ".conv.weight"], None, stride, pad, 1, groups)
        y = self._bn(y, p + ".bn", twice=ck)
        return F.relu(y) if relu else y

    def _drop_path(self, x, p):
        if p == 0.0 or not self.training:
            return x
        return x * R.drop_path_mask(x.shape[0], p)

    # ----- RepLKNet blocks (replknet_adapter.py:264-326) ----------------------
    def _replk_block(self, x, p, k, dp):
        sd = self.sd
        C = x.shape[1]
        out = self._bn(x, p + ".prelkb_bn", twice=True)
        adpt = R.b_adapter(out, sd[p + ".adapter.D_fc1.weight"], sd[p + ".adapter.D_fc1.bias"],
                           sd[p + ".adapter.D_fc2.weight"], sd[p + ".adapter.D_fc2.bias"])
        out = self._conv_bn(out, p + ".pw1", relu=True, ck=True)
        big = self._bn(R.dwconv(out, sd[p + ".large_kernel.lkb_origin.conv.weight"]),
                       p + ".large_kernel.lkb_origin.bn", twice=True)
        small = self._bn(R.dwconv(out, sd[p + ".large_kernel.small_conv.conv.weight"]),
                         p + ".large_kernel.small_conv.bn", twice=True)
        out = F.relu(big + small)
        out = self._conv_bn(out, p + ".pw2", ck=True)
        return x + self._drop_path(out, dp) + self.opt.g_blk * adpt

    def _conv_ffn(self, x, p, dp):
        sd = self.sd
        out = self._bn(x, p + ".preffn_bn", twice=True)
        adpt = R.mlp_adapter(out, sd[p + ".mlp_adapter.D_fc1.weight"], sd[p + ".mlp_adapter.D_fc1.bias"],
                             sd[p + ".mlp_adapter.D_fc2.weight"], sd[p + ".mlp_adapter.D_fc2.bias"])
        out = self._conv_bn(out, p + ".pw1", ck=True)
        out = F.gelu(out)
        out = self._conv_bn(out, p + ".pw2", ck=True)
        return x + self._drop_path(out, dp) + self.opt.g_ffn * adpt

    def _stage(self, x, p, s):
        first = sum(LAYERS[:s])
        for j in range(2 * LAYERS[s]):
            dp = self.dpr[first + j // 2]
            bp = f"{p}.stages.{s}.blocks.{j}"
            x = self._replk_block(x, bp, LK_SIZES[s], dp) if j % 2 == 0 else self._conv_ffn(x, bp, dp)
        return x

    def _stem(self, x, p):
        C = self.ch[0]
        x = self._conv_bn(x, p + ".stem.0", 2, 1, 1, relu=True)                  # not checkpointed
        x = self._conv_bn(x, p + ".stem.1", 1, 1, C, relu=True, ck=True)
        x = self._conv_bn(x, p + ".stem.2", 1, 0, 1, relu=True, ck=True)
        return self._conv_bn(x, p + ".stem.3", 2, 1, C, relu=True, ck=True)

    def _transition(self, x, p, s):
        C = self.ch[s + 1]
        x = self._conv_bn(x, f"{p}.transitions.{s}.0", relu=True)
        return self._conv_bn(x, f"{p}.transitions.{s}.1", 2, 1, C, relu=True)

    def mono_encoder(self, img):
        """RepLKNetAdapter.forward_features with out_indices (replknet_adapter.py:511-542)."""
        p = "mono_encoder"
        x = self._stem(img, p)
        feats = []
        for s in range(4):
            x = self._stage(x, p, s)
            feats.append(x)
            if s < 3:
                x = self._transition(x, p, s)
        return feats

    # ----- matching encoder (replk_matching_adapter.py:389-476) ---------------
    def matching_encoder(self, cur_img, lookup_imgs, poses, K, inv_K, min_bin, max_bin):
        p = "encoder.replk"
        sd = self.sd
        bins = R.depth_bins_log(min_bin, max_bin, self.opt.num_depth_bins)
        cur = self._stage(self._stem(cur_img, p), p, 0)
        with torch.no_grad():
            B, Fr = lookup_imgs.shape[:2]
            look = self._stage(self._stem(lookup_imgs.flatten(0, 1), p), p, 0)
            look = look.reshape(B, Fr, *look.shape[1:])
            cost, missing = R.cost_volume(cur, look, poses, K, inv_K, bins)
            self.debug = {"cur": cur.detach(), "look": look, "cost_filled": cost, "bins": bins, "poses": poses}
            conf, idx, lowest, cost = R.cost_volume_reduce(cost, missing, bins)
        x = F.relu(F.conv2d(torch.cat([cur, cost], 1), sd["encoder.reduce_conv.0.weight"],
                            sd["encoder.reduce_conv.0.bias"], padding=1))
        feats = [cur]
        x = self._transition(x, p, 0)
        for s in range(1, 4):
            x = self._stage(x, p, s)
            feats.append(x)
            if s < 3:
                x = self._transition(x, p, s)
        return feats, lowest, conf, idx

    # ----- depth decoder (depth_decoder_v2.py:172-245, dc=False) ---------------
    def _conv3x3(self, x, p):
        return F.conv2d(F.pad(x, (1, 1, 1, 1), mode="reflect"), self.sd[p + ".conv.weight"],
                        self.sd[p + ".conv.bias"])

    def depth_decoder(self, feats, p):
        up = lambda t: F.interpolate(t, scale_factor=2, mode="nearest")  # noqa: E731
        x = feats[-1]
        adpt = None
        if getattr(self.opt, "dc", False):
            # Stage-2 decoder adapter, design 1 (depth_decoder_v2.py:19-55, 178-182): token-wise
            # Linear -> GELU -> Linear over cat(feat0, nearest x8 of feat3), then ConvTranspose2d(3, s2, p1, op1)
            sd = self.sd
            t = torch.cat([feats[0], F.interpolate(x, scale_factor=8, mode="nearest")], 1)
            B, C, H, W = t.shape
            t = t.flatten(2).permute(0, 2, 1)
            t = F.linear(F.gelu(F.linear(t, sd[p + ".adapter.D_fc1.weight"], sd[p + ".adapter.D_fc1.bias"])),
                         sd[p + ".adapter.D_fc2.weight"], sd[p + ".adapter.D_fc2.bias"])
            t = t.permute(0, 2, 1).reshape(B, -1, H, W)
            adpt = F.conv_transpose2d(t, sd[p + ".deconv_adpt.weight"], sd[p + ".deconv_adpt.bias"], stride=2,
                                      padding=1, output_padding=1)
        for i in range(4):
            x = F.elu(self._conv3x3(x, f"{p}.upconvs_0.{i}.conv"))
            x = up(x)
            if i < 3:
                x = torch.cat([x, feats[2 - i]], 1)
            x = F.elu(self._conv3x3(x, f"{p}.upconvs_1.{i}.conv"))
        x = up(F.elu(self._conv3x3(x, f"{p}.upconvs_0.4.conv")))
        x = F.elu(self._conv3x3(x, f"{p}.upconvs_1.4.conv"))
        if adpt is not None:                                  # depth_decoder_v2.py:230-233
            x = x + F.interpolate(adpt, scale_factor=2)
        return torch.sigmoid(self._conv3x3(x, f"{p}.disp_convs.0"))

    # ----- pose network (resnet_encoder.py:397-409, pose_decoder.py:33-52) -----
    def _rbn(self, x, p):
        return self._bn(x, p)

    def _basic_block(self, x, p, stride, down):
        sd = self.sd
        out = F.relu(self._rbn(F.conv2d(x, sd[p + ".conv1.weight"], None, stride, 1), p + ".bn1"))
        out = self._rbn(F.conv2d(out, sd[p + ".conv2.weight"], None, 1, 1), p + ".bn2")
        if down:
            x = self._rbn(F.conv2d(x, sd[p + ".downsample.0.weight"], None, stride), p + ".downsample.1")
        return F.relu(out + x)

    def pose_net(self, pair):
        sd, p = self.sd, "pose_encoder.encoder"
        x = (pair - 0.45) / 0.225
        x = F.relu(self._rbn(F.conv2d(x, sd[p + ".conv1.weight"], None, 2, 3), p + ".bn1"))
        x = F.max_pool2d(x, 3, 2, 1)
        for li, stride in ((1, 1), (2, 2), (3, 2), (4, 2)):
            x = self._basic_block(x, f"{p}.layer{li}.0", stride, li > 1)
            x = self._basic_block(x, f"{p}.layer{li}.1", 1, False)
        x = F.relu(F.conv2d(x, sd["pose.net.0.weight"], sd["pose.net.0.bias"]))
        x = F.relu(F.conv2d(x, sd["pose.net.1.weight"], sd["pose.net.1.bias"], 1, 1))
        x = F.relu(F.conv2d(x, sd["pose.net.2.weight"], sd["pose.net.2.bias"], 1, 1))
        x = F.conv2d(x, sd["pose.net.3.weight"], sd["pose.net.3.bias"])
        out = 0.01 * x.mean(3).mean(2).reshape(-1, 2, 1, 6)
        return out[..., :3], out[..., 3:]

    def predict_poses(self, inputs):
        """repdepth.py:443-509 with frame_ids [0,-1,1], matching_ids [0,-1]."""
        outputs = {}
        f = {i: inputs[("color_aug", i, 0)] for i in (0, -1, 1)}
        for fi in (-1, 1):
            pair = torch.cat([f[fi], f[0]] if fi < 0 else [f[0], f[fi]], 1)
            aa, tt = self.pose_net(pair)
            outputs[("axisangle", 0, fi)] = aa
            outputs[("translation", 0, fi)] = tt
            outputs[("cam_T_cam", 0, fi)] = R.transformation_from_parameters(
                aa[:, 0], tt[:, 0], invert=(fi < 0))
        with torch.no_grad():
            aa, tt = self.pose_net(torch.cat([f[-1], f[0]], 1))
            pose = R.transformation_from_parameters(aa[:, 0], tt[:, 0], invert=True)
            empty = (f[-1].flatten(1).sum(1) == 0)
            pose = pose * (~empty).float()[:, None, None]
            inputs[("relative_pose", -1)] = pose
        return outputs

    # ----- inference path of Trainer.val (trainer.py:676-752), model.eval() ------------------
    @torch.no_grad()
    def predict_val(self, data, min_bin, max_bin, max_depth=100.0):
        """-> (scaled multi-frame disparity [B,H,W], scaled teacher disparity [B,H,W])."""
        assert not self.training
        c0, cm1 = data[("color", 0, 0)], data[("color", -1, 0)]
        aa, tt = self.pose_net(torch.cat([cm1, c0], 1))
        pose = R.transformation_from_parameters(aa[:, 0], tt[:, 0], invert=True)
        data[("relative_pose", -1)] = pose
        feats, _lowest, _conf, _idx = self.matching_encoder(c0, cm1[:, None], pose[:, None], data[("K", 2)],
                                                            data[("inv_K", 2)], min_bin, max_bin)
        disp, _ = R.disp_to_depth(self.depth_decoder(feats, "depth"), 1e-3, 80)
        mono, _ = R.disp_to_depth(self.depth_decoder(self.mono_encoder(c0), "mono_depth"), 1e-3, max_depth)
        return disp[:, 0], mono[:, 0]

    # ----- RepDepth.forward (repdepth.py:529-624) ------------------------------
    def forward(self, inputs, min_bin, max_bin):
        opt = self.opt
        mono_outputs, outputs = {}, {}
        pose_pred = self.predict_poses(inputs)
        outputs.update(pose_pred)
        mono_outputs.update(pose_pred)
        rel = torch.stack([inputs[("relative_pose", -1)]], 1).clone()
        look = torch.stack([inputs[("color_aug", -1, 0)]], 1).clone()
        B = look.shape[0]
        aug = torch.zeros(B, 1, 1, 1)
        for b in range(B):
            r = random.random()
            if r < 0.25:                       # static-camera augmentation
                look[b, 0] = inputs[("color", 0, 0)][b]
                aug[b] += 1
            elif r < 0.5:                      # missing cost volume augmentation
                rel[b] *= 0
                aug[b] += 1
        outputs["augmentation_mask"] = aug
        mono_outputs[("disp", 0)] = self.depth_decoder(self.mono_encoder(inputs[("color_aug", 0, 0)]),
                                                       "mono_depth")
        outputs[("mono_disp", 0)] = mono_outputs[("disp", 0)]
        feats, lowest, conf, idx = self.matching_encoder(
            inputs[("color_aug", 0, 0)], look, rel, inputs[("K", 2)], inputs[("inv_K", 2)],
            min_bin, max_bin)
        outputs[("disp", 0)] = self.depth_decoder(feats, "depth")
        size = [opt.height, opt.width]
        outputs["lowest_cost"] = F.interpolate(lowest[:, None], size, mode="nearest")[:, 0]
        outputs["consistency_mask"] = F.interpolate(conf[:, None], size, mode="nearest")[:, 0]
        outputs["argmin_bins"] = idx
        return mono_outputs, outputs


class RefTrainer:
    """Trainer.process_batch and what it calls (trainer.py:420-472, 871-919, 1032-1160)."""

    def __init__(self, model: RefRepDepth, opt):
        self.model, self.opt = model, opt
        self.bins = R.DepthBinTracker(opt.min_depth)

    def generate_images_pred(self, inputs, outputs, is_multi):
        opt = self.opt
        disp = F.interpolate(outputs[("disp", 0)], [opt.height, opt.width], mode="bilinear",
                             align_corners=False)
        _, depth = R.disp_to_depth(disp, opt.min_depth, opt.max_depth)
        outputs[("depth", 0, 0)] = depth
        for fi in (-1, 1):
            T = outputs[("cam_T_cam", 0, fi)]
            if is_multi:
                T = T.detach()
            pts = R.backproject(depth, inputs[("inv_K", 0)])
            grid = R.project3d(pts, inputs[("K", 0)], T, opt.height, opt.width)
            outputs[("sample", fi, 0)] = grid
            outputs[("color", fi, 0)] = R.grid_sample_border(inputs[("color", fi, 0)], grid)
            outputs[("color_identity", fi, 0)] = inputs[("color", fi, 0)]

    def compute_losses(self, inputs, outputs, is_multi):
        opt = self.opt
        losses = {}
        target = inputs[("color", 0, 0)]
        disp = outputs[("disp", 0)]
        rp = torch.cat([R.reprojection_loss(outputs[("color", fi, 0)], target) for fi in (-1, 1)], 1)
        idl = torch.cat([R.reprojection_loss(inputs[("color", fi, 0)], target) for fi in (-1, 1)], 1)
        identity = idl.min(1, keepdim=True)[0]
        reproj, _ = R.select_reprojection(rp, outputs[("color", -1, 0)].detach(),
                                          outputs[("color", 1, 0)].detach())
        identity = identity + torch.randn(identity.shape) * 0.00001      # tie-break noise (:1086)
        _, mask = R.automask(reproj, identity)
        if is_multi:
            mask = torch.ones_like(mask) * outputs["consistency_mask"].unsqueeze(1)
            mask = mask * (1 - outputs["augmentation_mask"])
            cons_mask = (1 - mask).float()
        rl = (reproj * mask).sum() / (mask.sum() + 1e-7)
        cl = 0
        if is_multi:
            multi_d = outputs[("depth", 0, 0)]
            mono_d = outputs[("mono_depth", 0, 0)].detach()
            cl = ((multi_d - mono_d).abs() * cons_mask).mean()
            outputs["consistency_target/0"] = 1 / (mono_d * cons_mask + multi_d.detach() * (1 - cons_mask))
            losses["consistency_loss/0"] = cl
        losses["reproj_loss/0"] = rl
        loss = rl + cl + opt.disparity_smoothness * R.normalised_smooth_loss(disp, target)
        losses["loss/0"] = loss
        losses["loss"] = loss
        return losses

    def process_batch(self, inputs):
        if self.bins.updated:
            mn, mx = self.bins.compute()
        else:
            mn, mx = torch.Tensor([self.bins.min_depth]), torch.Tensor([self.bins.max_depth])
        mono_outputs, outputs = self.model.forward(inputs, mn, mx)
        self.generate_images_pred(inputs, mono_outputs, False)
        mono_losses = self.compute_losses(inputs, mono_outputs, False)
        outputs[("mono_depth", 0, 0)] = mono_outputs[("depth", 0, 0)]
        outputs[("mono_disp", 0)] = mono_outputs[("disp", 0)]
        outputs["consistency_mask"] = outputs["consistency_mask"] * R.matching_mask(
            outputs["lowest_cost"], outputs[("mono_depth", 0, 0)])
        self.generate_images_pred(inputs, outputs, True)
        losses = self.compute_losses(inputs, outputs, True)
        for k, v in mono_losses.items():
            losses[k] = losses[k] + v
        self.bins.update(outputs[("mono_depth", 0, 0)])
        return outputs, losses


def leaf_state_dict(sd, opt):
    """Clone a state_dict into autograd leaves following the freeze rule."""
    out = {}
    for k, v in sd.items():
        t = v.detach().clone()
        if t.is_floating_point() and "running_" not in k and trainable(k, opt):
            t.requires_grad_(True)
        out[k] = t
    return out

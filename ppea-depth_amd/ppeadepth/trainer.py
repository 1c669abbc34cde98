"""Training step (reference: trainer.py).  `Trainer.process_batch(inputs, is_train)` keeps the
reference signature and the `outputs` / `losses` dictionaries (SURVEY.md 8(a) output contract).

What changed on the way to the MI355X:
  * BackprojectDepth + Project3D, grid_sample, SSIM + L1 + channel mean, the per-pixel
    min / selec_reproj / automask block and the smoothness term are single HIP kernels;
  * the identity reprojection loss depends on the inputs only, so it is computed once per step
    and shared by the teacher and the multi-frame pass (the reference computes it twice);
  * boolean-mask assignments (dynamic shapes, host syncs) became branch-free selects;
  * tie-break noise is drawn on the device unless `rng.set_mode("reference")`;
  * data-parallel gradient exchange is one flat RCCL all-reduce (see dist.py).
"""
import contextlib
import os

import torch
import torch.nn.functional as F

from . import ops, rng
from .layers import disp_to_depth


MIN_VAL_EVAL = 1e-3          # trainer.py:657


class DepthBins:
    """trainer.py:41-69 -- EMA(0.99) tracker of the teacher's depth range.  The state is two persistent
    0-dim device tensors that are only ever updated IN PLACE: a captured step graph reads and writes the
    same storage on every replay, so the adaptive bins keep moving under hipGraph replay exactly as they do
    in eager steps.  With several ranks `compute()` reduces min/max like torchmetrics does
    (dist_reduce_fx = "min" / "max") without writing the reduced values back."""

    def __init__(self, opt_min_depth, device="cpu"):
        self.min_depth = torch.tensor(0.1, device=device)
        self.max_depth = torch.tensor(10.0, device=device)
        self.opt_min_depth = opt_min_depth
        self.updated = False

    def to(self, device):
        self.min_depth, self.max_depth = self.min_depth.to(device), self.max_depth.to(device)
        return self

    @torch.no_grad()
    def update(self, mono_depth):
        self.updated = True
        d = mono_depth.detach().float()
        mn = d.amin((-1, -2)).mean()
        mx = d.amax((-1, -2)).mean()
        mn = torch.clamp(mn * 0.9, min=self.opt_min_depth)
        mx = mx * 1.1
        # same association as the reference (state * 0.99 + new * 0.01), written into the persistent tensors
        self.max_depth.mul_(0.99).add_(mx * 0.01)
        self.min_depth.mul_(0.99).add_(mn * 0.01)

    @torch.no_grad()
    def load(self, min_depth, max_depth):
        self.min_depth.copy_(torch.as_tensor(min_depth).reshape(()))
        self.max_depth.copy_(torch.as_tensor(max_depth).reshape(()))
        self.updated = True

    def compute(self):
        mn, mx = self.min_depth.float(), self.max_depth.float()
        import torch.distributed as dist
        from .dist import collectives_on, log_collective
        if collectives_on():
            pack = torch.stack([-mn.reshape(()), mx.reshape(())])
            log_collective("all_reduce_max", pack)
            dist.all_reduce(pack, op=dist.ReduceOp.MAX)          # min via max of the negation
            mn, mx = -pack[0], pack[1]
        return mn, mx


MONO_LOSS_SIDE = os.environ.get("PPEA_MONO_SIDE", "1") == "1"   # the teacher's loss on the side stream, beside the student's
FUSED_LOSS_TAIL = True      # compute_losses' mask / masked mean / consistency term on ops.loss_tail (HIP tensors)


class _NullAccelerator:
    """The slice of `accelerate.Accelerator` that process_batch touches (trainer.py:436, 453, 465)."""

    def __init__(self, device, autocast_dtype=None):
        self.device = torch.device(device)
        self.is_main_process = True
        self._dtype = autocast_dtype

    def autocast(self):
        return contextlib.nullcontext()

    def wait_for_everyone(self):
        return None


class Trainer:
    def __init__(self, options, model, device, acc=None, amp_dtype=None):
        self.opt = options
        assert self.opt.height % 32 == 0 and self.opt.width % 32 == 0
        assert self.opt.frame_ids[0] == 0 and len(self.opt.frame_ids) > 1
        self.device = torch.device(device)
        self.acc = acc if acc is not None else _NullAccelerator(self.device)
        self.model = model
        self.amp_dtype = amp_dtype
        self.depth_bin_tracker = DepthBins(self.opt.min_depth, self.device)
        self.freeze_tp = getattr(self.opt, "freeze_teacher_and_pose", False)
        self.step = 0
        self.is_main = True
        self._identity_cache = None

    def _module(self):
        return getattr(self.model, "module", self.model)

    # ---- trainer.py:420-472 -----------------------------------------------------------------------
    def process_batch(self, inputs, is_train=False):
        tracker = self.depth_bin_tracker
        if getattr(self.opt, "notadabins", False):
            min_depth, max_depth = tracker.min_depth, tracker.max_depth
        else:
            # trainer.py:425-431 reads the raw state before the first update and compute() afterwards.  Before the
            # first update the state is the same constant on every rank, so compute() returns the raw state there
            # too: no host-side `updated` branch, which a captured step graph would freeze at capture time.
            min_depth, max_depth = tracker.compute()
        self._identity_cache = None
        module = self._module()
        module.defer_flush_join = True           # joined below, after the losses
        try:
            if self.amp_dtype is not None:
                with torch.autocast("cuda", dtype=self.amp_dtype):
                    mono_outputs, outputs = self.model(inputs, min_depth, max_depth)
            else:
                mono_outputs, outputs = self.model(inputs, min_depth, max_depth)
        finally:
            module.defer_flush_join = False

        side = module.pending_flush if (MONO_LOSS_SIDE and not self.freeze_tp) else None
        if side is not None:
            # The teacher's loss needs nothing of the student and the student's loss needs only the teacher's depth and the
            # identity terms: those two are computed here, the rest of the teacher's loss goes to the side stream (idle at this
            # point of the step, already forked for the running-statistics update) and is joined where the two losses are summed.
            main = torch.cuda.current_stream()
            self.generate_images_pred(inputs, mono_outputs, depth_only=True)
            self._identity_losses(inputs)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                self.generate_images_pred(inputs, mono_outputs, skip_depth=True)
                mono_losses, _ = self.compute_losses(inputs, mono_outputs, is_multi=False)
                for v in mono_losses.values():
                    if torch.is_tensor(v):
                        v.record_stream(main)
        else:
            self.generate_images_pred(inputs, mono_outputs)
            mono_losses, _ = self.compute_losses(inputs, mono_outputs, is_multi=False)

        for key in list(mono_outputs.keys()):
            if key[0] in ("depth", "disp"):
                outputs[("mono_" + key[0],) + tuple(key[1:])] = mono_outputs[key]
        outputs["consistency_mask"] = outputs["consistency_mask"] * self.compute_matching_mask(outputs)

        self.generate_images_pred(inputs, outputs, is_multi=True)
        losses, _ = self.compute_losses(inputs, outputs, is_multi=True)

        if side is not None:
            module.join_flush()
        if not self.freeze_tp:
            for key, val in mono_losses.items():
                losses[key] = losses[key] + val
        if not self.freeze_tp and not getattr(self.opt, "notadabins", False):
            self.acc.wait_for_everyone()
            tracker.update(outputs[("mono_depth", 0, 0)])
        module.join_flush()
        return outputs, losses

    # ---- trainer.py:653-857 -----------------------------------------------------------------------
    @torch.no_grad()
    def predict_disps(self, data, mono=True):
        """Inference path of `val` for one batch (trainer.py:676-752): pose of the lookup frame from the pose network,
        cost volume + multi-frame encoder + decoder, optionally the single-frame teacher; model in eval mode.
        -> (scaled multi-frame disparity [B,H,W], scaled teacher disparity or None), as `disp_to_depth(., 1e-3, 80)`."""
        from .layers import transformation_from_parameters
        model = self._module()
        dev = self.device
        ctx = torch.autocast("cuda", dtype=self.amp_dtype) if self.amp_dtype is not None else contextlib.nullcontext()
        color = {f: data[("color", f, 0)].to(dev) for f in (0, -1)}
        with ctx:
            feats = [model.pose_encoder(torch.cat([color[-1], color[0]], 1))]
            axisangle, translation = model.pose(feats)
            pose = transformation_from_parameters(axisangle[:, 0].float(), translation[:, 0].float(), invert=True)
            data[("relative_pose", -1)] = pose
            tracker = self.depth_bin_tracker
            if getattr(self.opt, "notadabins", False):
                mn, mx = tracker.min_depth, tracker.max_depth
            else:
                mn, mx = tracker.compute()
            feats, _lowest, _conf = model.encoder(color[0], color[-1][:, None], pose[:, None],
                                                  data[("K", 2)].to(dev), data[("inv_K", 2)].to(dev), mn, mx)
            disp = model.depth(feats)[("disp", 0)].float()
            pred, _ = disp_to_depth(disp, MIN_VAL_EVAL, 80)
            pred_mono = None
            if mono:
                dm = model.mono_depth(model.mono_encoder(color[0]))[("disp", 0)].float()
                pred_mono, _ = disp_to_depth(dm, MIN_VAL_EVAL, self.opt.max_depth)
                pred_mono = pred_mono[:, 0]
        return pred[:, 0], pred_mono

    def val(self, batches, gt_depths, eval_split="eigen", hard_test_mono=False, median_scaling=None):
        """`Trainer.val` (trainer.py:653-857) over an iterable of row-P batches and the split's ground-truth depth
        maps (`gt_depths.npz["data"]`, trainer.py:766-767): mean of (abs_rel, sq_rel, rmse, rmse_log, a1, a2, a3) for
        the multi-frame network and, unless the teacher is frozen, for the teacher."""
        import numpy as np
        from . import evaluate
        model = self._module()
        was_training = model.training
        model.eval()
        mono_flag = (not self.freeze_tp) or hard_test_mono
        disps, disps_mono = [], []
        try:
            for data in batches:
                d, dm = self.predict_disps(data, mono_flag)
                disps.append(d.cpu().numpy())
                if mono_flag:
                    disps_mono.append(dm.cpu().numpy())
        finally:
            model.train(was_training)
        disps = np.concatenate(disps)
        # trainer.py:818-822: `--pred_depth_scale_factor`, `--disable_median_scaling` (the teacher is always median-scaled
        # and never takes the scale factor, trainer.py:838-842)
        scale = float(getattr(self.opt, "pred_depth_scale_factor", 1.0))
        if median_scaling is None:
            median_scaling = not getattr(self.opt, "disable_median_scaling", False)
        mean_errors = evaluate.evaluate_disps(disps, gt_depths, eval_split, median_scaling, scale)
        if mono_flag:
            return mean_errors, evaluate.evaluate_disps(np.concatenate(disps_mono), gt_depths, eval_split, True)
        return mean_errors

    # ---- trainer.py:859-869 -----------------------------------------------------------------------
    def compute_matching_mask(self, outputs):
        mono = outputs[("mono_depth", 0, 0)]
        matching = 1 / outputs["lowest_cost"].unsqueeze(1)
        mask = ((matching - mono) / mono) < 1.0
        mask = mask & (((mono - matching) / matching) < 1.0)
        return mask[:, 0]

    # ---- trainer.py:871-919 -----------------------------------------------------------------------
    def generate_images_pred(self, inputs, outputs, is_multi=False, depth_only=False, skip_depth=False):
        opt = self.opt
        for scale in range(opt.sclm + 1):
            if skip_depth:
                depth = outputs[("depth", 0, scale)]
            else:
                disp = outputs[("disp", scale)].float()
                if disp.shape[-2:] != (opt.height, opt.width):
                    disp = F.interpolate(disp, [opt.height, opt.width], mode="bilinear", align_corners=False)
                _, depth = disp_to_depth(disp, opt.min_depth, opt.max_depth)
                outputs[("depth", 0, scale)] = depth
            if depth_only:
                continue
            for frame_id in opt.frame_ids[1:]:
                T = outputs[("cam_T_cam", 0, frame_id)].float()
                if is_multi:
                    T = T.detach()
                grid = ops.backproject_project(depth, inputs[("inv_K", 0)], inputs[("K", 0)], T)
                outputs[("sample", frame_id, scale)] = grid
                outputs[("color", frame_id, scale)] = ops.grid_sample(inputs[("color", frame_id, 0)], grid,
                                                                      "border")
                if not opt.disable_automasking:
                    outputs[("color_identity", frame_id, scale)] = inputs[("color", frame_id, 0)]

    # ---- trainer.py:995-1007 ----------------------------------------------------------------------
    def compute_reprojection_loss(self, pred, target):
        if self.opt.no_ssim:
            return torch.abs(target - pred).mean(1, True)
        return ops.ssim_l1(pred, target, 0.85)

    @staticmethod
    def compute_loss_masks(reprojection_loss, identity_reprojection_loss):
        """trainer.py:1009-1027."""
        if identity_reprojection_loss is None:
            return torch.ones_like(reprojection_loss)
        idxs = torch.argmin(torch.cat([reprojection_loss, identity_reprojection_loss], 1), 1, keepdim=True)
        return (idxs == 0).float()

    def _identity_losses(self, inputs):
        if self._identity_cache is None:
            target = inputs[("color", 0, 0)]
            with torch.no_grad():
                self._identity_cache = torch.cat(
                    [self.compute_reprojection_loss(inputs[("color", f, 0)], target)
                     for f in self.opt.frame_ids[1:]], 1)
        return self._identity_cache

    # ---- trainer.py:1032-1160 ---------------------------------------------------------------------
    def compute_losses(self, inputs, outputs, is_multi=False):
        opt = self.opt
        losses, total_loss = {}, 0
        for scale in range(opt.sclm + 1):
            disp = outputs[("disp", scale)].float()
            color = inputs[("color", 0, scale)]
            target = inputs[("color", 0, 0)]
            fids = opt.frame_ids[1:]
            reprojection_losses = torch.cat(
                [self.compute_reprojection_loss(outputs[("color", f, scale)], target) for f in fids], 1)
            identity_losses = self._identity_losses(inputs)
            B, _, H, W = reprojection_losses.shape
            noise = None
            if not opt.disable_automasking:
                noise = rng.randn_like_cpu_order((B, 1, H, W), self.device) * 0.00001
            reprojection_loss, _src, frame_idxs, auto_idx = ops.loss_select(
                reprojection_losses, identity_losses, outputs[("color", fids[0], scale)].detach(),
                outputs[("color", fids[1], scale)].detach(), noise, opt.selec_reproj)
            outputs[("frame_idxs", scale)] = frame_idxs
            if not opt.disable_automasking:
                outputs[("automask_idxs", scale)] = auto_idx
            if FUSED_LOSS_TAIL and reprojection_loss.is_cuda:
                # mask, masked mean, consistency term and its target in one pass per direction (ops.loss_tail)
                cons = outputs["consistency_mask"] if (is_multi and not opt.disable_motion_masking) else None
                aug = outputs["augmentation_mask"][:B] if (is_multi and not opt.no_matching_augmentation) else None
                res = ops.loss_tail(reprojection_losses, reprojection_loss.detach(), _src,
                                    None if (opt.disable_automasking or is_multi) else auto_idx, cons, aug,
                                    outputs[("depth", 0, scale)] if is_multi else None,
                                    outputs[("mono_depth", 0, scale)] if is_multi else None, is_multi)
                rl, consistency_loss = res[0], (res[1] if is_multi else 0)
                if is_multi:
                    outputs["consistency_target/{}".format(scale)] = res[3]
                    losses["consistency_loss/{}".format(scale)] = consistency_loss
            else:
                if opt.disable_automasking:
                    mask = torch.ones_like(reprojection_loss)
                else:
                    mask = (auto_idx == 0).float()
                if is_multi:
                    mask = torch.ones_like(mask)
                    if not opt.disable_motion_masking:
                        mask = mask * outputs["consistency_mask"].unsqueeze(1)
                    if not opt.no_matching_augmentation:
                        mask = mask * (1 - outputs["augmentation_mask"][:B])
                    consistency_mask = (1 - mask).float()
                rl = (reprojection_loss * mask).sum() / (mask.sum() + 1e-7)
                if is_multi:
                    multi_depth = outputs[("depth", 0, scale)]
                    mono_depth = outputs[("mono_depth", 0, scale)].detach()
                    consistency_loss = (torch.abs(multi_depth - mono_depth) * consistency_mask).mean()
                    outputs["consistency_target/{}".format(scale)] = 1 / (
                        mono_depth * consistency_mask + multi_depth.detach() * (1 - consistency_mask))
                    losses["consistency_loss/{}".format(scale)] = consistency_loss
                else:
                    consistency_loss = 0
            losses["reproj_loss/{}".format(scale)] = rl
            loss = rl + consistency_loss
            mean_disp = disp.mean(2, True).mean(3, True)
            smooth = ops.smooth_loss(disp / (mean_disp + 1e-7), color)
            loss = loss + opt.disparity_smoothness * smooth / (2 ** scale)
            total_loss = total_loss + loss
            losses["loss/{}".format(scale)] = loss
        losses["loss"] = total_loss / (opt.sclm + 1)
        return losses, []

"""Model aggregate `RepDepth(opt)` (reference: networks/repdepth.py).

Sub-module names (`encoder`, `depth`, `mono_encoder`, `mono_depth`, `pose_encoder`, `pose`), the
freeze rule and `forward(inputs, min_depth_bin, max_depth_bin) -> (mono_outputs, outputs)` are the
reference's.  Host-side differences: no per-item `.sum() == 0` device syncs (done as tensor masks),
the matching augmentation is applied with batched index ops, and BN bookkeeping of the whole
forward is flushed in a handful of multi-tensor kernels.
"""
import os

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops, rng
from ..batchnorm import DeferredStats, set_deferred
from ..layers import transformation_from_parameters
from .depth_decoder_v2 import DepthDecoderV2
from .pose_decoder import PoseDecoder
from .replk_matching_adapter import RepLKMatchingAdapter
from .replknet_adapter import create_RepLKNet31B_Adapter, create_RepLKNet31L_Adapter
from .resnet_encoder import ResnetEncoder, replay_updates

_ENC_CH = {"b": [128, 256, 512, 1024], "l": [192, 384, 768, 1536]}


TWO_STREAMS = True
POSE_SIDE = os.environ.get("PPEA_POSE_SIDE", "1") == "1"     # pose network (forward and backward) on the adapter side stream
POSE_SIDE_INLINE0 = os.environ.get("PPEA_POSE_SIDE_INLINE0", "1") == "1"
ASYNC_FLUSH = os.environ.get("PPEA_ASYNC_FLUSH", "1") == "1"   # deferred running-statistics update beside the loss
BATCHED_POSES = True      # third (no_grad) pose pass replayed instead of recomputed (see _predict_poses_batched)
POSE_ONE_BATCH = True      # both pairs as one 2B batch with per-pair BN statistics (needs the fused NHWC BN path)


def _g(opt, name, default):
    return getattr(opt, name, default)


class RepDepth(nn.Module):
    def __init__(self, opt):
        super().__init__()
        self.opt = opt
        if not _g(opt, "adapter", True):
            raise NotImplementedError("only the --adapter path is in scope (README always passes it)")
        if opt.rep_size not in _ENC_CH:
            raise NotImplementedError(f"rep_size {opt.rep_size}")
        num_ch_enc = np.array(_ENC_CH[opt.rep_size])
        dc, dec_id = _g(opt, "dc", False), _g(opt, "dec_id", 1)

        # repdepth.py:26-29 (built first: switches the module-level sync-BN flag on for both encoders)
        self.encoder = RepLKMatchingAdapter(
            opt.rep_size, opt.use_checkpoint, _g(opt, "trans", False), _g(opt, "input", False), opt.adpt_test,
            g_blk=opt.g_blk, g_ffn=opt.g_ffn, ratio=opt.ratio, adaptive_bins=not _g(opt, "notadabins", False),
            min_depth_bin=0.1, max_depth_bin=20.0, depth_binning=opt.depth_binning,
            num_depth_bins=opt.num_depth_bins)
        self.depth = DepthDecoderV2(num_ch_enc, opt.scales, _g(opt, "debug", False), dc=dc, test_id=dec_id)
        self._apply_freeze_rule(self.encoder, ("adpt", "adapter", "reduce", "bn"))

        factory = create_RepLKNet31B_Adapter if opt.rep_size == "b" else create_RepLKNet31L_Adapter
        path = ("./pretrained/RepLKNet-31B_ImageNet-1K_224.pth" if opt.rep_size == "b"
                else "./pretrained/RepLKNet-31L_ImageNet-22K.pth")
        self.mono_encoder = factory(
            drop_path_rate=0.3, num_classes=None, out_indices=(0, 1, 2, 3), use_checkpoint=opt.use_checkpoint,
            small_kernel_merged=False, pretrained=path, use_sync_bn=False, g_blk=opt.g_blk, g_ffn=opt.g_ffn,
            ratio=opt.ratio, trans_adpt=_g(opt, "mono_trans", False), input_adpt=_g(opt, "mono_input", False),
            adpt_test=opt.adpt_test)
        self.mono_depth = DepthDecoderV2(num_ch_enc, opt.scales, _g(opt, "debug", False), dc=dc, test_id=dec_id)
        self._apply_freeze_rule(self.mono_encoder, ("adpt", "adapter", "bn"))

        if _g(opt, "lps2", False):
            for m in (self.encoder, self.mono_encoder):
                for p in m.parameters():
                    p.requires_grad = False

        if _g(opt, "pose_cnn", False):
            raise NotImplementedError("--pose_cnn is broken upstream (repdepth.py:538) and out of scope")
        self.pose_encoder = ResnetEncoder(18, opt.weights_init == "pretrained", num_input_images=2)
        self.need_pose_dec = True
        self.pose = PoseDecoder(self.pose_encoder.num_ch_enc, num_input_features=1, num_frames_to_predict_for=2)

        self.matching_ids = [0]
        if _g(opt, "use_future_frame", False):
            self.matching_ids.append(1)
        for idx in range(-1, -1 - _g(opt, "num_matching_frames", 1), -1):
            self.matching_ids.append(idx)
        self.freeze_tp = False
        self.freeze_pose = False
        self.dc = dc
        self._bn_book = DeferredStats()
        self.defer_flush_join = False      # set by a caller that calls join_flush() itself (Trainer.process_batch)
        self.pending_flush = None

    # repdepth.py:47-66, 121-140
    def _apply_freeze_rule(self, module, keep):
        opt = self.opt
        if _g(opt, "fullft_reb", False):
            return
        for name, p in module.named_parameters():
            if not any(k in name for k in keep):
                p.requires_grad = False
            if _g(opt, "dc", False) and "adapter" in name and _g(opt, "dec_id", 1) in (5, 6):
                last = ["3.blocks.3", "2.blocks.35", "1.blocks.3", "0.blocks.3"]
                if opt.dec_id == 6:
                    last += ["3.blocks.2", "2.blocks.34", "1.blocks.2", "0.blocks.2"]
                if not any(s in name for s in last):
                    p.requires_grad = False
        if _g(opt, "dec_only", False):
            for p in module.parameters():
                p.requires_grad = False

    # repdepth.py:175-262 (design 1 / dec_id 1, 5, 6)
    def dc_ft_init(self, adpt=True):
        if not adpt:
            return
        for dec in (self.depth, self.mono_depth):
            dec.add_decoder_adapter(self.opt.dec_id, mlp_ratio=_g(self.opt, "dec_ratio", 0.25))
            for name, p in dec.named_parameters():
                if "adpt" not in name and "adapter" not in name:
                    p.requires_grad = False
        self.dc = True

    def freeze_tp_net(self):
        for m in (self.mono_encoder, self.mono_depth, self.pose_encoder, self.pose):
            for p in m.parameters():
                p.requires_grad = False
        self.freeze_tp = True

    def freeze_pose_net(self):
        for m in (self.pose_encoder, self.pose):
            for p in m.parameters():
                p.requires_grad = False
        self.freeze_pose = True

    def print_num_param(self):
        for n in ("mono_encoder", "encoder", "pose_encoder", "depth", "mono_depth"):
            m = getattr(self, n)
            print(f"for {n} ", sum(p.numel() for p in m.parameters() if p.requires_grad),
                  sum(p.numel() for p in m.parameters()))

    # repdepth.py:443-509
    def _side_stream(self, device):
        st = getattr(self, "_side", None)
        if st is None or st.device != device:
            st = self._side = torch.cuda.Stream(device)
            from . import replknet_adapter
            # The teacher's adapters run in line on this stream: forking them again would put event edges between two
            # streams that are BOTH non-origin streams of the capture (teacher stream <-> its adapter stream), and HIP's
            # stream capture faults on those (round 1: "nested fork"; round 2: rc 139 twice with a pre-joined adapter
            # stream -- pre-joining changes where the stream enters the capture, not the side <-> adapter edges).
            replknet_adapter.NO_FORK_ON.add(st.cuda_stream)
        return st

    def predict_poses(self, inputs):
        if (BATCHED_POSES and self.training and list(self.opt.frame_ids) == [0, -1, 1]
                and list(self.matching_ids) == [0, -1]):
            return self._predict_poses_batched(inputs)
        return self._predict_poses_sequential(inputs)

    def _predict_poses_batched(self, inputs):
        """The reference runs the pose network three times per step: pairs (-1, 0) and (0, +1) with gradients, then
        (-1, 0) AGAIN without, for the matching frame (repdepth.py:443-509).  The third pass has the same input and
        the same weights as the first, so its outputs are the first pass's and the only thing it changes is one more
        BatchNorm running-statistics update with the first pass's batch statistics: that update is replayed
        (GroupBN.replay_update) and the pass itself is skipped.  POSE_ONE_BATCH additionally sends both pairs through
        the network as one 2B batch with per-sub-batch statistics (measured slower: the per-layer concatenations
        cost more than the halved launch count saves)."""
        outputs = {}
        f = {i: inputs[("color_aug", i, 0)] for i in (0, -1, 1)}
        B = f[0].shape[0]
        pairs = {-1: torch.cat([f[-1], f[0]], 1), 1: torch.cat([f[0], f[1]], 1)}
        if POSE_ONE_BATCH:
            aa2, tt2 = self.pose([self.pose_encoder(torch.cat([pairs[-1], pairs[1]], 0), groups=2, record=True)])
            res = {-1: (aa2[:B], tt2[:B]), 1: (aa2[B:], tt2[B:])}
            recorded = self.pose_encoder.recorded
        else:
            snap = self.pose_encoder.snapshot_running()
            res = {-1: self.pose([self.pose_encoder(pairs[-1])])}
            stats_a = self.pose_encoder.batch_stats_since(snap)
            res[1] = self.pose([self.pose_encoder(pairs[1])])
            recorded = None
        for f_i in (-1, 1):
            aa, tt = res[f_i]
            outputs[("axisangle", 0, f_i)] = aa
            outputs[("translation", 0, f_i)] = tt
            outputs[("cam_T_cam", 0, f_i)] = transformation_from_parameters(aa[:, 0], tt[:, 0], invert=(f_i < 0))
        with torch.no_grad():
            if recorded is not None:                                          # the no_grad pass on pair (-1, 0)
                replay_updates(recorded)
                self.pose_encoder.recorded = None
            else:
                self.pose_encoder.replay_pass(stats_a)
            pose = outputs[("cam_T_cam", 0, -1)].detach()
            present = (f[-1].flatten(1).sum(1) != 0).to(pose.dtype)
            inputs[("relative_pose", -1)] = pose * present[:, None, None]
        return outputs

    def _predict_poses_sequential(self, inputs):
        outputs = {}
        frames = {f: inputs[("color_aug", f, 0)] for f in self.opt.frame_ids}
        for f_i in self.opt.frame_ids[1:]:
            if f_i == "s":
                continue
            pair = [frames[f_i], frames[0]] if f_i < 0 else [frames[0], frames[f_i]]
            axisangle, translation = self.pose([self.pose_encoder(torch.cat(pair, 1))])
            outputs[("axisangle", 0, f_i)] = axisangle
            outputs[("translation", 0, f_i)] = translation
            outputs[("cam_T_cam", 0, f_i)] = transformation_from_parameters(
                axisangle[:, 0], translation[:, 0], invert=(f_i < 0))
        frames = {f: inputs[("color_aug", f, 0)] for f in self.matching_ids}
        with torch.no_grad():
            for fi in self.matching_ids[1:]:
                if fi < 0:
                    pair = [frames[fi], frames[fi + 1]]
                    aa, tt = self.pose([self.pose_encoder(torch.cat(pair, 1))])
                    pose = transformation_from_parameters(aa[:, 0], tt[:, 0], invert=True)
                    if fi != -1:
                        pose = torch.matmul(pose, inputs[("relative_pose", fi + 1)])
                else:
                    pair = [frames[fi - 1], frames[fi]]
                    aa, tt = self.pose([self.pose_encoder(torch.cat(pair, 1))])
                    pose = transformation_from_parameters(aa[:, 0], tt[:, 0], invert=False)
                    if fi != 1:
                        pose = torch.matmul(pose, inputs[("relative_pose", fi - 1)])
                # missing (all-zero) lookup image -> zero pose; as a mask, without a host sync per item
                present = (frames[fi].flatten(1).sum(1) != 0).to(pose.dtype)
                inputs[("relative_pose", fi)] = pose * present[:, None, None]
        return outputs

    # repdepth.py:529-624
    def forward(self, inputs, min_depth_bin, max_depth_bin):
        opt = self.opt
        mono_outputs, outputs = {}, {}
        book = self._bn_book if self.training else None
        set_deferred(book)
        # Teacher branch (mono encoder + decoder) and student branch (poses, matching encoder, cost volume,
        # multi-frame encoder + decoder) do not depend on each other before the loss: the teacher is enqueued
        # on a side stream, forked here and joined below.  The step is thousands of small kernels that cannot
        # fill 256 CUs one at a time; two independent chains (forward and, through autograd's stream
        # bookkeeping, backward; both captured as parallel branches of the step graph) overlap them.
        img_aug = inputs[("color_aug", 0, 0)]
        side = None
        if TWO_STREAMS and img_aug.is_cuda and self.training and not self.freeze_tp:
            side = self._side_stream(img_aug.device)
            side.wait_stream(torch.cuda.current_stream())
        try:
            if side is not None:
                with torch.cuda.stream(side):
                    mono_outputs.update(self.mono_depth(self.mono_encoder(img_aug)))
            pose_side = None
            if not self.freeze_tp and not self.freeze_pose:
                if POSE_SIDE and side is not None:
                    # The pose network needs the frames only and is needed by the cost volume only: it runs on the adapter
                    # side stream beside the student's stem / stage 0 (the stage-0 adapters queue behind it there) and is
                    # joined where the matching encoder asks for the relative poses.
                    main = torch.cuda.current_stream()
                    pose_side = ops.side_stream_of(main)
                    if POSE_SIDE_INLINE0:
                        for blk in self.encoder.replk.stages[0].blocks:     # (they would wait for the pose network there)
                            blk.inline_adapter = True
                    pose_side.wait_stream(main)
                    with torch.cuda.stream(pose_side):
                        pose_pred = self.predict_poses(inputs)
                else:
                    pose_pred = self.predict_poses(inputs)
            else:
                with torch.no_grad():
                    pose_pred = self.predict_poses(inputs)
            outputs.update(pose_pred)
            mono_outputs.update(pose_pred)

            lookup_frames = torch.stack([inputs[("color_aug", i, 0)] for i in self.matching_ids[1:]], 1)
            device = lookup_frames.device
            B = lookup_frames.shape[0]

            # matching augmentation (repdepth.py:559-575): one uniform draw per item on the host,
            # applied with two masked selects
            draws = rng.aug_draws(B, device)
            static = draws < 0.25
            nopose = (draws >= 0.25) & (draws < 0.5)
            cur = inputs[("color", 0, 0)]
            lookup_frames = torch.where(static[:, None, None, None, None], cur[:, None].expand_as(lookup_frames),
                                        lookup_frames)
            outputs["augmentation_mask"] = (static | nopose).float().reshape(B, 1, 1, 1)

            def relative_poses():
                # evaluated by the matching encoder right before the cost volume
                if pose_side is not None:
                    main = torch.cuda.current_stream()
                    main.wait_stream(pose_side)
                    for v in list(pose_pred.values()) + [inputs[("relative_pose", i)] for i in self.matching_ids[1:]]:
                        if torch.is_tensor(v) and v.is_cuda:
                            v.record_stream(main)
                rp = torch.stack([inputs[("relative_pose", i)] for i in self.matching_ids[1:]], 1)
                return rp * (~nopose).to(rp.dtype)[:, None, None, None]

            if side is not None:
                pass
            elif not self.freeze_tp:
                mono_outputs.update(self.mono_depth(self.mono_encoder(img_aug)))
            else:
                with torch.no_grad():
                    mono_outputs.update(self.mono_depth(self.mono_encoder(img_aug)))
            for key in list(mono_outputs.keys()):
                if key[0] in ("depth", "disp"):
                    outputs[("mono_" + key[0],) + tuple(key[1:])] = mono_outputs[key]

            features, lowest_cost, confidence_mask = self.encoder(
                img_aug, lookup_frames, relative_poses, inputs[("K", 2)], inputs[("inv_K", 2)],
                min_depth_bin=min_depth_bin, max_depth_bin=max_depth_bin)
            outputs.update(self.depth(features))
            size = [opt.height, opt.width]
            outputs["lowest_cost"] = F.interpolate(lowest_cost.unsqueeze(1), size, mode="nearest")[:, 0]
            outputs["consistency_mask"] = F.interpolate(confidence_mask.unsqueeze(1), size, mode="nearest")[:, 0]
            if side is not None:
                main = torch.cuda.current_stream()
                main.wait_stream(side)
                for v in mono_outputs.values():          # produced on the side stream, consumed on this one
                    if torch.is_tensor(v) and v.is_cuda:
                        v.record_stream(main)
        finally:
            set_deferred(None)
            if book is not None:
                main = torch.cuda.current_stream() if torch.cuda.is_available() else None
                if ASYNC_FLUSH and self.defer_flush_join and main is not None and next(self.parameters()).is_cuda:
                    # nothing reads the running statistics before the next step: the ~45 multi-tensor launches go to the
                    # side stream, beside the loss; the caller (Trainer.process_batch) joins it
                    side = ops.side_stream_of(main)
                    side.wait_stream(main)
                    with torch.cuda.stream(side):
                        book.flush()
                    self.pending_flush = side
                else:
                    book.flush()
        return mono_outputs, outputs

    def join_flush(self):
        """Order the current stream after a running-statistics update that `forward` left on the side stream."""
        side, self.pending_flush = self.pending_flush, None
        if side is not None:
            torch.cuda.current_stream().wait_stream(side)

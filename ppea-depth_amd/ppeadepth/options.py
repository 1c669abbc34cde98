"""Options that shape the hot path (reference: options.py; resolved defaults ckpt/models/opt.json).

`MonodepthOptions().parse()` keeps the reference's flag names and defaults for every flag the
training step reads; flags of out-of-scope subsystems (datasets, logging, evaluation CLIs,
semantic/panoptic heads) are not reproduced.
"""
import argparse


class MonodepthOptions:
    def __init__(self):
        p = argparse.ArgumentParser(description="PPEA-Depth (MI355X) hot-path options")
        a = p.add_argument
        a("--height", type=int, default=192)
        a("--width", type=int, default=640)
        a("--batch_size", type=int, default=12)
        a("--learning_rate", type=float, default=1e-4)
        a("--scheduler_step_size", type=int, default=15)
        a("--num_epochs", type=int, default=30)
        a("--frame_ids", nargs="+", type=int, default=[0, -1, 1])
        a("--scales", nargs="+", type=int, default=[0, 1, 2, 3])
        a("--min_depth", type=float, default=0.1)
        a("--max_depth", type=float, default=100.0)
        a("--disparity_smoothness", type=float, default=1e-3)
        a("--depth_binning", type=str, default="log", choices=["linear", "inverse", "log"])
        a("--num_depth_bins", type=int, default=96)
        a("--num_matching_frames", type=int, default=1)
        a("--weights_init", type=str, default="pretrained", choices=["pretrained", "scratch"])
        a("--rep_size", type=str, default="b", choices=["b", "l"])
        a("--sclm", type=int, default=0)
        a("--adpt_test", type=int, default=4)
        a("--ratio", type=float, default=0.25)
        a("--g_blk", type=float, default=1.0)
        a("--g_ffn", type=float, default=1.0)
        a("--dec_id", type=int, default=1)
        a("--dec_ratio", type=float, default=0.25)
        a("--pred_depth_scale_factor", type=float, default=1.0)
        a("--disable_median_scaling", action="store_true")
        for flag in ("adapter", "use_checkpoint", "dc", "notadabins", "freeze_teacher_and_pose", "freeze_pose",
                     "no_ssim", "disable_automasking", "disable_motion_masking", "no_matching_augmentation",
                     "v1_multiscale", "use_future_frame", "train_cs", "ddad", "fullft_reb", "dec_only", "lps2",
                     "trans", "input", "mono_trans", "mono_input", "pose_cnn", "loss_pct", "debug"):
            a("--" + flag, action="store_true")
        a("--selec_reproj", action="store_false")      # default True (options.py:428-430)
        # NOT a reference flag.  Accelerate's wrapped StepLR steps once per RANK per scheduler.step() (accelerate
        # scheduler.py:69-82), so the reference's learning rate drops every ceil(scheduler_step_size / ranks) epochs.
        # Preserved by default (a drop-in must train on the reference's schedule); --no_lr_quirk gives the schedule the
        # options describe.  TrainEngine.describe_schedule() prints the effective one.
        a("--no_lr_quirk", dest="lr_quirk", action="store_false")
        self.parser = p

    def parse(self, argv=None):
        self.options = self.parser.parse_args(argv)
        return self.options


def default_options(**overrides):
    """Stage-1 defaults of the README training command (`--adapter --use_checkpoint`) with
    `--weights_init scratch` (no network in this environment)."""
    opt = MonodepthOptions().parse(["--adapter", "--use_checkpoint", "--weights_init", "scratch"])
    for k, v in overrides.items():
        if not hasattr(opt, k):
            raise AttributeError(k)
        setattr(opt, k, v)
    return opt

"""Depth evaluation -- the AbsRel harness of SURVEY 8(f)-2.

Restates the reference's metric code: `compute_errors` (evaluate_depth.py:35-54) and the per-image protocol inside
`Trainer.val` (trainer.py:780-835: resize the predicted disparity to the ground-truth size, Eigen crop, validity mask,
median scaling, clamp to [1e-3, 80]).  `cv2.resize(..., INTER_LINEAR)` is bilinear with half-pixel centres and no
antialiasing, i.e. `F.interpolate(mode="bilinear", align_corners=False)`.  Host-side numpy like the reference: the
metric runs once per validation image on a few hundred thousand LiDAR points, not in the training step.
"""
import numpy as np
import torch
import torch.nn.functional as F

MIN_VAL, MAX_VAL = 1e-3, 80.0          # trainer.py:657-658

ERROR_NAMES = ("abs_rel", "sq_rel", "rmse", "rmse_log", "a1", "a2", "a3")


def compute_errors(gt, pred):
    """evaluate_depth.py:35-54 (numpy, same operation order)."""
    thresh = np.maximum((gt / pred), (pred / gt))
    a1 = (thresh < 1.25).mean()
    a2 = (thresh < 1.25 ** 2).mean()
    a3 = (thresh < 1.25 ** 3).mean()
    rmse = np.sqrt(((gt - pred) ** 2).mean())
    rmse_log = np.sqrt(((np.log(gt) - np.log(pred)) ** 2).mean())
    abs_rel = np.mean(np.abs(gt - pred) / gt)
    sq_rel = np.mean(((gt - pred) ** 2) / gt)
    return abs_rel, sq_rel, rmse, rmse_log, a1, a2, a3


def resize_linear(img, width, height):
    """cv2.resize(img, (width, height)) with the default INTER_LINEAR for a 2-D float array."""
    t = torch.from_numpy(np.ascontiguousarray(img, dtype=np.float32))[None, None]
    return F.interpolate(t, (height, width), mode="bilinear", align_corners=False)[0, 0].numpy()


def eigen_crop_mask(gt_depth):
    """trainer.py:804-811: valid LiDAR returns inside Garg/Eigen's crop."""
    gt_height, gt_width = gt_depth.shape[:2]
    mask = np.logical_and(gt_depth > MIN_VAL, gt_depth < MAX_VAL)
    crop = np.array([0.40810811 * gt_height, 0.99189189 * gt_height,
                     0.03594771 * gt_width, 0.96405229 * gt_width]).astype(np.int32)
    crop_mask = np.zeros(mask.shape)
    crop_mask[crop[0]:crop[1], crop[2]:crop[3]] = 1
    return np.logical_and(mask, crop_mask)


def evaluate_image(pred_disp, gt_depth, eval_split="eigen", median_scaling=True, pred_depth_scale_factor=1.0):
    """One validation image (trainer.py:780-835): `pred_disp` [h,w] is the scaled disparity of
    `disp_to_depth(disp, 1e-3, 80)`; returns (the 7 errors, median ratio or None)."""
    gt_height, gt_width = gt_depth.shape[:2]
    if eval_split == "cityscapes":
        # the bottom 25 % (ego car) is cut off the ground truth first -- the loader did the same to the frames
        # (trainer.py:775-778); the prediction is resized to the CROPPED height
        gt_height = int(round(gt_height * 0.75))
        gt_depth = gt_depth[:gt_height]
    pred_depth = 1 / resize_linear(pred_disp, gt_width, gt_height)
    if eval_split == "cityscapes":
        gt_depth = gt_depth[256:, 192:1856]
        pred_depth = pred_depth[256:, 192:1856]
    if eval_split == "eigen":
        mask = eigen_crop_mask(gt_depth)
    else:
        mask = np.logical_and(gt_depth > MIN_VAL, gt_depth < MAX_VAL)
    pred_depth = pred_depth[mask]
    gt_depth = gt_depth[mask]
    pred_depth = pred_depth * pred_depth_scale_factor
    ratio = None
    if median_scaling:
        ratio = np.median(gt_depth) / np.median(pred_depth)
        pred_depth = pred_depth * ratio
    pred_depth[pred_depth < MIN_VAL] = MIN_VAL
    pred_depth[pred_depth > MAX_VAL] = MAX_VAL
    return compute_errors(gt_depth, pred_depth), ratio


def evaluate_disps(pred_disps, gt_depths, eval_split="eigen", median_scaling=True, pred_depth_scale_factor=1.0):
    """Mean of the 7 errors over a split (trainer.py:843)."""
    errors = [evaluate_image(pred_disps[i], gt_depths[i], eval_split, median_scaling, pred_depth_scale_factor)[0]
              for i in range(len(pred_disps))]
    return np.array(errors).mean(0)

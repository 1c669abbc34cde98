"""TEST INFRASTRUCTURE ONLY -- the reference's colour augmentation, restated with Pillow.

The reference augments PIL images with `torchvision.transforms.ColorJitter((0.8, 1.2), (0.8, 1.2), (0.8, 1.2), (-0.1, 0.1))`
(datasets/mono_dataset.py:62-75, 183-190; applied per frame and scale in `preprocess`, :88-111).  torchvision (pinned
0.11.1, requirements.txt:266) is absent from this image and from /root/reference; its published algorithm for PIL inputs
(`transforms/functional_pil.py`) is restated here on top of Pillow, which IS the arithmetic the reference runs:

  adjust_brightness(img, f) = ImageEnhance.Brightness(img).enhance(f)
  adjust_contrast(img, f)   = ImageEnhance.Contrast(img).enhance(f)
  adjust_saturation(img, f) = ImageEnhance.Color(img).enhance(f)
  adjust_hue(img, h)        : H channel of img.convert("HSV") += uint8(h * 255) (wrapping), back to RGB
  ColorJitter.forward       : fn_idx = randperm(4); the four operations in that order (0 brightness, 1 contrast,
                              2 saturation, 3 hue), a NEW parameter draw on every call (torchvision >= 0.9)

Parity pin: third-party algorithm, no golden vectors upstream -> pinned by running Pillow itself (tests/test_host_cpu.py).
"""
import numpy as np
from PIL import Image, ImageEnhance


def adjust_hue(img, hue_factor):
    h, s, v = img.convert("HSV").split()
    np_h = np.array(h, dtype=np.uint8)
    np_h = (np_h.astype(np.int64) + (int(hue_factor * 255) & 255)).astype(np.uint8)      # uint8 wrap-around add
    return Image.merge("HSV", (Image.fromarray(np_h, "L"), s, v)).convert("RGB")


def color_jitter(img_u8, order, brightness, contrast, saturation, hue):
    """img_u8 [H,W,3] uint8 -> [H,W,3] uint8; `order`: permutation of (0, 1, 2, 3)."""
    img = Image.fromarray(np.ascontiguousarray(img_u8))
    for fn in order:
        if fn == 0:
            img = ImageEnhance.Brightness(img).enhance(float(brightness))
        elif fn == 1:
            img = ImageEnhance.Contrast(img).enhance(float(contrast))
        elif fn == 2:
            img = ImageEnhance.Color(img).enhance(float(saturation))
        else:
            img = adjust_hue(img, float(hue))
    return np.asarray(img)

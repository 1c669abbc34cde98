"""Input pipeline on the device: raw uint8 frames -> the row-P dictionary `process_batch` consumes.

Replaces, for a whole batch at once, what the reference does per item on CPU workers with PIL / torchvision
(datasets/mono_dataset.py:89-112 `preprocess`, :143-190 `__getitem__`): horizontal flip, the LANCZOS image pyramid
(scale s is resized from scale s-1, like the reference's chained `transforms.Resize`), ColorJitter with one parameter
draw per item shared by all its frames, `ToTensor`, and the per-scale intrinsics.

* The pyramid reproduces PIL's `Image.resize(..., LANCZOS)` on 8-bit images BIT-EXACTLY: same support (3 x scale), same
  coefficient normalisation, same 22-bit fixed-point coefficients, horizontal pass then vertical pass with an 8-bit
  intermediate (Pillow `Resample.c`); the fixed-point dot products are evaluated as fp64 GEMMs, which are exact for these
  magnitudes (< 2^53).  tests/test_host_cpu.py compares against Pillow itself.
* ColorJitter follows torchvision's functional definitions (brightness / contrast / saturation blends, hue through HSV)
  on float images; torchvision's PIL path quantises to 8 bits after every operation, this one once at the end -- the
  augmentation distribution is the same, individual pixels can differ by an LSB or two.
"""
import math

import numpy as np
import torch

KITTI_K = ((0.58, 0, 0.5, 0), (0, 1.92, 0.5, 0), (0, 0, 1, 0), (0, 0, 0, 1))      # kitti_dataset.py:26-29
PRECISION_BITS = 32 - 8 - 2                                                        # Pillow Resample.c


def _lanczos(x):
    def sinc(v):                                # Pillow Resample.c sinc_filter
        if v == 0.0:
            return 1.0
        v = v * math.pi
        return math.sin(v) / v
    if -3.0 <= x < 3.0:
        return sinc(x) * sinc(x / 3)
    return 0.0


def lanczos_matrix(insize, outsize):
    """[outsize, insize] fp64 matrix of Pillow's fixed-point LANCZOS coefficients (precompute_coeffs + normalize_coeffs_8bpc)."""
    scale = insize / outsize
    filterscale = max(scale, 1.0)
    support = 3.0 * filterscale
    m = np.zeros((outsize, insize), dtype=np.float64)
    for xx in range(outsize):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), insize)
        ss = 1.0 / filterscale
        w = np.array([_lanczos((x + xmin - center + 0.5) * ss) for x in range(xmax - xmin)], dtype=np.float64)
        w /= w.sum()
        fixed = np.where(w < 0, np.trunc(-0.5 + w * (1 << PRECISION_BITS)), np.trunc(0.5 + w * (1 << PRECISION_BITS)))
        m[xx, xmin:xmax] = fixed
    return m


class LanczosResize:
    """uint8 [.., Hin, Win] -> uint8 [.., Hout, Wout], bit-exact with PIL's LANCZOS resize of an 8-bit image."""

    def __init__(self, in_hw, out_hw, device):
        self.in_hw, self.out_hw = tuple(in_hw), tuple(out_hw)
        self.mh = torch.from_numpy(lanczos_matrix(in_hw[1], out_hw[1])).to(device)      # horizontal: [Wout, Win]
        self.mv = torch.from_numpy(lanczos_matrix(in_hw[0], out_hw[0])).to(device)      # vertical:   [Hout, Hin]

    @staticmethod
    def _clip8(acc):
        return torch.floor((acc + float(1 << (PRECISION_BITS - 1))) / float(1 << PRECISION_BITS)).clamp_(0, 255)

    def __call__(self, img_u8):
        x = img_u8.to(torch.float64)
        if self.in_hw[1] != self.out_hw[1]:
            x = self._clip8(x @ self.mh.t())                        # rows stay, columns resampled; 8-bit intermediate
        if self.in_hw[0] != self.out_hw[0]:
            x = self._clip8(self.mv @ x)
        return x.to(torch.uint8)


# ---- torchvision.transforms.functional colour operations on float [B,3,H,W] images in [0,1] ----------------------
def _gray(img):
    return (0.2989 * img[:, 0] + 0.587 * img[:, 1] + 0.114 * img[:, 2]).unsqueeze(1)


def _blend(a, b, ratio):
    return (ratio * a + (1.0 - ratio) * b).clamp_(0.0, 1.0)


def adjust_brightness(img, f):
    return _blend(img, torch.zeros_like(img), f)


def adjust_contrast(img, f):
    return _blend(img, _gray(img).mean((-3, -2, -1), keepdim=True), f)


def adjust_saturation(img, f):
    return _blend(img, _gray(img), f)


def adjust_hue(img, h):
    r, g, b = img.unbind(1)
    maxc, minc = img.amax(1), img.amin(1)
    eqc = maxc == minc
    cr = maxc - minc
    ones = torch.ones_like(maxc)
    s = cr / torch.where(eqc, ones, maxc)
    crd = torch.where(eqc, ones, cr)
    rc, gc, bc = (maxc - r) / crd, (maxc - g) / crd, (maxc - b) / crd
    hr = (maxc == r) * (bc - gc)
    hg = ((maxc == g) & (maxc != r)) * (2.0 + rc - bc)
    hb = ((maxc != g) & (maxc != r)) * (4.0 + gc - rc)
    hh = torch.fmod((hr + hg + hb) / 6.0 + 1.0, 1.0)
    hh = (hh + h.reshape(-1, 1, 1)) % 1.0
    v = maxc
    i = torch.floor(hh * 6.0)
    f = hh * 6.0 - i
    i = i.to(torch.int64) % 6
    p = (v * (1.0 - s)).clamp(0.0, 1.0)
    q = (v * (1.0 - s * f)).clamp(0.0, 1.0)
    t = (v * (1.0 - s * (1.0 - f))).clamp(0.0, 1.0)
    sel = torch.stack([torch.stack([v, q, p, p, t, v], 1), torch.stack([t, v, v, q, p, p], 1),
                       torch.stack([p, p, t, v, v, q], 1)], 1)                               # [B,3,6,H,W]
    idx = i[:, None, None].expand(-1, 3, 1, -1, -1)
    return sel.gather(2, idx)[:, :, 0]


def draw_jitter_params(batch, generator=None, brightness=(0.8, 1.2), contrast=(0.8, 1.2), saturation=(0.8, 1.2),
                       hue=(-0.1, 0.1)):
    """One ColorJitter.get_params draw per item (mono_dataset.py:183-185): order permutation + four factors."""
    u = lambda lo, hi: lo + (hi - lo) * torch.rand(batch, generator=generator)          # noqa: E731
    order = torch.stack([torch.randperm(4, generator=generator) for _ in range(batch)])
    return {"order": order, "brightness": u(*brightness), "contrast": u(*contrast), "saturation": u(*saturation),
            "hue": u(*hue)}


def color_jitter(img, params, apply):
    """img float [B,3,H,W]; params from draw_jitter_params; apply [B] bool (mono_dataset.py:143: p = 0.5 per item)."""
    dev = img.device
    fac = {k: params[k].to(dev).reshape(-1, 1, 1, 1) for k in ("brightness", "contrast", "saturation")}
    hue = params["hue"].to(dev)
    order_host = params["order"].cpu()                 # which operations are due at a step is decided on the host
    order = order_host.to(dev)
    ops = (lambda x: adjust_brightness(x, fac["brightness"]), lambda x: adjust_contrast(x, fac["contrast"]),
           lambda x: adjust_saturation(x, fac["saturation"]), lambda x: adjust_hue(x, hue))
    out = img
    for step in range(4):                      # per-item operation order: evaluate each op, select where it is due
        nxt = out
        for j, op in enumerate(ops):
            due = (order[:, step] == j).reshape(-1, 1, 1, 1)
            if bool((order_host[:, step] == j).any()):            # no device sync
                nxt = torch.where(due, op(out), nxt)
        out = nxt
    return torch.where(apply.to(dev).reshape(-1, 1, 1, 1), out, img)


class DeviceInputPipeline:
    def __init__(self, raw_hw, height, width, device, num_scales=4, frame_idxs=(0, -1, 1), K=KITTI_K, is_train=True):
        self.device = torch.device(device)
        self.height, self.width, self.num_scales = height, width, num_scales
        self.frame_idxs, self.is_train = tuple(frame_idxs), is_train
        self.resize, prev = [], tuple(raw_hw)
        for s in range(num_scales):
            hw = (height // 2 ** s, width // 2 ** s)
            self.resize.append(LanczosResize(prev, hw, self.device))
            prev = hw
        self.K, self.inv_K = [], []
        for s in range(num_scales):                        # mono_dataset.py:173-182
            k = np.array(K, dtype=np.float32)
            k[0, :] *= width // (2 ** s)
            k[1, :] *= height // (2 ** s)
            self.K.append(torch.from_numpy(k).to(self.device))
            self.inv_K.append(torch.from_numpy(np.linalg.pinv(k)).to(self.device))

    @torch.no_grad()
    def __call__(self, raw, do_color_aug=None, do_flip=None, jitter=None, generator=None):
        """raw: {frame id: uint8 [B,3,Hraw,Wraw]} (a missing neighbour = all zeros, mono_dataset.py:160-164).
        do_color_aug / do_flip: [B] bool (default: drawn with p = 0.5 each when is_train, mono_dataset.py:143-144)."""
        B = raw[self.frame_idxs[0]].shape[0]
        if do_color_aug is None:
            do_color_aug = (torch.rand(B, generator=generator) > 0.5) if self.is_train else torch.zeros(B, dtype=torch.bool)
        if do_flip is None:
            do_flip = (torch.rand(B, generator=generator) > 0.5) if self.is_train else torch.zeros(B, dtype=torch.bool)
        if jitter is None:
            jitter = draw_jitter_params(B, generator)
        flip = do_flip.to(self.device).reshape(-1, 1, 1, 1)
        inputs = {}
        for f in self.frame_idxs:
            img = raw[f].to(self.device)
            img = torch.where(flip, img.flip(-1), img)
            blank = (img.reshape(B, -1).sum(1) == 0)
            for s in range(self.num_scales):
                img = self.resize[s](img)                               # uint8, chained from the previous scale
                color = img.to(torch.float32) / 255.0                    # ToTensor
                aug = color_jitter(color, jitter, do_color_aug.to(self.device) & ~blank)
                inputs[("color", f, s)] = color
                inputs[("color_aug", f, s)] = aug
        for s in range(self.num_scales):
            inputs[("K", s)] = self.K[s][None].repeat(B, 1, 1)
            inputs[("inv_K", s)] = self.inv_K[s][None].repeat(B, 1, 1)
        return inputs

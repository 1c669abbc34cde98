"""Input pipeline on the device: raw uint8 frames -> the row-P dictionary `process_batch` consumes.

Replaces, for a whole batch at once, what the reference does per item on CPU workers with PIL / torchvision
(datasets/mono_dataset.py:89-112 `preprocess`, :143-190 `__getitem__`): horizontal flip, the LANCZOS image pyramid
(scale s is resized from scale s-1, like the reference's chained `transforms.Resize`), ColorJitter with one parameter
draw per item shared by all its frames, `ToTensor`, and the per-scale intrinsics.

* The pyramid reproduces PIL's `Image.resize(..., LANCZOS)` on 8-bit images BIT-EXACTLY: same support (3 x scale), same
  coefficient normalisation, same 22-bit fixed-point coefficients, horizontal pass then vertical pass with an 8-bit
  intermediate (Pillow `Resample.c`); the fixed-point dot products are evaluated as fp64 GEMMs, which are exact for these
  magnitudes (< 2^53).  tests/test_host_cpu.py compares against Pillow itself.
* ColorJitter is torchvision's PIL path (what the reference runs: PIL images through `transforms.ColorJitter`) restated
  on uint8 batches -- ImageEnhance blends, Pillow's RGB <-> HSV integer conversions, 8-bit quantisation after every
  operation, a fresh parameter draw for every frame and scale -- BIT-EXACT with Pillow (tests/test_host_cpu.py against
  a Pillow-based restatement kept with the test infrastructure).
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


# ---- ColorJitter: torchvision's PIL path, bit for bit, on uint8 batches -------------------------------------------------
# The reference jitters PIL images (mono_dataset.py:183-190 -> torchvision functional_pil -> Pillow), i.e. 8-bit images
# with a quantisation after every operation.  Each operation below is Pillow's arithmetic restated on tensors (fp32 / fp64
# exactly where the C code uses float / double), verified bit-exact against Pillow itself (tests/test_host_cpu.py):
#   ImagingBlend (ImageEnhance):  out = clip(trunc(deg + f * (img - deg)))           in fp32
#   RGB -> L:                     (19595 R + 38470 G + 7471 B + 0x8000) >> 16
#   contrast's degenerate image:  int(mean(L) + 0.5);  saturation's: L replicated;  brightness's: zeros
#   hue: Convert.c rgb2hsv / hsv2rgb (8-bit H, S, V) around  H += uint8(h * 255)
def _gray_u8(img):
    """int64 [B,3,H,W] -> [B,1,H,W] (Pillow's RGB -> L)."""
    return ((img[:, 0] * 19595 + img[:, 1] * 38470 + img[:, 2] * 7471 + 0x8000) >> 16).unsqueeze(1)


def _blend_u8(deg, img, f):
    """PIL.Image.blend(degenerate, image, factor) for 8-bit images; f [B]."""
    f = f.to(torch.float32).reshape(-1, 1, 1, 1)
    d, x = deg.to(torch.float32), img.to(torch.float32)
    t = d + f * (x - d)                                       # fp32, one rounding per operation as in C
    return torch.trunc(t).clamp_(0, 255).to(torch.int64)


def adjust_brightness(img, f):
    return _blend_u8(torch.zeros_like(img), img, f)


def adjust_contrast(img, f):
    g = _gray_u8(img)
    mean = torch.floor(g.to(torch.float64).mean((1, 2, 3)) + 0.5).to(torch.int64).reshape(-1, 1, 1, 1)
    return _blend_u8(mean.expand_as(img), img, f)


def adjust_saturation(img, f):
    return _blend_u8(_gray_u8(img).expand_as(img), img, f)


def _rgb2hsv_u8(img):
    r, g, b = img.unbind(1)
    maxc, minc = img.amax(1), img.amin(1)
    eq = maxc == minc
    f32, f64 = torch.float32, torch.float64
    cr = (maxc - minc).to(f32)
    crs = torch.where(eq, torch.ones_like(cr), cr)
    s = cr / torch.where(eq, torch.ones_like(maxc), maxc).to(f32)
    rc, gc, bc = (maxc - r).to(f32) / crs, (maxc - g).to(f32) / crs, (maxc - b).to(f32) / crs
    h = torch.where(r == maxc, bc - gc,
                    torch.where(g == maxc, (2.0 + rc.to(f64) - bc.to(f64)).to(f32), (4.0 + gc.to(f64) - rc.to(f64)).to(f32)))
    h = torch.fmod(h.to(f64) / 6.0 + 1.0, 1.0).to(f32)
    uh = (h.to(f64) * 255.0).to(torch.int64).clamp_(0, 255)
    us = (s.to(f64) * 255.0).to(torch.int64).clamp_(0, 255)
    zero = torch.zeros_like(uh)
    return torch.where(eq, zero, uh), torch.where(eq, zero, us), maxc


def _hsv2rgb_u8(h, s, v):
    f32, f64 = torch.float32, torch.float64
    hf = h.to(f32).to(f64) * 6.0 / 255.0
    i = torch.floor(hf)
    f = (hf - i).to(f32).to(f64)
    fs = (s.to(f32).to(f64) / 255.0).to(f32).to(f64)
    vv = v.to(f32).to(f64)

    def rnd(x):                                                # C round(): positive values, half away from zero
        return torch.floor(x + 0.5).clamp_(0, 255).to(torch.int64)
    p, q, t = rnd(vv * (1.0 - fs)), rnd(vv * (1.0 - fs * f)), rnd(vv * (1.0 - fs * (1.0 - f)))
    sel = i.to(torch.int64) % 6
    pick = lambda opts: torch.stack(opts, 0).gather(0, sel.unsqueeze(0))[0]      # noqa: E731
    r, g, b = pick([v, q, p, p, t, v]), pick([t, v, v, q, p, p]), pick([p, p, t, v, v, q])
    gray = s == 0
    return torch.stack([torch.where(gray, v, r), torch.where(gray, v, g), torch.where(gray, v, b)], 1)


def adjust_hue(img, hue):
    """hue [B] in [-0.5, 0.5]: H (8 bit) += uint8(hue * 255), wrapping (torchvision functional_pil.adjust_hue)."""
    h, s, v = _rgb2hsv_u8(img)
    shift = torch.trunc(hue.to(torch.float64) * 255.0).to(torch.int64) & 255
    return _hsv2rgb_u8((h + shift.to(h.device).reshape(-1, 1, 1)) & 255, s, v)


def draw_jitter_params(batch, generator=None, brightness=(0.8, 1.2), contrast=(0.8, 1.2), saturation=(0.8, 1.2),
                       hue=(-0.1, 0.1)):
    """One `ColorJitter.get_params` draw per item (torchvision order: randperm(4), then brightness, contrast, saturation,
    hue), as the reference's transform object draws on EVERY call (mono_dataset.py:183-185 builds
    `transforms.ColorJitter(...)`, whose forward re-draws: each frame and scale of an item gets its own parameters)."""
    order, fac = [], {k: [] for k in ("brightness", "contrast", "saturation", "hue")}
    rng = {"brightness": brightness, "contrast": contrast, "saturation": saturation, "hue": hue}
    for _ in range(batch):
        order.append(torch.randperm(4, generator=generator))
        for k in ("brightness", "contrast", "saturation", "hue"):
            fac[k].append(torch.empty(1).uniform_(rng[k][0], rng[k][1], generator=generator))
    out = {k: torch.cat(v) for k, v in fac.items()}
    out["order"] = torch.stack(order)
    return out


def color_jitter(img_u8, params, apply):
    """img_u8 uint8 [B,3,H,W] -> uint8; params from draw_jitter_params; apply [B] bool (mono_dataset.py:143: p = 0.5 per
    item; blank frames are never jittered, :107-110).  Per item the four operations run in the item's own order."""
    dev = img_u8.device
    fac = {k: params[k].to(dev) for k in ("brightness", "contrast", "saturation", "hue")}
    order_host = params["order"].cpu()                 # which operations are due at a step is decided on the host
    order = order_host.to(dev)
    ops = (lambda x: adjust_brightness(x, fac["brightness"]), lambda x: adjust_contrast(x, fac["contrast"]),
           lambda x: adjust_saturation(x, fac["saturation"]), lambda x: adjust_hue(x, fac["hue"]))
    img = img_u8.to(torch.int64)
    out = img
    for step in range(4):                      # per-item operation order: evaluate each op, select where it is due
        nxt = out
        for j, op in enumerate(ops):
            if bool((order_host[:, step] == j).any()):            # no device sync
                due = (order[:, step] == j).reshape(-1, 1, 1, 1)
                nxt = torch.where(due, op(out), nxt)
        out = nxt
    return torch.where(apply.to(dev).reshape(-1, 1, 1, 1), out, img).to(torch.uint8)


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
        flip = do_flip.to(self.device).reshape(-1, 1, 1, 1)
        inputs = {}
        for f in self.frame_idxs:
            img = raw[f].to(self.device)
            img = torch.where(flip, img.flip(-1), img)
            blank = (img.reshape(B, -1).sum(1) == 0)
            for s in range(self.num_scales):
                img = self.resize[s](img)                               # uint8, chained from the previous scale
                # `jitter`: a dict (one draw for every call: tests) or {(frame, scale): dict}; default: a fresh draw per
                # frame and scale, like the reference's transform object (see draw_jitter_params)
                prm = jitter.get((f, s), jitter) if isinstance(jitter, dict) and (f, s) in jitter else jitter
                if prm is None:
                    prm = draw_jitter_params(B, generator)
                aug = color_jitter(img, prm, do_color_aug.to(self.device) & ~blank)
                inputs[("color", f, s)] = img.to(torch.float32) / 255.0          # ToTensor
                inputs[("color_aug", f, s)] = aug.to(torch.float32) / 255.0
        for s in range(self.num_scales):
            inputs[("K", s)] = self.K[s][None].repeat(B, 1, 1)
            inputs[("inv_K", s)] = self.inv_K[s][None].repeat(B, 1, 1)
        return inputs

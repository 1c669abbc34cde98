"""TEST INFRASTRUCTURE ONLY -- names, shapes and dtypes of every entry of
`networks.RepDepth(opt).state_dict()` for `--adapter` Stage-1 (SURVEY.md 8(b)-2:
2 888 entries at 31B), enumerated from the architecture.  Checked against the
reference's own state_dict by tests/golden/state_spec.npz.
"""
import torch

from .ref_model import CHANNELS, LAYERS, LK_SIZES

F32, I64 = torch.float32, torch.int64


def _bn(spec, p, c):
    spec[p + ".weight"] = ((c,), F32)
    spec[p + ".bias"] = ((c,), F32)
    spec[p + ".running_mean"] = ((c,), F32)
    spec[p + ".running_var"] = ((c,), F32)
    spec[p + ".num_batches_tracked"] = ((), I64)


def _conv_bn(spec, p, co, ci, k, groups=1):
    spec[p + ".conv.weight"] = ((co, ci // groups, k, k), F32)
    _bn(spec, p + ".bn", co)


def _replknet(spec, p, ch, ratio=0.25):
    c0 = ch[0]
    _conv_bn(spec, p + ".stem.0", c0, 3, 3)
    _conv_bn(spec, p + ".stem.1", c0, c0, 3, c0)
    _conv_bn(spec, p + ".stem.2", c0, c0, 1)
    _conv_bn(spec, p + ".stem.3", c0, c0, 3, c0)
    for s in range(4):
        c, k, hid = ch[s], LK_SIZES[s], int(ch[s] * ratio)
        for j in range(2 * LAYERS[s]):
            b = f"{p}.stages.{s}.blocks.{j}"
            if j % 2 == 0:
                _conv_bn(spec, b + ".pw1", c, c, 1)
                _conv_bn(spec, b + ".pw2", c, c, 1)
                _conv_bn(spec, b + ".large_kernel.lkb_origin", c, c, k, c)
                _conv_bn(spec, b + ".large_kernel.small_conv", c, c, 5, c)
                _bn(spec, b + ".prelkb_bn", c)
                spec[b + ".adapter.D_fc1.weight"] = ((hid, c, 3, 3), F32)
                spec[b + ".adapter.D_fc1.bias"] = ((hid,), F32)
                spec[b + ".adapter.D_fc2.weight"] = ((c, hid), F32)
                spec[b + ".adapter.D_fc2.bias"] = ((c,), F32)
            else:
                _bn(spec, b + ".preffn_bn", c)
                _conv_bn(spec, b + ".pw1", 4 * c, c, 1)
                _conv_bn(spec, b + ".pw2", c, 4 * c, 1)
                h2 = int(c * 0.25)
                spec[b + ".mlp_adapter.D_fc1.weight"] = ((h2, c), F32)
                spec[b + ".mlp_adapter.D_fc1.bias"] = ((h2,), F32)
                spec[b + ".mlp_adapter.D_fc2.weight"] = ((c, h2), F32)
                spec[b + ".mlp_adapter.D_fc2.bias"] = ((c,), F32)
        if s < 3:
            _conv_bn(spec, f"{p}.transitions.{s}.0", ch[s + 1], c, 1)
            _conv_bn(spec, f"{p}.transitions.{s}.1", ch[s + 1], ch[s + 1], 3, ch[s + 1])


def _decoder(spec, p, ch):
    def cb(name, ci, co):
        spec[f"{p}.{name}.conv.conv.weight"] = ((co, ci, 3, 3), F32)
        spec[f"{p}.{name}.conv.conv.bias"] = ((co,), F32)
    for n, i in enumerate(range(3, -1, -1)):
        ci, co = ch[i], ch[i] // 2
        cb(f"upconvs_0.{n}", ci, co)
        cb(f"upconvs_1.{n}", co if i == 0 else ci, co)
    cb("upconvs_0.4", ch[0] // 2, ch[0] // 4)
    cb("upconvs_1.4", ch[0] // 4, ch[0] // 4)
    spec[f"{p}.disp_convs.0.conv.weight"] = ((1, ch[0] // 4, 3, 3), F32)
    spec[f"{p}.disp_convs.0.conv.bias"] = ((1,), F32)


def _resnet18(spec, p):
    spec[p + ".conv1.weight"] = ((64, 6, 7, 7), F32)
    _bn(spec, p + ".bn1", 64)
    cin = 64
    for li, c in ((1, 64), (2, 128), (3, 256), (4, 512)):
        for b in range(2):
            q = f"{p}.layer{li}.{b}"
            spec[q + ".conv1.weight"] = ((c, cin if b == 0 else c, 3, 3), F32)
            _bn(spec, q + ".bn1", c)
            spec[q + ".conv2.weight"] = ((c, c, 3, 3), F32)
            _bn(spec, q + ".bn2", c)
            if b == 0 and li > 1:
                spec[q + ".downsample.0.weight"] = ((c, cin, 1, 1), F32)
                _bn(spec, q + ".downsample.1", c)
        cin = c
    spec[p + ".fc.weight"] = ((1000, 512), F32)
    spec[p + ".fc.bias"] = ((1000,), F32)


def _decoder_adapter(spec, p, ch, ratio=0.25):
    """Stage-2 `--dc` additions of dc_ft_init design 1 (repdepth.py:199-203; depth_decoder_v2.py:19-30)."""
    cin, cout = ch[3] + ch[0], ch[0] // 4
    hid = int((cin + cout) / 2 * ratio)
    spec[f"{p}.adapter.D_fc1.weight"] = ((hid, cin), F32)
    spec[f"{p}.adapter.D_fc1.bias"] = ((hid,), F32)
    spec[f"{p}.adapter.D_fc2.weight"] = ((cout, hid), F32)
    spec[f"{p}.adapter.D_fc2.bias"] = ((cout,), F32)
    spec[f"{p}.deconv_adpt.weight"] = ((cout, cout, 3, 3), F32)
    spec[f"{p}.deconv_adpt.bias"] = ((cout,), F32)


def state_spec(rep_size="b", num_depth_bins=96, dc=False):
    ch = CHANNELS[rep_size]
    spec = {}
    _replknet(spec, "encoder.replk", ch)
    spec["encoder.reduce_conv.0.weight"] = ((ch[0], ch[0] + num_depth_bins, 3, 3), F32)
    spec["encoder.reduce_conv.0.bias"] = ((ch[0],), F32)
    _decoder(spec, "depth", ch)
    _replknet(spec, "mono_encoder", ch)
    _decoder(spec, "mono_depth", ch)
    _resnet18(spec, "pose_encoder.encoder")
    for i, (co, ci, k) in enumerate(((256, 512, 1), (256, 256, 3), (256, 256, 3), (12, 256, 1))):
        spec[f"pose.net.{i}.weight"] = ((co, ci, k, k), F32)
        spec[f"pose.net.{i}.bias"] = ((co,), F32)
    if dc:
        _decoder_adapter(spec, "depth", ch)
        _decoder_adapter(spec, "mono_depth", ch)
    return spec

"""`ppeadepth.networks` module API (reference: networks/__init__.py:2-12), --adapter path."""
from .resnet_encoder import ResnetEncoder
from .pose_decoder import PoseDecoder
from .replknet_adapter import (conv_bn, conv_bn_relu, create_RepLKNet31B_Adapter, create_RepLKNet31L_Adapter,
                               RepLKNetAdapter)
from .replk_matching_adapter import RepLKMatchingAdapter
from .depth_decoder_v2 import DepthDecoderV2
from .repdepth import RepDepth

__all__ = ["ResnetEncoder", "PoseDecoder", "conv_bn", "conv_bn_relu", "create_RepLKNet31B_Adapter",
           "create_RepLKNet31L_Adapter", "RepLKNetAdapter", "RepLKMatchingAdapter", "DepthDecoderV2",
           "RepDepth"]

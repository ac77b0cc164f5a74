"""Fast-SCNN — module tree / state_dict of segmentron/models/fast_scnn.py:16-161, forward on the
HIP kernels ("only hot-path ops — works on the new kernels for free").

No backbone (`SegBaseModel(need_backbone=False)`): learning-to-downsample (3x3 stride-2 conv
WITHOUT padding + two stride-2 separable convs) -> global feature extractor (nine inverted
residuals, pyramid pooling, 1x1) -> feature fusion (bilinear x4 of the 1/32 branch, two 1x1+BN
paths, ReLU of their sum) -> classifier (two separable convs, Dropout2d, 1x1) -> bilinear to the
input size.  With cfg.SOLVER.AUX two more heads read the 1/8 and 1/32 features.  Quirks kept:
the stem's `_ConvBNReLU` ignores `norm_layer` (fast_scnn.py:75), `FeatureFusionModule.dwconv` is
a dense 1x1 despite its name (:124), the class is spelt `Classifer` (:146)."""
import torch
import torch.nn as nn

from .. import functional as F
from ..config import cfg
from ..modules import InvertedResidual, PyramidPooling, SeparableConv2d, _ConvBNReLU, get_norm
from .model_zoo import MODEL_REGISTRY
from .segbase import SegBaseModel

__all__ = ["FastSCNN"]


def _logits(act, conv, nclass):
    """1x1 classifier (+bias) -> NHWC logits as a view of a channel-padded buffer."""
    N, H, W, _ = act.shape
    vec = 8 if act.t.dtype == torch.bfloat16 else 4
    pitch = (nclass + 2 * vec - 1) // vec * vec
    out = torch.empty((N, H, W, pitch), dtype=act.t.dtype, device=act.t.device)[..., :nclass]
    return F.conv_bn(act, conv, None, out=out).t


def _dropout2d(act, drop, training):
    """nn.Dropout2d on a deferred activation: a per-(image, channel) multiplier applied while the
    activation is materialised (seg_bn_apply chan_mul)."""
    p = drop.p
    if not training or p <= 0.0:
        return act
    N, C = act.shape[0], act.shape[-1]
    mul = (torch.rand((N, C), device=act.t.device) >= p).float() / (1.0 - p)
    return F.Act(F.materialize(act, chan_mul=mul))


@MODEL_REGISTRY.register()
class FastSCNN(SegBaseModel):
    def __init__(self):
        super().__init__(need_backbone=False)
        self.aux = cfg.SOLVER.AUX
        self.norm_layer = get_norm(cfg.MODEL.BN_TYPE)
        self.learning_to_downsample = LearningToDownsample(32, 48, 64, norm_layer=self.norm_layer)
        self.global_feature_extractor = GlobalFeatureExtractor(
            64, [64, 96, 128], 128, 6, [3, 3, 3], norm_layer=self.norm_layer)
        self.feature_fusion = FeatureFusionModule(64, 128, 128, norm_layer=self.norm_layer)
        self.classifier = Classifer(128, self.nclass, norm_layer=self.norm_layer)
        decoder_list = ["learning_to_downsample", "global_feature_extractor", "feature_fusion",
                        "classifier"]
        if self.aux:
            self.auxlayer1 = _aux_head(64, self.nclass, self.norm_layer)
            self.auxlayer2 = _aux_head(128, self.nclass, self.norm_layer)
            decoder_list += ["auxlayer1", "auxlayer2"]
        self.__setattr__("decoder", decoder_list)

    def forward(self, x):
        from .. import compute_dtype
        size = x.shape[2:]
        lazy = F.want_lazy_logits(self.training)
        hi = self.learning_to_downsample(F.Act(F.image_to_nhwc(x, compute_dtype())))
        lo = self.global_feature_extractor(hi)
        y = self.classifier(self.feature_fusion(hi, lo))
        outputs = [F.logits_to_nchw(y, size, align_corners=True, lazy=lazy)]
        if self.aux:
            for head, feat in ((self.auxlayer1, hi), (self.auxlayer2, lo)):
                a = F.conv_bn(feat, head[0], head[1])
                a.relu = True
                a = _dropout2d(a, head[3], self.training)
                outputs.append(F.logits_to_nchw(_logits(a, head[4], self.nclass), size,
                                                align_corners=True, lazy=lazy))
        return tuple(outputs)


def _aux_head(in_channels, nclass, norm_layer):
    return nn.Sequential(nn.Conv2d(in_channels, 32, 3, padding=1, bias=False), norm_layer(32),
                         nn.ReLU(True), nn.Dropout2d(0.1), nn.Conv2d(32, nclass, 1))


class LearningToDownsample(nn.Module):
    def __init__(self, dw_channels1=32, dw_channels2=48, out_channels=64,
                 norm_layer=nn.BatchNorm2d):
        super().__init__()
        self.conv = _ConvBNReLU(3, dw_channels1, 3, 2)  # (padding 0, default norm: as the reference)
        self.dsconv1 = SeparableConv2d(dw_channels1, dw_channels2, stride=2, relu_first=False,
                                       norm_layer=norm_layer)
        self.dsconv2 = SeparableConv2d(dw_channels2, out_channels, stride=2, relu_first=False,
                                       norm_layer=norm_layer)

    def forward(self, act):
        return self.dsconv2(self.dsconv1(self.conv(act)))


class GlobalFeatureExtractor(nn.Module):
    def __init__(self, in_channels=64, block_channels=(64, 96, 128), out_channels=128, t=6,
                 num_blocks=(3, 3, 3), norm_layer=nn.BatchNorm2d):
        super().__init__()
        self.bottleneck1 = self._make_layer(InvertedResidual, in_channels, block_channels[0],
                                            num_blocks[0], t, 2, norm_layer=norm_layer)
        self.bottleneck2 = self._make_layer(InvertedResidual, block_channels[0],
                                            block_channels[1], num_blocks[1], t, 2,
                                            norm_layer=norm_layer)
        self.bottleneck3 = self._make_layer(InvertedResidual, block_channels[1],
                                            block_channels[2], num_blocks[2], t, 1,
                                            norm_layer=norm_layer)
        self.ppm = PyramidPooling(block_channels[2], norm_layer=norm_layer)
        self.out = _ConvBNReLU(block_channels[2] * 2, out_channels, 1, norm_layer=norm_layer)

    def _make_layer(self, block, inplanes, planes, blocks, t=6, stride=1,
                    norm_layer=nn.BatchNorm2d):
        layers = [block(inplanes, planes, stride, t, norm_layer=norm_layer)]
        for _ in range(1, blocks):
            layers.append(block(planes, planes, 1, t, norm_layer=norm_layer))
        return nn.Sequential(*layers)

    def forward(self, a):
        for seq in (self.bottleneck1, self.bottleneck2, self.bottleneck3):
            for m in seq:
                a = m(a)
        return self.out(self.ppm(a))


class FeatureFusionModule(nn.Module):
    def __init__(self, highter_in_channels, lower_in_channels, out_channels, scale_factor=4,
                 norm_layer=nn.BatchNorm2d):
        super().__init__()
        self.scale_factor = scale_factor
        self.dwconv = _ConvBNReLU(lower_in_channels, out_channels, 1, norm_layer=norm_layer)
        self.conv_lower_res = nn.Sequential(nn.Conv2d(out_channels, out_channels, 1),
                                            norm_layer(out_channels))
        self.conv_higher_res = nn.Sequential(nn.Conv2d(highter_in_channels, out_channels, 1),
                                             norm_layer(out_channels))
        self.relu = nn.ReLU(True)

    def forward(self, hi, lo):
        # F.interpolate(scale_factor=4, align_corners=True) (fast_scnn.py:137): floor(in * 4)
        N, H, W, _ = lo.shape
        up = F.Act(F.bilinear(lo, (H * 4, W * 4), align_corners=True))
        up = self.dwconv(up)
        up = F.conv_bn(up, self.conv_lower_res[0], self.conv_lower_res[1])
        hr = F.conv_bn(hi, self.conv_higher_res[0], self.conv_higher_res[1])
        if tuple(hr.shape) != tuple(up.shape):
            raise RuntimeError("The size of tensor a (%d) must match the size of tensor b (%d) at "
                               "non-singleton dimension 3" % (hr.shape[2], up.shape[2]))
        return F.Act(F.materialize(hr, residual=up, post_relu=True))


class Classifer(nn.Module):
    def __init__(self, dw_channels, num_classes, stride=1, norm_layer=nn.BatchNorm2d):
        super().__init__()
        self.dsconv1 = SeparableConv2d(dw_channels, dw_channels, stride=stride, relu_first=False,
                                       norm_layer=norm_layer)
        self.dsconv2 = SeparableConv2d(dw_channels, dw_channels, stride=stride, relu_first=False,
                                       norm_layer=norm_layer)
        self.conv = nn.Sequential(nn.Dropout2d(0.1), nn.Conv2d(dw_channels, num_classes, 1))
        self.num_classes = num_classes

    def forward(self, act):
        a = self.dsconv2(self.dsconv1(act))
        a = _dropout2d(a, self.conv[0], self.training)
        return _logits(a, self.conv[1], self.num_classes)

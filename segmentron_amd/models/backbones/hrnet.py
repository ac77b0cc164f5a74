"""HRNet backbone — module tree / state_dict of segmentron/models/backbones/hrnet.py:25-505
(BasicBlock :25, Bottleneck :57, HighResolutionModule :98, HighResolutionNet :241,
hrnet_w18_small_v1 :503), forward on the HIP kernels.

As in the reference the blocks / fuse layers hard-code ``nn.BatchNorm2d`` (only the stem, layer1
and transition1 take ``norm_layer``), `nn.Upsample` / `nn.ReLU` children are kept so the
Sequential indices (= state_dict keys) match, and the backbone returns a tuple of one activation
per branch.  MI355X form: every conv+BN is a deferred activation; the cross-resolution fuse
`sum_j f_ij(x_j)` + ReLU runs as one elementwise pass per term with the 1x1-conv+BN+nearest-
upsample operand gathered in place (`seg_nearest_add`) — the upsampled tensors never exist."""
import logging

import torch.nn as nn

from ... import functional as F
from ...config import cfg
from .build import BACKBONE_REGISTRY

__all__ = ["HighResolutionNet", "hrnet_w18_small_v1"]


def conv3x3(in_planes, out_planes, stride=1):
    return nn.Conv2d(in_planes, out_planes, 3, stride, 1, bias=False)


def _residual_out(out, x, downsample):
    identity = x if downsample is None else F.conv_bn(x, downsample[0], downsample[1])
    return F.Act(F.materialize(out, residual=identity, post_relu=True))


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = conv3x3(inplanes, planes, stride)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = conv3x3(planes, planes)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        out = F.conv_bn(x, self.conv1, self.bn1)
        out.relu = True
        out = F.conv_bn(out, self.conv2, self.bn2)
        return _residual_out(out, x, self.downsample)


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * self.expansion, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * self.expansion)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        out = F.conv_bn(x, self.conv1, self.bn1)
        out.relu = True
        out = F.conv_bn(out, self.conv2, self.bn2)
        out.relu = True
        out = F.conv_bn(out, self.conv3, self.bn3)
        return _residual_out(out, x, self.downsample)


def _conv_bn_chain(a, seq_of_seqs):
    """Sequential of Sequential(conv, bn[, relu]) on a deferred activation."""
    for s in seq_of_seqs:
        a = F.conv_bn(a, s[0], s[1])
        a.relu = len(s) > 2
    return a


class HighResolutionModule(nn.Module):
    def __init__(self, num_branches, blocks, num_blocks, num_inchannels, num_channels,
                 fuse_method, multi_scale_output=True):
        super().__init__()
        self._check_branches(num_branches, blocks, num_blocks, num_inchannels, num_channels)
        self.num_inchannels = num_inchannels
        self.fuse_method = fuse_method
        self.num_branches = num_branches
        self.multi_scale_output = multi_scale_output
        self.branches = self._make_branches(num_branches, blocks, num_blocks, num_channels)
        self.fuse_layers = self._make_fuse_layers()
        self.relu = nn.ReLU(False)

    def _check_branches(self, num_branches, blocks, num_blocks, num_inchannels, num_channels):
        for what, seq in (("NUM_BLOCKS", num_blocks), ("NUM_CHANNELS", num_channels),
                          ("NUM_INCHANNELS", num_inchannels)):
            if num_branches != len(seq):
                msg = "NUM_BRANCHES({}) <> {}({})".format(num_branches, what, len(seq))
                logging.error(msg)
                raise ValueError(msg)

    def _make_one_branch(self, index, block, num_blocks, num_channels, stride=1):
        downsample = None
        out_ch = num_channels[index] * block.expansion
        if stride != 1 or self.num_inchannels[index] != out_ch:
            downsample = nn.Sequential(
                nn.Conv2d(self.num_inchannels[index], out_ch, 1, stride, bias=False),
                nn.BatchNorm2d(out_ch))
        layers = [block(self.num_inchannels[index], num_channels[index], stride, downsample)]
        self.num_inchannels[index] = out_ch
        for _ in range(1, num_blocks[index]):
            layers.append(block(self.num_inchannels[index], num_channels[index]))
        return nn.Sequential(*layers)

    def _make_branches(self, num_branches, block, num_blocks, num_channels):
        return nn.ModuleList([self._make_one_branch(i, block, num_blocks, num_channels)
                              for i in range(num_branches)])

    def _make_fuse_layers(self):
        if self.num_branches == 1:
            return None
        ch = self.num_inchannels
        fuse_layers = []
        for i in range(self.num_branches if self.multi_scale_output else 1):
            row = []
            for j in range(self.num_branches):
                if j > i:
                    row.append(nn.Sequential(
                        nn.Conv2d(ch[j], ch[i], 1, 1, 0, bias=False),
                        nn.BatchNorm2d(ch[i]),
                        nn.Upsample(scale_factor=2 ** (j - i), mode="nearest")))
                elif j == i:
                    row.append(None)
                else:
                    chain = []
                    for k in range(i - j):
                        if k == i - j - 1:
                            chain.append(nn.Sequential(
                                nn.Conv2d(ch[j], ch[i], 3, 2, 1, bias=False),
                                nn.BatchNorm2d(ch[i])))
                        else:
                            chain.append(nn.Sequential(
                                nn.Conv2d(ch[j], ch[j], 3, 2, 1, bias=False),
                                nn.BatchNorm2d(ch[j]), nn.ReLU(False)))
                    row.append(nn.Sequential(*chain))
            fuse_layers.append(nn.ModuleList(row))
        return nn.ModuleList(fuse_layers)

    def get_num_inchannels(self):
        return self.num_inchannels

    def forward(self, x):
        """x: list of activations, one per branch -> list of fused activations (hrnet.py:211-229)."""
        if self.num_branches == 1:
            a = x[0]
            for blk in self.branches[0]:
                a = blk(a)
            return [a]
        x = list(x)
        for i in range(self.num_branches):
            for blk in self.branches[i]:
                x[i] = blk(x[i])
        fused = []
        nb = self.num_branches
        for i in range(len(self.fuse_layers)):
            row = self.fuse_layers[i]
            # same-resolution terms (identity, strided-conv chains) and upsampled terms
            same, ups = [], []
            for j in range(nb):
                if j == i:
                    same.append(x[j])
                elif j < i:
                    same.append(_conv_bn_chain(x[j], row[j]))
                else:
                    ups.append((F.conv_bn(x[j], row[j][0], row[j][1]), j - i))
            n_ops = max(len(same) - 1, 0) + len(ups)
            y, done = same[0], 0
            k = 1
            while k < len(same):  # y = act(y) + act(same[k])
                done += 1
                y = F.Act(F.materialize(y, residual=same[k], post_relu=done == n_ops))
                k += 1
            for up, shift in ups:
                done += 1
                y = F.Act(F.add_upsampled(y, up, shift, post_relu=done == n_ops))
            fused.append(y)
        return fused


blocks_dict = {"BASIC": BasicBlock, "BOTTLENECK": Bottleneck}


class HighResolutionNet(nn.Module):
    def __init__(self, norm_layer=nn.BatchNorm2d):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 3, 2, 1, bias=False)
        self.bn1 = norm_layer(64)
        self.conv2 = nn.Conv2d(64, 64, 3, 2, 1, bias=False)
        self.bn2 = norm_layer(64)
        self.relu = nn.ReLU(inplace=True)

        self.stage1_cfg = cfg.MODEL.HRNET.STAGE1
        num_channels = self.stage1_cfg["NUM_CHANNELS"][0]
        block = blocks_dict[self.stage1_cfg["BLOCK"]]
        num_blocks = self.stage1_cfg["NUM_BLOCKS"][0]
        self.layer1 = self._make_layer(block, 64, num_channels, num_blocks, norm_layer=norm_layer)
        pre = [block.expansion * num_channels]

        def expand(stage_cfg):
            blk = blocks_dict[stage_cfg["BLOCK"]]
            return [c * blk.expansion for c in stage_cfg["NUM_CHANNELS"]]

        self.stage2_cfg = cfg.MODEL.HRNET.STAGE2
        num_channels = expand(self.stage2_cfg)
        self.transition1 = self._make_transition_layer(pre, num_channels, norm_layer=norm_layer)
        self.stage2, pre = self._make_stage(self.stage2_cfg, num_channels)

        self.stage3_cfg = cfg.MODEL.HRNET.STAGE3
        num_channels = expand(self.stage3_cfg)
        self.transition2 = self._make_transition_layer(pre, num_channels)
        self.stage3, pre = self._make_stage(self.stage3_cfg, num_channels)

        self.stage4_cfg = cfg.MODEL.HRNET.STAGE4
        num_channels = expand(self.stage4_cfg)
        self.transition3 = self._make_transition_layer(pre, num_channels)
        self.stage4, pre = self._make_stage(self.stage4_cfg, num_channels,
                                            multi_scale_output=True)
        self.last_inp_channels = int(sum(pre))

    def _make_transition_layer(self, pre, cur, norm_layer=nn.BatchNorm2d):
        layers = []
        for i in range(len(cur)):
            if i < len(pre):
                if cur[i] != pre[i]:
                    layers.append(nn.Sequential(nn.Conv2d(pre[i], cur[i], 3, 1, 1, bias=False),
                                                norm_layer(cur[i]), nn.ReLU(inplace=True)))
                else:
                    layers.append(None)
            else:
                chain = []
                for j in range(i + 1 - len(pre)):
                    inch = pre[-1]
                    outch = cur[i] if j == i - len(pre) else inch
                    chain.append(nn.Sequential(nn.Conv2d(inch, outch, 3, 2, 1, bias=False),
                                               norm_layer(outch), nn.ReLU(inplace=True)))
                layers.append(nn.Sequential(*chain))
        return nn.ModuleList(layers)

    def _make_layer(self, block, inplanes, planes, blocks, stride=1, norm_layer=nn.BatchNorm2d):
        downsample = None
        if stride != 1 or inplanes != planes * block.expansion:
            downsample = nn.Sequential(
                nn.Conv2d(inplanes, planes * block.expansion, 1, stride, bias=False),
                norm_layer(planes * block.expansion))
        layers = [block(inplanes, planes, stride, downsample)]
        inplanes = planes * block.expansion
        for _ in range(1, blocks):
            layers.append(block(inplanes, planes))
        return nn.Sequential(*layers)

    def _make_stage(self, layer_config, num_inchannels, multi_scale_output=True):
        num_modules = layer_config["NUM_MODULES"]
        block = blocks_dict[layer_config["BLOCK"]]
        modules = []
        for i in range(num_modules):
            mso = multi_scale_output or i != num_modules - 1
            modules.append(HighResolutionModule(
                layer_config["NUM_BRANCHES"], block, layer_config["NUM_BLOCKS"], num_inchannels,
                layer_config["NUM_CHANNELS"], layer_config["FUSE_METHOD"], mso))
            num_inchannels = modules[-1].get_num_inchannels()
        return nn.Sequential(*modules), num_inchannels

    @staticmethod
    def _transition(layers, i, src):
        t = layers[i]
        if t is None:
            return src
        if isinstance(t[0], nn.Conv2d):  # Sequential(conv, bn, relu)
            a = F.conv_bn(src, t[0], t[1])
            a.relu = True
            return a
        return _conv_bn_chain(src, t)

    @staticmethod
    def _run_stage(stage, xs):
        for mod in stage:
            xs = mod(xs)
        return xs

    def forward(self, x):
        from ... import compute_dtype
        a = F.Act(F.image_to_nhwc(x, compute_dtype()))
        a = F.conv_bn(a, self.conv1, self.bn1)
        a.relu = True
        a = F.conv_bn(a, self.conv2, self.bn2)
        a.relu = True
        for blk in self.layer1:
            a = blk(a)
        xs = [self._transition(self.transition1, i, a)
              for i in range(self.stage2_cfg["NUM_BRANCHES"])]
        ys = self._run_stage(self.stage2, xs)
        xs = [self._transition(self.transition2, i, ys[-1] if self.transition2[i] is not None
                               else ys[i]) for i in range(self.stage3_cfg["NUM_BRANCHES"])]
        ys = self._run_stage(self.stage3, xs)
        xs = [self._transition(self.transition3, i, ys[-1] if self.transition3[i] is not None
                               else ys[i]) for i in range(self.stage4_cfg["NUM_BRANCHES"])]
        ys = self._run_stage(self.stage4, xs)
        return tuple(ys)

    def init_weights(self, pretrained=""):
        import os

        import torch
        logging.info("=> init weights from normal distribution")
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
        if os.path.isfile(pretrained):
            pretrained_dict = torch.load(pretrained)
            model_dict = self.state_dict()
            model_dict.update({k: v for k, v in pretrained_dict.items() if k in model_dict})
            self.load_state_dict(model_dict)


@BACKBONE_REGISTRY.register()
def hrnet_w18_small_v1(norm_layer=nn.BatchNorm2d):
    return HighResolutionNet(norm_layer=norm_layer)

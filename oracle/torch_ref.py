"""CPU fp32 restatement of the reference's hot-path module graph (TEST INFRASTRUCTURE ONLY).

Plain ``torch.nn.functional`` calls on CPU tensors driven by a flat ``state_dict`` — the same
arithmetic the reference delegates to ``torch.nn`` ("the CPU oracle is torch CPU
kernels driven by the reference's module graph").  Every function cites the reference lines it
follows.  It is pinned against the reference itself by ``oracle/gen_golden.py`` (fixtures under
``tests/golden``); the reference has no tests/golden vectors of its own.

Never imported by the product package.  Tolerance contract (BASELINE.json north_star):
HIP fp32 path within 1e-3 relative of this oracle, argmax masks identical.
"""
import torch
import torch.nn.functional as F


class OracleNet:
    """Functional model over a flat state_dict.

    sd        : dict name -> CPU tensor (parameters may require grad for train-mode oracles)
    training  : BN uses batch statistics and updates running stats (in ``sd``, in place)
    eps_encoder / eps_decoder : what segmentron/solver/optimizer.py:18-30 setattr's on every
                BatchNorm2d after construction (cfg.MODEL.BN_EPS_FOR_ENCODER / _DECODER);
                None -> torch default 1e-5.
    momentum  : cfg.MODEL.BN_MOMENTUM or torch default 0.1.
    drop_p    : ASPP Dropout2d / FCN-head Dropout probability (0 for parity runs).
    multi_dilation : cfg.MODEL.DANET.MULTI_DILATION (ResNet layer4 multi-grid, resnet.py:166-175).
    """

    def __init__(self, sd, training=False, eps_encoder=None, eps_decoder=None, momentum=None,
                 drop_p=0.1, output_stride=16, aux=False, nclass=19, multi_dilation=None,
                 norm="BN"):
        self.sd = sd
        # cfg.MODEL.BN_TYPE: "BN" or "GN" = nn.GroupNorm(min(32, C), C), eps 1e-5 — the eps setter
        # of solver/optimizer.py:10-11 only touches BatchNorm classes (modules/batch_norm.py:105-108)
        self.norm = norm
        self.training = training
        self.eps_encoder = 1e-5 if eps_encoder is None else eps_encoder
        self.eps_decoder = 1e-5 if eps_decoder is None else eps_decoder
        self.momentum = 0.1 if momentum is None else momentum
        self.drop_p = drop_p
        self.output_stride = output_stride
        self.aux = aux
        self.nclass = nclass
        self.multi_dilation = multi_dilation  # cfg.MODEL.DANET.MULTI_DILATION when MULTI_GRID

    # ------------------------------------------------------------------ primitives
    def _eps(self, prefix):
        return self.eps_encoder if prefix.startswith("encoder.") else self.eps_decoder

    def bn(self, x, prefix):
        """nn.BatchNorm2d forward: train = biased batch var for
        normalisation, unbiased for running_var; eval = running stats."""
        sd = self.sd
        if self.norm == "GN" and (prefix + ".running_mean") not in sd:
            # (the reference builds its heads without passing norm_layer — pspnet.py:23-25 —
            # so they keep BatchNorm2d under BN_TYPE 'GN': the state_dict tells which is which)
            C = x.shape[1]
            return F.group_norm(x, min(32, C), sd[prefix + ".weight"], sd[prefix + ".bias"], 1e-5)
        if self.training:
            sd[prefix + ".num_batches_tracked"] += 1
        return F.batch_norm(x, sd[prefix + ".running_mean"], sd[prefix + ".running_var"],
                            sd[prefix + ".weight"], sd[prefix + ".bias"], self.training,
                            self.momentum, self._eps(prefix))

    def conv(self, x, prefix, stride=1, padding=0, dilation=1, groups=1):
        return F.conv2d(x, self.sd[prefix + ".weight"], self.sd.get(prefix + ".bias"),
                        stride, padding, dilation, groups)

    # ------------------------------------------------------------------ blocks
    def separable_conv(self, x, prefix, stride=1, dilation=1, relu_first=True):
        """SeparableConv2d — segmentron/modules/basic.py:34-62.
        relu_first: ReLU(non-inplace) -> dw3x3 -> BN -> pw1x1 -> BN
        else      : dw3x3 -> BN -> ReLU -> pw1x1 -> BN -> ReLU
        depthwise padding = dilation (basic.py:38-40)."""
        p = prefix + ".block."
        c = x.shape[1]
        if relu_first:
            x = F.relu(x)
            x = self.conv(x, p + "depthwise", stride, dilation, dilation, groups=c)
            x = self.bn(x, p + "bn_depth")
            x = self.conv(x, p + "pointwise")
            x = self.bn(x, p + "bn_point")
        else:
            x = self.conv(x, p + "depthwise", stride, dilation, dilation, groups=c)
            x = F.relu(self.bn(x, p + "bn_depth"))
            x = self.conv(x, p + "pointwise")
            x = F.relu(self.bn(x, p + "bn_point"))
        return x

    def conv_bn_relu(self, x, prefix, stride=1, padding=0, dilation=1, groups=1, relu6=False):
        """_ConvBNReLU — segmentron/modules/basic.py:65-77."""
        x = self.bn(self.conv(x, prefix + ".conv", stride, padding, dilation, groups),
                    prefix + ".bn")
        return F.relu6(x) if relu6 else F.relu(x)

    def xception_block(self, x, prefix, stride=1, dilation=1, skip="conv", relu_first=True,
                       low_feat=False):
        """XceptionBlock — segmentron/models/backbones/xception.py:10-51."""
        sc1 = self.separable_conv(x, prefix + ".sep_conv1", 1, dilation, relu_first)
        sc2 = self.separable_conv(sc1, prefix + ".sep_conv2", 1, dilation, relu_first)
        res = self.separable_conv(sc2, prefix + ".sep_conv3", stride, dilation, relu_first)
        if skip == "conv":
            short = self.bn(self.conv(x, prefix + ".conv", stride), prefix + ".bn")
            out = res + short
        elif skip == "sum":
            out = res + x
        else:
            out = res
        return (out, sc2) if low_feat else out

    def xception65(self, x, prefix="encoder"):
        """Xception65 — segmentron/models/backbones/xception.py:54-165."""
        os_ = self.output_stride
        if os_ == 32:
            b3s, mid_d, exit_d, exit_s = 2, 1, (1, 1), 2
        elif os_ == 16:
            b3s, mid_d, exit_d, exit_s = 2, 1, (1, 2), 1
        elif os_ == 8:
            b3s, mid_d, exit_d, exit_s = 1, 2, (2, 4), 1
        else:
            raise NotImplementedError
        p = prefix + "."
        x = F.relu(self.bn(self.conv(x, p + "conv1", 2, 1), p + "bn1"))
        x = F.relu(self.bn(self.conv(x, p + "conv2", 1, 1), p + "bn2"))
        x = self.xception_block(x, p + "block1", stride=2)
        x, c1 = self.xception_block(x, p + "block2", stride=2, low_feat=True)
        x, c2 = self.xception_block(x, p + "block3", stride=b3s, low_feat=True)
        for i in range(4, 20):
            x = self.xception_block(x, p + "block%d" % i, dilation=mid_d, skip="sum")
        c3 = x
        x = self.xception_block(c3, p + "block20", stride=exit_s, dilation=exit_d[0])
        c4 = self.xception_block(x, p + "block21", dilation=exit_d[1], skip="none",
                                 relu_first=False)
        return c1, c2, c3, c4

    def aspp(self, x, prefix="head.aspp"):
        """_ASPP — segmentron/modules/module.py:32-77."""
        os_ = self.output_stride
        if os_ in (16, 32):
            dil = (6, 12, 18)
        elif os_ == 8:
            dil = (12, 24, 36)
        else:
            raise NotImplementedError
        p = prefix + "."
        pool = F.adaptive_avg_pool2d(x, 1)
        pool = F.relu(self.bn(self.conv(pool, p + "image_pooling.conv"), p + "image_pooling.bn"))
        pool = F.interpolate(pool, size=x.shape[2:], mode="bilinear", align_corners=True)
        x0 = F.relu(self.bn(self.conv(x, p + "aspp0.conv"), p + "aspp0.bn"))
        x1 = self.separable_conv(x, p + "aspp1", 1, dil[0], relu_first=False)
        x2 = self.separable_conv(x, p + "aspp2", 1, dil[1], relu_first=False)
        x3 = self.separable_conv(x, p + "aspp3", 1, dil[2], relu_first=False)
        x = torch.cat((pool, x0, x1, x2, x3), dim=1)
        x = F.relu(self.bn(self.conv(x, p + "conv"), p + "bn"))
        return F.dropout2d(x, self.drop_p, self.training)

    def fcn_head(self, x, prefix):
        """_FCNHead — segmentron/modules/module.py:13-26."""
        p = prefix + ".block."
        x = F.relu(self.bn(self.conv(x, p + "0", 1, 1), p + "1"))
        x = F.dropout(x, self.drop_p, self.training)
        return self.conv(x, p + "4")

    def deeplab_head(self, c4, c1, prefix="head"):
        """_DeepLabHead — segmentron/models/deeplabv3_plus.py:49-75 (USE_ASPP, ENABLE_DECODER)."""
        p = prefix + "."
        x = self.aspp(c4, p + "aspp")
        x = F.interpolate(x, c1.shape[2:], mode="bilinear", align_corners=True)
        c1 = self.conv_bn_relu(c1, p + "c1_block")
        x = torch.cat([x, c1], dim=1)
        x = self.separable_conv(x, p + "block.0", relu_first=False)
        x = self.separable_conv(x, p + "block.1", relu_first=False)
        return self.conv(x, p + "block.2")

    def deeplabv3_plus_xception65(self, x):
        """DeepLabV3Plus.forward — segmentron/models/deeplabv3_plus.py:33-46."""
        size = x.shape[2:]
        c1, _, c3, c4 = self.xception65(x)
        out = self.deeplab_head(c4, c1)
        outs = [F.interpolate(out, size, mode="bilinear", align_corners=True)]
        if self.aux:
            a = self.fcn_head(c3, "auxlayer")
            outs.append(F.interpolate(a, size, mode="bilinear", align_corners=True))
        return tuple(outs)


# ---------------------------------------------------------------------- ResNet / FCN / PSPNet
def _resnet_plan(self):
    os_ = self.output_stride
    if os_ == 32:
        return [1, 1], [2, 2]
    if os_ == 16:
        return [1, 2], [2, 1]
    if os_ == 8:
        return [2, 4], [1, 1]
    raise NotImplementedError


def _res_block(self, x, p, stride, dilation, previous_dilation):
    """BasicBlockV1b / BottleneckV1b — segmentron/models/backbones/resnet.py:9-81."""
    sd = self.sd
    identity = x
    if (p + ".conv3.weight") in sd:  # bottleneck: 1x1 -> 3x3(stride, dil) -> 1x1
        out = F.relu(self.bn(self.conv(x, p + ".conv1"), p + ".bn1"))
        out = F.relu(self.bn(self.conv(out, p + ".conv2", stride, dilation, dilation), p + ".bn2"))
        out = self.bn(self.conv(out, p + ".conv3"), p + ".bn3")
    else:  # basic: 3x3(stride, dil) -> 3x3(previous_dilation)
        out = F.relu(self.bn(self.conv(x, p + ".conv1", stride, dilation, dilation), p + ".bn1"))
        out = self.bn(self.conv(out, p + ".conv2", 1, previous_dilation, previous_dilation),
                      p + ".bn2")
    if (p + ".downsample.0.weight") in sd:
        identity = self.bn(self.conv(x, p + ".downsample.0", stride), p + ".downsample.1")
    return F.relu(out + identity)


def _res_layer(self, x, p, stride, dilation, multi_dilation=None):
    """ResNetV1._make_layer — resnet.py:147-181 (first block of a dilated stage: dilation/2;
    multi_dilation = cfg.MODEL.DANET.MULTI_DILATION when MULTI_GRID: block i of layer4 is dilated
    by multi_dilation[i % len], :166-175)."""
    first = multi_dilation[0] if multi_dilation else (1 if dilation in (1, 2) else 2)
    x = _res_block(self, x, p + ".0", stride, first, dilation)
    j = 1
    while (p + ".%d.conv1.weight" % j) in self.sd:
        d = multi_dilation[j % len(multi_dilation)] if multi_dilation else dilation
        x = _res_block(self, x, p + ".%d" % j, 1, d, dilation)
        j += 1
    return x


def _resnet(self, x, prefix="encoder"):
    """ResNetV1.forward — resnet.py:183-199 (plain and deep-stem variants)."""
    dil, strides = _resnet_plan(self)
    p = prefix + "."
    if (p + "conv1.0.weight") in self.sd:  # deep stem
        x = F.relu(self.bn(self.conv(x, p + "conv1.0", 2, 1), p + "conv1.1"))
        x = F.relu(self.bn(self.conv(x, p + "conv1.3", 1, 1), p + "conv1.4"))
        x = self.conv(x, p + "conv1.6", 1, 1)
    else:
        x = self.conv(x, p + "conv1", 2, 3)
    x = F.relu(self.bn(x, p + "bn1"))
    x = F.max_pool2d(x, 3, 2, 1)
    c1 = _res_layer(self, x, p + "layer1", 1, 1)
    c2 = _res_layer(self, c1, p + "layer2", 2, 1)
    c3 = _res_layer(self, c2, p + "layer3", strides[0], dil[0])
    c4 = _res_layer(self, c3, p + "layer4", strides[1], dil[1],
                    getattr(self, "multi_dilation", None))
    return c1, c2, c3, c4


def _fcn_resnet(self, x):
    """FCN.forward — segmentron/models/fcn.py:22-34."""
    size = x.shape[2:]
    _, _, c3, c4 = _resnet(self, x)
    outs = [F.interpolate(self.fcn_head(c4, "head"), size, mode="bilinear", align_corners=True)]
    if self.aux:
        outs.append(F.interpolate(self.fcn_head(c3, "auxlayer"), size, mode="bilinear",
                                  align_corners=True))
    return tuple(outs)


def _psp_head(self, c4, p="head"):
    """_PSPHead + PyramidPooling — pspnet.py:44-58, module.py:82-97 (logits at c4 resolution)."""
    hw = c4.shape[2:]
    feats = [c4]
    for i, o in enumerate((1, 2, 3, 6)):
        f = F.adaptive_avg_pool2d(c4, o)
        f = self.conv_bn_relu(f, p + ".psp.convs.%d" % i)
        feats.append(F.interpolate(f, hw, mode="bilinear", align_corners=True))
    y = torch.cat(feats, dim=1)
    y = F.relu(self.bn(self.conv(y, p + ".block.0", 1, 1), p + ".block.1"))
    y = F.dropout(y, self.drop_p, self.training)
    return self.conv(y, p + ".block.4")


def _pspnet_resnet(self, x):
    """PSPNet.forward + _PSPHead + PyramidPooling — pspnet.py:28-58, module.py:82-97."""
    size = x.shape[2:]
    _, _, c3, c4 = _resnet(self, x)
    y = _psp_head(self, c4, "head")
    outs = [F.interpolate(y, size, mode="bilinear", align_corners=True)]
    if self.aux:
        outs.append(F.interpolate(self.fcn_head(c3, "auxlayer"), size, mode="bilinear",
                                  align_corners=True))
    return tuple(outs)


# ---------------------------------------------------------------------- MobileNetV2
def _inverted_residual(self, x, p, stride, dilation, expand):
    """InvertedResidual — segmentron/modules/basic.py:139-163."""
    y, i = x, 0
    if expand:
        y = self.conv_bn_relu(y, p + ".conv.0", relu6=True)
        i = 1
    c = y.shape[1]
    y = self.conv_bn_relu(y, p + ".conv.%d" % i, stride, dilation, dilation, groups=c, relu6=True)
    y = self.bn(self.conv(y, p + ".conv.%d" % (i + 1)), p + ".conv.%d" % (i + 2))
    if stride == 1 and x.shape[1] == y.shape[1]:
        return x + y
    return y


def _mobilenet_v2(self, x, prefix="encoder"):
    """MobileNetV2 — segmentron/models/backbones/mobilenet.py:55-143 (dilation only on the first
    block of a dilated group, :125 vs :128)."""
    os_ = self.output_stride
    dil = {32: (1, 1), 16: (1, 2), 8: (2, 4)}[os_]
    setting = [[1, 16, 1, 1], [6, 24, 2, 2], [6, 32, 3, 2], [6, 64, 4, 2], [6, 96, 3, 1],
               [6, 160, 3, 2], [6, 320, 1, 1]]
    groups = [(setting[0:1], 1), (setting[1:2], 1), (setting[2:3], 1), (setting[3:5], dil[0]),
              (setting[5:], dil[1])]
    p = prefix + "."
    x = self.conv_bn_relu(x, p + "conv1", 2, 1, relu6=True)
    outs = []
    for gi, (sets, d) in enumerate(groups):
        j = 0
        for t, c, n, s_ in sets:
            stride = s_ if d == 1 else 1
            x = _inverted_residual(self, x, p + "block%d.%d" % (gi + 1, j), stride, d, t != 1)
            j += 1
            for _ in range(n - 1):
                x = _inverted_residual(self, x, p + "block%d.%d" % (gi + 1, j), 1, 1, t != 1)
                j += 1
        if gi > 0:
            outs.append(x)
    return tuple(outs)


def _deeplab_mobilenet(self, x):
    """DeepLabV3Plus.forward with USE_ASPP False / ENABLE_DECODER False
    (configs/cityscapes_deeplabv3_plus_mobilenet.yaml:21-23; deeplabv3_plus.py:66-75)."""
    size = x.shape[2:]
    _, _, _, c4 = _mobilenet_v2(self, x)
    y = self.separable_conv(c4, "head.block.0", relu_first=False)
    y = self.separable_conv(y, "head.block.1", relu_first=False)
    y = self.conv(y, "head.block.2")
    return (F.interpolate(y, size, mode="bilinear", align_corners=True),)


def _hr_block(self, x, p):
    """BasicBlock / Bottleneck of HRNet (always stride 1) — backbones/hrnet.py:25-95."""
    sd = self.sd
    if (p + ".conv3.weight") in sd:
        out = F.relu(self.bn(self.conv(x, p + ".conv1"), p + ".bn1"))
        out = F.relu(self.bn(self.conv(out, p + ".conv2", 1, 1), p + ".bn2"))
        out = self.bn(self.conv(out, p + ".conv3"), p + ".bn3")
    else:
        out = F.relu(self.bn(self.conv(x, p + ".conv1", 1, 1), p + ".bn1"))
        out = self.bn(self.conv(out, p + ".conv2", 1, 1), p + ".bn2")
    residual = x
    if (p + ".downsample.0.weight") in sd:
        residual = self.bn(self.conv(x, p + ".downsample.0"), p + ".downsample.1")
    return F.relu(out + residual)


def _hr_blocks(self, x, p):
    j = 0
    while (p + ".%d.conv1.weight" % j) in self.sd:
        x = _hr_block(self, x, p + ".%d" % j)
        j += 1
    return x


def _hr_down_chain(self, x, p, relu_last):
    """Sequential of (3x3 stride-2 conv, BN[, ReLU]) — hrnet.py:190-207 / :350-359."""
    k = 0
    while (p + ".%d.0.weight" % k) in self.sd:
        x = self.bn(self.conv(x, p + ".%d.0" % k, 2, 1), p + ".%d.1" % k)
        last = (p + ".%d.0.weight" % (k + 1)) not in self.sd
        if relu_last or not last:
            x = F.relu(x)
        k += 1
    return x


def _hr_module(self, xs, p):
    """HighResolutionModule.forward — hrnet.py:211-229."""
    sd = self.sd
    nb = len(xs)
    xs = [_hr_blocks(self, xs[i], p + ".branches.%d" % i) for i in range(nb)]
    if nb == 1:
        return xs
    fused = []
    for i in range(nb):
        q = p + ".fuse_layers.%d" % i
        if not any((q + ".%d.0.weight" % j) in sd or (q + ".%d.0.0.weight" % j) in sd
                   for j in range(nb)):
            break  # multi_scale_output=False: only row 0 exists
        y = None
        for j in range(nb):
            if j == i:
                t = xs[j]
            elif j > i:
                t = self.bn(self.conv(xs[j], q + ".%d.0" % j), q + ".%d.1" % j)
                t = F.interpolate(t, scale_factor=2 ** (j - i), mode="nearest")
            else:
                t = _hr_down_chain(self, xs[j], q + ".%d" % j, relu_last=False)
            y = t if y is None else y + t
        fused.append(F.relu(y))
    return fused


def _hr_stage(self, xs, p):
    m = 0
    while (p + ".%d.branches.0.0.conv1.weight" % m) in self.sd:
        xs = _hr_module(self, xs, p + ".%d" % m)
        m += 1
    return xs


def _hr_transition(self, prev, p, nb):
    """HighResolutionNet.forward's transition step — hrnet.py:438-458: a present transition is
    applied to the LAST (coarsest) previous branch, an absent one passes branch i through."""
    sd = self.sd
    out = []
    for i in range(nb):
        q = p + ".%d" % i
        if (q + ".0.weight") in sd:      # Sequential(conv3x3, bn, relu)
            out.append(F.relu(self.bn(self.conv(prev[-1], q + ".0", 1, 1), q + ".1")))
        elif (q + ".0.0.weight") in sd:  # chain of stride-2 convs
            out.append(_hr_down_chain(self, prev[-1], q, relu_last=True))
        else:
            out.append(prev[i])
    return out


def _hrnet(self, x, prefix="encoder"):
    """HighResolutionNet.forward — backbones/hrnet.py:429-480."""
    sd = self.sd
    p = prefix + "."
    x = F.relu(self.bn(self.conv(x, p + "conv1", 2, 1), p + "bn1"))
    x = F.relu(self.bn(self.conv(x, p + "conv2", 2, 1), p + "bn2"))
    x = _hr_blocks(self, x, p + "layer1")
    ys = [x]
    for s in (2, 3, 4):
        nb = 0
        while (p + "stage%d.0.branches.%d.0.conv1.weight" % (s, nb)) in sd:
            nb += 1
        xs = _hr_transition(self, ys, p + "transition%d" % (s - 1), nb)
        ys = _hr_stage(self, xs, p + "stage%d" % s)
    return tuple(ys)


def _hrnet_seg(self, x):
    """HighResolutionNet (model 'HRNet') + _HRNetHead — models/hrnet_seg.py:23-60."""
    size = x.shape[2:]
    ys = _hrnet(self, x)
    hw = ys[0].shape[2:]
    feats = [ys[0]] + [F.interpolate(t, size=hw, mode="bilinear", align_corners=False)
                       for t in ys[1:]]
    q = "hrnet_head.last_layer"
    y = torch.cat(feats, 1)
    y = F.relu(self.bn(self.conv(y, q + ".0"), q + ".1"))
    k = self.sd[q + ".3.weight"].shape[-1]
    y = self.conv(y, q + ".3", 1, 1 if k == 3 else 0)
    return [F.interpolate(y, size=size, mode="bilinear", align_corners=False)]


# ---------------------------------------------------------------------- DANet
def pam(self, x, p):
    """PAM_Module (position attention) — segmentron/modules/module.py:100-130."""
    B, C, H, W = x.shape
    q = self.conv(x, p + ".query_conv").view(B, -1, H * W).permute(0, 2, 1)
    k = self.conv(x, p + ".key_conv").view(B, -1, H * W)
    att = F.softmax(torch.bmm(q, k), dim=-1)
    v = self.conv(x, p + ".value_conv").view(B, -1, H * W)
    out = torch.bmm(v, att.permute(0, 2, 1)).view(B, C, H, W)
    return self.sd[p + ".gamma"] * out + x


def cam(self, x, p):
    """CAM_Module (channel attention) — segmentron/modules/module.py:133-162."""
    B, C, H, W = x.shape
    # (three separate views of x, as the reference: autograd then accumulates four gradient
    # terms at x in the reference's order — bit-identical backward)
    proj_query = x.view(B, C, -1)
    proj_key = x.view(B, C, -1).permute(0, 2, 1)
    energy = torch.bmm(proj_query, proj_key)
    energy_new = torch.max(energy, -1, keepdim=True)[0].expand_as(energy) - energy
    att = F.softmax(energy_new, dim=-1)
    proj_value = x.view(B, C, -1)
    out = torch.bmm(att, proj_value).view(B, C, H, W)
    return self.sd[p + ".gamma"] * out + x


def _danet_resnet(self, x):
    """DANet.forward + DANetHead — segmentron/models/danet.py:14-89 (three outputs: fused,
    position-attention and channel-attention heads)."""
    size = x.shape[2:]
    _, _, _, c4 = _resnet(self, x)
    h = "head."

    def cbr(t, name):
        return F.relu(self.bn(self.conv(t, h + name + ".0", 1, 1), h + name + ".1"))


    # (node creation order of danet.py:70-88: it fixes autograd's accumulation order)
    sa_conv = cbr(pam(self, cbr(c4, "conv5a"), h + "sa"), "conv51")
    sa_out = self.conv(F.dropout2d(sa_conv, self.drop_p, self.training), h + "conv6.1")
    sc_conv = cbr(cam(self, cbr(c4, "conv5c"), h + "sc"), "conv52")
    sc_out = self.conv(F.dropout2d(sc_conv, self.drop_p, self.training), h + "conv7.1")
    sasc_out = self.conv(F.dropout2d(sa_conv + sc_conv, self.drop_p, self.training),
                         h + "conv8.1")
    return tuple(F.interpolate(y, size, mode="bilinear", align_corners=True)
                 for y in (sasc_out, sa_out, sc_out))


OracleNet.danet_resnet = _danet_resnet


# ---------------------------------------------------------------------- Fast-SCNN
def _fast_scnn(self, x):
    """FastSCNN.forward and its four sub-modules — segmentron/models/fast_scnn.py:16-161.
    learning_to_downsample: 3x3 s2 conv WITHOUT padding (:75, `_ConvBNReLU(3, 32, 3, 2)`) + two
    stride-2 separable convs; global_feature_extractor: 3 x 3 inverted residuals (t = 6,
    strides 2, 2, 1) + PyramidPooling (module.py:82-97) + 1x1; feature_fusion: bilinear x4
    (align_corners) of the low-resolution branch -> 1x1+BN+ReLU -> 1x1(bias)+BN, plus
    1x1(bias)+BN of the high-resolution branch, ReLU of the sum; classifier: two separable convs
    + Dropout2d + 1x1(bias).  Aux heads (cfg.SOLVER.AUX): conv3x3+BN+ReLU+Dropout2d+1x1 on both
    branch outputs."""
    size = x.shape[2:]
    p = "learning_to_downsample."
    h = self.conv_bn_relu(x, p + "conv", 2, 0)
    h = self.separable_conv(h, p + "dsconv1", 2, 1, relu_first=False)
    hi = self.separable_conv(h, p + "dsconv2", 2, 1, relu_first=False)
    g = "global_feature_extractor."
    lo = hi
    for name, stride in (("bottleneck1", 2), ("bottleneck2", 2), ("bottleneck3", 1)):
        for j in range(3):
            lo = _inverted_residual(self, lo, g + "%s.%d" % (name, j), stride if j == 0 else 1,
                                    1, True)
    hw = lo.shape[2:]
    feats = [lo]
    for i, o in enumerate((1, 2, 3, 6)):
        f = self.conv_bn_relu(F.adaptive_avg_pool2d(lo, o), g + "ppm.convs.%d" % i)
        feats.append(F.interpolate(f, hw, mode="bilinear", align_corners=True))
    lo = self.conv_bn_relu(torch.cat(feats, dim=1), g + "out")
    f = "feature_fusion."
    up = F.interpolate(lo, scale_factor=4, mode="bilinear", align_corners=True)
    up = self.conv_bn_relu(up, f + "dwconv")
    up = self.bn(self.conv(up, f + "conv_lower_res.0"), f + "conv_lower_res.1")
    hr = self.bn(self.conv(hi, f + "conv_higher_res.0"), f + "conv_higher_res.1")
    y = F.relu(hr + up)
    c = "classifier."
    y = self.separable_conv(y, c + "dsconv1", 1, 1, relu_first=False)
    y = self.separable_conv(y, c + "dsconv2", 1, 1, relu_first=False)
    y = F.dropout2d(y, self.drop_p, self.training)
    outs = [F.interpolate(self.conv(y, c + "conv.1"), size, mode="bilinear", align_corners=True)]
    if self.aux:
        for name, feat in (("auxlayer1", hi), ("auxlayer2", lo)):
            a = F.relu(self.bn(self.conv(feat, name + ".0", 1, 1), name + ".1"))
            a = F.dropout2d(a, self.drop_p, self.training)
            outs.append(F.interpolate(self.conv(a, name + ".4"), size, mode="bilinear",
                                      align_corners=True))
    return tuple(outs)


OracleNet.fast_scnn = _fast_scnn
OracleNet.hrnet = _hrnet
OracleNet.hrnet_seg = _hrnet_seg
OracleNet.mobilenet_v2 = _mobilenet_v2
OracleNet.deeplab_mobilenet = _deeplab_mobilenet
OracleNet.resnet = _resnet
OracleNet.fcn_resnet = _fcn_resnet
OracleNet.pspnet_resnet = _pspnet_resnet


# ---------------------------------------------------------------------------------------------
# Criss-cross attention (CCNet).  The reference computes these with its CUDA extension
# (segmentron/modules/csrc/criss_cross_attention/ca_cuda.cu), which cannot be built here (CUDA);
# `cca_weight` / `cca_map` restate the forward kernels with tensor algebra (their backward kernels,
# ca_cuda.cu:39-94 and :125-177, are the analytic gradients — autograd supplies them), and
# `cca_weight_loops` / `cca_map_loops` follow the kernel source statement by statement (pure
# Python, small cases only) to pin the algebra: tests/test_cca.py.
def _cca_col_index(H, device):
    """idx[y, i] = i < y ? i : i + 1   (ca_cuda.cu:27-28): the column partner of entry W + i."""
    y = torch.arange(H, device=device).view(H, 1)
    i = torch.arange(H - 1, device=device).view(1, H - 1)
    return torch.where(i < y, i, i + 1)


def cca_weight(t, f):
    """ca_forward_kernel, ca_cuda.cu:8-37.  t, f: [N, C, H, W] -> energy [N, W + H - 1, H, W]."""
    N, C, H, W = t.shape
    row = torch.einsum("ncyx,ncyi->niyx", t, f)                 # z < W : f at (y, z)
    full = torch.einsum("ncyx,ncjx->nyxj", t, f)                # f at (j, x) for every j
    idx = _cca_col_index(H, t.device)                           # [H, H-1]
    col = torch.gather(full, 3, idx.view(1, H, 1, H - 1).expand(N, H, W, H - 1))
    return torch.cat([row, col.permute(0, 3, 1, 2)], dim=1)


def cca_map(weight, g):
    """ca_map_forward_kernel, ca_cuda.cu:96-123.  weight [N, W+H-1, H, W], g [N, C, H, W]."""
    N, C, H, W = g.shape
    out = torch.einsum("ncyi,niyx->ncyx", g, weight[:, :W])
    idx = _cca_col_index(H, g.device)
    wcol = weight[:, W:].permute(0, 2, 3, 1)                    # [N, y, x, i']
    full = torch.zeros(N, H, W, H, dtype=weight.dtype, device=weight.device)
    full = full.scatter(3, idx.view(1, H, 1, H - 1).expand(N, H, W, H - 1), wcol)  # [n,y,x,i]
    return out + torch.einsum("ncix,nyxi->ncyx", g, full)


def cca_weight_loops(t, f):
    N, C, H, W = t.shape
    L = H + W - 1
    w = torch.zeros(N, L, H, W, dtype=t.dtype)
    for b in range(N):
        for y in range(H):
            for x in range(W):
                for z in range(L):
                    for pl in range(C):
                        _t = t[b, pl, y, x]
                        if z < W:
                            w[b, z, y, x] += _t * f[b, pl, y, z]
                        else:
                            i = z - W
                            j = i if i < y else i + 1
                            w[b, W + i, y, x] += _t * f[b, pl, j, x]
    return w


def cca_map_loops(weight, g):
    N, C, H, W = g.shape
    out = torch.zeros_like(g)
    for b in range(N):
        for pl in range(C):
            for y in range(H):
                for x in range(W):
                    for i in range(W):
                        out[b, pl, y, x] += g[b, pl, y, i] * weight[b, i, y, x]
                    for i in range(H):
                        if i == y:
                            continue
                        j = i if i < y else i - 1
                        out[b, pl, y, x] += g[b, pl, i, x] * weight[b, W + j, y, x]
    return out


def _criss_cross_attention(self, x, p):
    """CrissCrossAttention.forward — segmentron/modules/cc_attention.py:60-72."""
    q = self.conv(x, p + ".query_conv")
    k = self.conv(x, p + ".key_conv")
    v = self.conv(x, p + ".value_conv")
    att = F.softmax(cca_weight(q, k), 1)
    return self.sd[p + ".gamma"] * cca_map(att, v) + x


def _ccnet_resnet(self, x, recurrence=2):
    """CCNet.forward + _CCHead + _RCCAModule — segmentron/models/ccnet.py:28-86."""
    size = x.shape[2:]
    _, _, c3, c4 = _resnet(self, x)
    p = "head.rcca."
    out = F.relu(self.bn(self.conv(c4, p + "conva.0", 1, 1), p + "conva.1"))
    for _ in range(recurrence):
        out = _criss_cross_attention(self, out, p + "cca")
    out = F.relu(self.bn(self.conv(out, p + "convb.0", 1, 1), p + "convb.1"))
    out = torch.cat([c4, out], dim=1)
    out = self.bn(self.conv(out, p + "bottleneck.0", 1, 1), p + "bottleneck.1")
    out = F.dropout2d(out, self.drop_p, self.training)
    out = self.conv(out, "head.out")
    outs = [F.interpolate(out, size, mode="bilinear", align_corners=True)]
    if self.aux:
        outs.append(F.interpolate(self.fcn_head(c3, "auxlayer"), size, mode="bilinear",
                                  align_corners=True))
    return tuple(outs)


OracleNet.ccnet_resnet = _ccnet_resnet


def mix_softmax_ce(outputs, target, aux_weight=0.4, ignore_index=-1):
    """MixSoftmaxCrossEntropyLoss — segmentron/solver/loss.py:16-46 (sum of per-output CE,
    aux outputs weighted by cfg.SOLVER.AUX_WEIGHT)."""
    loss = F.cross_entropy(outputs[0], target, ignore_index=ignore_index)
    for o in outputs[1:]:
        loss = loss + aux_weight * F.cross_entropy(o, target, ignore_index=ignore_index)
    return loss


def clone_state(sd, requires_grad=False):
    out = {}
    for k, v in sd.items():
        t = v.detach().clone()
        if requires_grad and t.is_floating_point() and "running_" not in k:
            t.requires_grad_(True)
        out[k] = t
    return out

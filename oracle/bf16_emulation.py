"""CPU emulation of the bf16 throughput path (TEST INFRASTRUCTURE ONLY).

Why this exists: the randomly-initialised DeepLabv3+/xception65 used for parity (no released
checkpoint is reachable) is a chaotic map — rounding only the conv WEIGHTS to bf16 in the CPU
oracle moves its logits by L2-rel 0.2-0.6 (probe recorded in DESIGN.md §Numerics), so "bf16 logits
vs fp32 logits" says nothing about kernel correctness.  This module restates the network with
bf16 rounding applied at exactly the points where the HIP bf16 path rounds, so the HIP result can
be compared at a tight tolerance; the distance between THIS and the fp32 oracle is the documented
cost of bf16 on this model.

Rounding points of the HIP bf16 path (segmentron_amd/csrc):
  * input image -> bf16 (seg_nchw_to_nhwc_pad)
  * 1x1 / dense conv (seg_conv_gemm_fwd): activation operand = bf16(act(raw)) staged in LDS,
    weights bf16, fp32 MFMA accumulation, raw output stored as bf16(acc [+ bias]); BN statistics
    from the fp32 accumulators (dense / strided convs) or, on the 256x128-tile kernel that runs
    every 1x1 stride-1 conv, from the values as stored (bf16-rounded; O >= 384 and >= 4096 output pixels, or O >= 256 and
    >= 65536 output pixels)
  * 1x1 conv whose input carries a LINEAR pending BN (no ReLU; csrc/fold.hip): operand = the raw
    bf16 tensor as stored, weights = bf16(W * scale), the constant W @ shift is dropped when a
    training-mode BN follows (it cancels) and added as an fp32 bias otherwise
  * depthwise (seg_dwconv3x3): operand act(raw) rounded to bf16 once by the LDS-tiled kernels
    (stride 1 with dilation 2, and stride 2 with dilation 1; the activated tile is parked in
    LDS in the storage dtype) and kept in fp32 by the register-sliding kernels (r06: stride 1,
    dilation 1 — csrc/dwconv_slide.hip keeps the activated window in fp32 registers) and by the
    strip kernels (wide dilations on maps too small for the row-chain kernel); weights fp32,
    fp32 accumulation, statistics from fp32, output stored bf16
  * BN finalize in fp64 -> fp32 scale/shift;  act(x) = relu(fma(x, scale, shift)) in fp32
  * materialise / residual add / bilinear / global pool: fp32 math, bf16 store
  * logits upsample: bf16 in, fp32 NCHW out
  * r05: ASPP image pooling / PSP pyramid bins: pooled values, their 1x1 conv and BatchNorm in
    float32, ONE bf16 rounding behind BN + ReLU (functional.global_avg_pool: with 2 samples per
    channel the normalised value is the sign of a difference that bf16 storage cannot hold)
  * r04: a BatchNorm over at most 1024 samples (hip_ops.SMALL_BN_ROWS: the ASPP image-pooling
    branch, PSP's pyramid bins, every layer of a tiny test input) takes its statistics two-pass
    from the tensor AS STORED (seg_bn_finalize_small), whatever kernel produced it
"""
import torch
import torch.nn.functional as F


SMALL_BN_ROWS = 1024  # == segmentron_amd.hip_ops.SMALL_BN_ROWS


class _R16(torch.autograd.Function):
    """round-to-bf16 with a straight-through gradient: the emulation is differentiable, so
    torch autograd of it is the backward pass of "the bf16 forward with exact gradient
    arithmetic" — same ReLU masks and BatchNorm statistics as the HIP bf16 path, whose own
    backward then differs only by the bf16 storage of its gradient tensors."""

    @staticmethod
    def forward(ctx, x):
        return x.to(torch.bfloat16).float()

    @staticmethod
    def backward(ctx, g):
        return g


def r16(x):
    return _R16.apply(x) if x.requires_grad else x.to(torch.bfloat16).float()


class _A:
    """raw (bf16-representable, NCHW fp32) + pending fp32 scale/shift + pending relu"""
    __slots__ = ("t", "s", "b", "relu")

    def __init__(self, t, s=None, b=None, relu=False):
        self.t, self.s, self.b, self.relu = t, s, b, relu

    def val(self):
        v = self.t
        if self.s is not None:
            v = torch.addcmul(self.b.view(1, -1, 1, 1), v, self.s.view(1, -1, 1, 1))
        if self.relu == 6:
            return torch.clamp(v, 0.0, 6.0)
        return torch.relu(v) if self.relu else v

    def with_relu(self):
        return _A(self.t, self.s, self.b, True)


class Bf16EmuNet:
    def __init__(self, sd, training=False, eps_encoder=1e-3, eps_decoder=1e-5, momentum=0.1,
                 output_stride=16, accum64=False, dw_round_operand=False):
        """dw_round_operand: also round the activated operand of the stride-1 / dilation-1
        depthwise convs (what the LDS-tiled kernels did until r05) — a second, equally valid
        bf16 pipeline; together with accum64 it samples the family of results the chaotic
        default fixture allows (tests/test_model_gpu.py takes the family's own spread as floor).
        accum64: run every convolution's accumulation in float64 (rounded once to fp32) with
        the SAME bf16 rounding points — the distance between accum64=False and True is the
        network's sensitivity to fp32 accumulation order, i.e. the floor below which two correct
        implementations of the bf16 path cannot be expected to agree on this chaotic net."""
        assert output_stride == 16
        self.accum64 = accum64
        self.dw_round_operand = dw_round_operand
        self.sd, self.training = sd, training
        self.eps_encoder, self.eps_decoder, self.momentum = eps_encoder, eps_decoder, momentum

    def _eps(self, p):
        return self.eps_encoder if p.startswith("encoder.") else self.eps_decoder

    def _bn(self, y, p, offset=None):
        """y: fp32 accumulators -> (scale, shift) like seg_bn_finalize / seg_bn_eval_affine.
        offset: per-channel constant left out of `y` (folded conv, see conv()): it re-enters the
        running_mean update only (seg_bn_finalize_p mean_offset)."""
        sd = self.sd
        g, b = sd[p + ".weight"], sd[p + ".bias"]
        eps = self._eps(p)
        if self.training:
            n = y.numel() // y.shape[1]
            yd = y.double()
            mean = yd.sum((0, 2, 3)) / n
            var = ((yd * yd).sum((0, 2, 3)) / n - mean * mean).clamp_min(0)
            invstd = 1.0 / torch.sqrt(var + eps)
            scale = (g.double() * invstd).float()
            shift = (b.double() - mean * g.double() * invstd).float()
            unb = var * n / (n - 1) if n > 1 else var
            m = self.momentum
            mo = mean if offset is None else mean + offset.double()
            sd[p + ".running_mean"] = ((1 - m) * sd[p + ".running_mean"].double() + m * mo).float()
            sd[p + ".running_var"] = ((1 - m) * sd[p + ".running_var"].double() + m * unb).float()
            return scale, shift
        invstd = 1.0 / torch.sqrt(sd[p + ".running_var"] + eps)
        return g * invstd, b - sd[p + ".running_mean"] * g * invstd

    def conv(self, a, p, bnp=None, stride=1, pad=0, dil=1):
        wf = self.sd[p + ".weight"]
        if a.s is not None and not a.relu and wf.shape[2:] == (1, 1) and (p + ".bias") not in self.sd:
            # folded linear BatchNorm (see module docstring)
            w = r16(wf * a.s.view(1, -1, 1, 1))
            conv = (lambda u, v: F.conv2d(u.double(), v.double(), None, stride).float()) \
                if self.accum64 else (lambda u, v: F.conv2d(u, v, None, stride))
            y = conv(a.t, w)
            keep_const = not (bnp is not None and self.training)
            if keep_const:
                y = y + (wf.view(wf.shape[0], -1) @ a.b).view(1, -1, 1, 1)
            if bnp is None:
                return _A(r16(y))
            m_ = y.shape[0] * y.shape[2] * y.shape[3]
            px256 = stride == 1 and ((w.shape[0] >= 384 and m_ >= 4096) or (w.shape[0] >= 256 and m_ >= 65536))
            const = None if keep_const else (wf.view(wf.shape[0], -1) @ a.b).detach()
            s, b = self._bn(r16(y) if (px256 or m_ <= SMALL_BN_ROWS) else y, bnp, const)
            return _A(r16(y), s, b)
        w = r16(wf)
        if self.accum64:
            y = F.conv2d(r16(a.val()).double(), w.double(), None, stride, pad, dil).float()
        else:
            y = F.conv2d(r16(a.val()), w, None, stride, pad, dil)
        bias = self.sd.get(p + ".bias")
        if bnp is None:
            return _A(r16(y if bias is None else y + bias.view(1, -1, 1, 1)))
        m_ = y.shape[0] * y.shape[2] * y.shape[3]
        O_, C_, kh = wf.shape[0], wf.shape[1], wf.shape[2]
        fast = (tuple(wf.shape[2:]) == (1, 1) and stride == 1 and pad == 0
                and ((O_ >= 384 and m_ >= 4096) or (O_ >= 256 and m_ >= 65536)))
        # stride-1 KxK on the direct-to-LDS pipeline (functional.conv_bn materialises a pending
        # BN / ReLU in front of every conv with O >= 256, so the operand is prologue-free) and the
        # direct halo-tile 3x3 kernel: statistics of the values AS STORED
        kxk = kh > 1 and stride == 1 and C_ % 32 == 0 and O_ >= 256 and m_ >= 4096 and bias is None
        direct = (kh == 3 and stride == 1 and pad == 1 and dil == 1 and m_ >= 65536
                  and (C_, O_) in ((32, 32), (32, 64), (64, 32), (16, 16)))
        s, b = self._bn(r16(y) if (fast or kxk or direct or m_ <= SMALL_BN_ROWS) else y, bnp)
        return _A(r16(y), s, b)

    def conv_f32(self, a, p, bnp):
        """1x1 conv + training/eval BatchNorm of a float32 operand in float32 (no rounding point):
        the few-row pooled branches since r05."""
        y = F.conv2d(a.val(), self.sd[p + ".weight"])
        s, b = self._bn(y, bnp)
        return _A(y, s, b)

    def dw(self, a, p, bnp, stride, dil):
        c = a.t.shape[1]
        v = a.val()
        if (stride == 1 and dil == 2) or (stride == 2 and dil == 1) or \
                (self.dw_round_operand and stride == 1 and dil == 1):
            v = r16(v)
        if self.accum64:
            y = F.conv2d(v.double(), self.sd[p + ".weight"].double(), None, stride, dil, dil,
                         groups=c).float()
        else:
            y = F.conv2d(v, self.sd[p + ".weight"], None, stride, dil, dil, groups=c)
        small = y.shape[0] * y.shape[2] * y.shape[3] <= SMALL_BN_ROWS
        s, b = self._bn(r16(y) if small else y, bnp)
        return _A(r16(y), s, b)

    def sep(self, a, p, stride=1, dil=1, relu_first=True):
        q = p + ".block."
        if relu_first:
            d = self.dw(a.with_relu(), q + "depthwise", q + "bn_depth", stride, dil)
            return self.conv(d, q + "pointwise", q + "bn_point")
        d = self.dw(a, q + "depthwise", q + "bn_depth", stride, dil)
        d.relu = True
        o = self.conv(d, q + "pointwise", q + "bn_point")
        o.relu = True
        return o

    def block(self, a, p, stride=1, dil=1, skip="conv", relu_first=True, low=False):
        s1 = self.sep(a, p + ".sep_conv1", 1, dil, relu_first)
        s2 = self.sep(s1, p + ".sep_conv2", 1, dil, relu_first)
        res = self.sep(s2, p + ".sep_conv3", stride, dil, relu_first)
        if skip == "conv":
            sh = self.conv(a, p + ".conv", p + ".bn", stride)
            out = _A(r16(res.val() + sh.val()))
        elif skip == "sum":
            out = _A(r16(res.val() + a.val()))
        else:
            out = res
        return (out, s2) if low else out

    def forward(self, x):
        size = x.shape[2:]
        e = "encoder."
        a = _A(r16(x))
        a = self.conv(a, e + "conv1", e + "bn1", 2, 1)
        a.relu = True
        a = self.conv(a, e + "conv2", e + "bn2", 1, 1)
        a.relu = True
        a = self.block(a, e + "block1", 2)
        a, c1 = self.block(a, e + "block2", 2, low=True)
        a, _ = self.block(a, e + "block3", 2, low=True)
        for i in range(4, 20):
            a = self.block(a, e + "block%d" % i, skip="sum")
        a = self.block(a, e + "block20")
        c4 = self.block(a, e + "block21", dil=2, skip="none", relu_first=False)
        logits = self.head(c4, c1)
        return F.interpolate(logits, size, mode="bilinear", align_corners=True)

    def aspp(self, c4, h="head.aspp."):
        """_ASPP on a deferred c4 (segmentron_amd/modules/module.py): c4 is materialised once."""
        xm = _A(r16(c4.val()))
        H, W = xm.t.shape[2:]
        # r05: the image-pooling branch runs in float32 (pooled vector, 1x1 conv, its N-sample
        # BatchNorm) and is rounded once behind BN + ReLU (functional.global_avg_pool)
        pooled = _A((xm.t.double().sum((2, 3), keepdim=True) / (H * W)).float())
        pa = self.conv_f32(pooled, h + "image_pooling.conv", h + "image_pooling.bn")
        pa.relu = True
        parts = [r16(r16(pa.val()).expand(-1, -1, H, W))]
        b0 = self.conv(xm, h + "aspp0.conv", h + "aspp0.bn")
        b0.relu = True
        parts.append(r16(b0.val()))
        for i, d in enumerate((6, 12, 18)):
            parts.append(r16(self.sep(xm, h + "aspp%d" % (i + 1), 1, d, False).val()))
        y = self.conv(_A(torch.cat(parts, 1)), h + "conv", h + "bn")
        y.relu = True
        return y

    def head(self, c4, c1, p="head."):
        """_DeepLabHead: ASPP -> x-up -> cat(c1_block(c1)) -> 2 SepConv -> classifier; returns
        the bf16-stored logits at c1 resolution (NCHW fp32 tensor)."""
        y = self.aspp(c4, p + "aspp.")
        up = r16(F.interpolate(y.val(), c1.t.shape[2:], mode="bilinear", align_corners=True))
        low = self.conv(c1, p + "c1_block.conv", p + "c1_block.bn")
        low.relu = True
        a = _A(torch.cat([up, r16(low.val())], 1))
        a = self.sep(a, p + "block.0", relu_first=False)
        a = self.sep(a, p + "block.1", relu_first=False)
        return self.conv(a, p + "block.2").t

    def inverted_residual(self, a, p, stride=1, dil=1, expand=True):
        """MobileNetV2 block (segmentron_amd/modules/basic.py InvertedResidual): expand 1x1 +
        BN + ReLU6 -> dw3x3 + BN + ReLU6 -> linear 1x1 + BN (+ x, materialised)."""
        x, i = a, 0
        if expand:
            a = self.conv(a, p + ".conv.0.conv", p + ".conv.0.bn")
            a.relu = 6
            i = 1
        a = self.dw(a, p + ".conv.%d.conv" % i, p + ".conv.%d.bn" % i, stride, dil)
        a.relu = 6
        out = self.conv(a, p + ".conv.%d" % (i + 1), p + ".conv.%d" % (i + 2))
        if stride == 1 and x.t.shape[1] == out.t.shape[1]:
            return _A(r16(out.val() + x.val()))
        return out

    # ------------------------------------------------------------------ ResNet / PSP / FCN / HRNet
    def res_block(self, a, p, stride=1, dil=1, prev_dil=1):
        """BottleneckV1b / BasicBlockV1b / HRNet's BasicBlock / Bottleneck
        (segmentron_amd/models/backbones/resnet.py, hrnet.py): conv+BN(+ReLU) deferred, one
        materialising pass relu(bn_last(conv_last) + identity) stored in bf16."""
        sd = self.sd
        if (p + ".conv3.weight") in sd:
            o = self.conv(a, p + ".conv1", p + ".bn1")
            o.relu = True
            o = self.conv(o, p + ".conv2", p + ".bn2", stride, dil, dil)
            o.relu = True
            o = self.conv(o, p + ".conv3", p + ".bn3")
        else:
            o = self.conv(a, p + ".conv1", p + ".bn1", stride, dil, dil)
            o.relu = True
            o = self.conv(o, p + ".conv2", p + ".bn2", 1, prev_dil, prev_dil)
        idn = a
        if (p + ".downsample.0.weight") in sd:
            idn = self.conv(a, p + ".downsample.0", p + ".downsample.1", stride)
        return _A(r16(torch.relu(o.val() + idn.val())))

    def head_tail(self, a, q):
        """3x3 conv + BN + ReLU (+ Dropout, p = 0 here) + 1x1 classifier with bias
        (segmentron_amd/modules/module.py head_tail): bf16-stored logits, NCHW fp32 tensor."""
        y = self.conv(a, q + ".0", q + ".1", 1, 1, 1)
        y.relu = True
        return self.conv(y, q + ".4").t

    def fcn_head(self, a, p):
        return self.head_tail(a, p + ".block")

    def psp_head(self, c4, p="head"):
        """PyramidPooling + _PSPHead (segmentron_amd/modules/module.py, models/pspnet.py): pooled
        bins stored bf16, 1x1 conv + BN over N*o*o samples + ReLU applied inside the bilinear
        upsample (bf16 store), concat, head_tail."""
        x = r16(c4.val())
        H, W = x.shape[2:]
        parts = [x]
        for i, o in enumerate((1, 2, 3, 6)):
            # r05: bins, 1x1 conv and its few-sample BatchNorm in float32, one rounding behind
            # BN + ReLU, then the bf16 bilinear upsample
            pooled = _A(F.adaptive_avg_pool2d(x.double(), o).float())
            a = self.conv_f32(pooled, p + ".psp.convs.%d.conv" % i, p + ".psp.convs.%d.bn" % i)
            a.relu = True
            parts.append(r16(F.interpolate(r16(a.val()), (H, W), mode="bilinear",
                                           align_corners=True)))
        return self.head_tail(_A(torch.cat(parts, 1)), p + ".block")

    def _hr_blocks(self, a, p):
        j = 0
        while (p + ".%d.conv1.weight" % j) in self.sd:
            a = self.res_block(a, p + ".%d" % j)
            j += 1
        return a

    def _hr_down_chain(self, a, p):
        k = 0
        while (p + ".%d.0.weight" % k) in self.sd:
            a = self.conv(a, p + ".%d.0" % k, p + ".%d.1" % k, 2, 1)
            a.relu = (p + ".%d.0.weight" % (k + 1)) in self.sd  # every link but the last
            k += 1
        return a

    def hr_module(self, xs, p):
        """HighResolutionModule.forward (segmentron_amd/models/backbones/hrnet.py): the fuse sum
        runs as one materialising pass per term — same-resolution terms first (identity and
        strided chains, in branch order), then the nearest-upsampled 1x1-conv terms — each
        stored in bf16, ReLU on the last."""
        sd = self.sd
        nb = len(xs)
        xs = [self._hr_blocks(xs[i], p + ".branches.%d" % i) for i in range(nb)]
        fused = []
        for i in range(nb):
            q = p + ".fuse_layers.%d" % i
            if not any((q + ".%d.0.weight" % j) in sd or (q + ".%d.0.0.weight" % j) in sd
                       for j in range(nb)):
                break
            same, ups = [], []
            for j in range(nb):
                if j == i:
                    same.append(xs[j])
                elif j < i:
                    same.append(self._hr_down_chain(xs[j], q + ".%d" % j))
                else:
                    ups.append((self.conv(xs[j], q + ".%d.0" % j, q + ".%d.1" % j), j - i))
            n_ops = max(len(same) - 1, 0) + len(ups)
            y, done = same[0], 0
            for t in same[1:]:
                done += 1
                v = y.val() + t.val()
                y = _A(r16(torch.relu(v) if done == n_ops else v))
            for up, sh in ups:
                done += 1
                v = y.val() + F.interpolate(up.val(), scale_factor=2 ** sh, mode="nearest")
                y = _A(r16(torch.relu(v) if done == n_ops else v))
            fused.append(y)
        return fused

"""GPU parity, teacher-forced composites at REALISTIC shapes (VERDICT r01 "what's weak" 1-2).

The whole-network oracle comparisons run at 65x129 .. 161x225, where the middle flow has
M = N*H*W < 4096 pixels: the dominant 256x128-tile GEMM, the multi-tile persistent depthwise
kernels and the split weight-gradient reduction never execute there, and on the random-init
(chaotic) network the bf16 path can only be bounded loosely.  Here every composite of
SURVEY.md §8(a15-a22) gets the ORACLE'S input at the shapes of the metric's configuration
(>= 65x129 x 728 channels, batch 2) and its forward output, input gradient and EVERY parameter
gradient are compared with

  * fp32 path : the float64 oracle (oracle/torch_ref.py, the reference graph): as accurate as
                the CPU float32 oracle itself (err <= 4 x its distance to float64 + 5e-4, half
                the north-star 1e-3; measured: forward <= 1e-6, gradients 2e-4 .. 2.5e-3 where
                the CPU fp32 path is at 4e-4 .. 1e-3 itself)
  * bf16 path : the float64-accumulating bf16 emulation (oracle/bf16_emulation.py: bf16 rounding
                at exactly the kernels' FORWARD rounding points, straight-through gradients):
                forward output within 1e-2 (L2, measured <= 4e-3); gradients within 1.5e-1
                (measured 1e-3 .. 1e-1, largest on the ASPP + decoder chain: the kernels also store every intermediate gradient in
                bf16 and train-mode BatchNorm backward amplifies that rounding, which the
                emulation's exact-arithmetic backward does not model).  A structural error — a
                wrong tile, a dropped term, a stale statistic — moves these numbers to O(1).
                For context the emulation itself sits 7e-2 .. 2e-1 from the un-rounded float64
                oracle on the gradients (ReLU-mask flips of pre-activations within bf16 rounding
                of zero), so bf16 gradients cannot be compared with float64 directly.
"""
import pytest
import torch

from _util import DEV, quant, rnd, to_cpu_nchw, to_dev_nhwc
from oracle import synth, torch_ref
from oracle.bf16_emulation import Bf16EmuNet, _A, r16

pytestmark = pytest.mark.gpu
DTYPES = [torch.float32, torch.bfloat16]
IDS = ["fp32", "bf16"]
PFX = "encoder.m"          # "encoder." prefix: the oracle applies eps_encoder = 1e-3


def _l2(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()


def _mx(a, b):
    return ((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30)).item()


def _setup(module, seed):
    """synthetic parameters by key name; BN eps as solver/optimizer.py:18-20 sets it"""
    sd = synth.synth_like(module.state_dict(), seed=seed)
    module.load_state_dict(sd)
    for m in module.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.eps = 1e-3
    return module.to(DEV).train(), sd


def _ref_state(sd, dtype):
    out = {}
    for k, v in sd.items():
        t = v.to(dtype) if v.is_floating_point() else v.clone()
        if t.is_floating_point() and "running_" not in k:
            t.requires_grad_()
        out[PFX + "." + k] = t
    return out


def _compare(name, dtype, got, ref64, emu, bars, ref32=None):
    """got/ref64/emu/ref32: dict tensor-name -> CPU tensor.  bars = (fp32 bar, bf16-vs-emulation
    bar, bf16-vs-fp64 sanity bar) on the L2-relative error.  fp32: the HIP result must be as
    close to the float64 oracle as the CPU float32 oracle is (x4) + the bar — train-mode BatchNorm
    backward chains amplify fp32 rounding to 1e-4..1e-3 in ANY evaluation order (the CPU fp32
    oracle's own dx sits 4e-4 from float64 on the 728-channel block)."""
    worst = 0.0
    stats, fails = {}, []
    # gradients that are identically zero in exact arithmetic (the bias of a BatchNorm / conv
    # in front of a training-mode BatchNorm) come out as 0 here and as 1e-13 noise in the
    # float64 oracle: measure every parameter gradient against max(its norm, 1e-6 x the largest
    # parameter-gradient norm) instead of dividing noise by noise
    gmax = max(v.double().norm().item() for k, v in ref64.items() if k.startswith("d:"))

    def _l2(a, b, key=""):
        den = b.double().norm().item()
        if key.startswith("d:"):
            den = max(den, 1e-6 * gmax)
            # BatchNorm (gamma, beta): a BN -> ReLU -> depthwise conv -> train-mode BN chain is
            # almost invariant to gamma (exactly for beta = 0: relu(g*x) = g*relu(x) and the next
            # BN divides the scale out again), so dgamma is the tiny residual of a cancelling
            # sum — measure both gradients of a layer against the larger of their two norms
            pat = SIBLINGS.get(name, {}).get(key)
            if pat is not None:
                den = max([den] + [ref64[pat % i].double().norm().item() for i in range(8)
                                   if (pat % i) in ref64])
            for a_, b_ in ((".weight", ".bias"), (".bias", ".weight")):
                sib = key[:-len(a_)] + b_ if key.endswith(a_) else None
                if sib in ref64 and ref64[sib].dim() == 1 and ref64[key].dim() == 1:
                    den = max(den, ref64[sib].double().norm().item())
        return (a.double() - b.double()).norm().item() / max(den, 1e-30)

    for k, g in got.items():
        assert torch.isfinite(g).all(), "%s %s: non-finite" % (name, k)
        e64 = _l2(g, ref64[k], k)
        if dtype == torch.float32:
            floor = _l2(ref32[k], ref64[k], k)
            assert e64 <= 4 * floor + bars[0], (
                "%s %s fp32: L2-rel %.3e vs fp64 oracle (CPU fp32 oracle: %.3e)" % (name, k, e64,
                                                                                      floor))
            worst = max(worst, e64 / (4 * floor + bars[0]))
        else:
            # (a parameter the emulation never touches — the bias of a BatchNorm folded away in
            # front of a training-mode BatchNorm — has no gradient there: exactly zero)
            e_k = emu[k] if k in emu else torch.zeros_like(ref64[k])
            ee = _l2(g, e_k, k)
            floor = _l2(e_k, ref64[k], k)
            # bars: output | data / weight gradients (dim >= 2) | BatchNorm gamma / beta and conv
            # biases (1-d: residuals of cancelling sums, see _l2 above)
            bar = bars[1] if k == "y" else (bars[3] if (k.startswith("dx") or ref64[k].dim() >= 2)
                                            else bars[2])
            cls = "y" if k == "y" else ("dx/dW" if (k.startswith("dx") or ref64[k].dim() >= 2)
                                        else "gamma/beta")
            stats[cls] = max(stats.get(cls, 0.0), ee)
            if SIBLINGS.get(name, {}).get(k) is not None or k in SENSITIVE.get(name, ()):
                # ill-conditioned by construction: its own sensitivity to bf16 rounding (the
                # emulation's distance to fp64, same normalisation) sets the bar
                bar = max(bar, 2.0 * floor + bars[3])
            if ee > bar:
                fails.append("%s %s bf16: L2-rel %.3e vs bf16 emulation > %.1e (max-normalised "
                             "%.3e; emulation itself is %.3e from fp64)"
                             % (name, k, ee, bar, _mx(g, e_k), floor))
            if e64 > 2.0 * floor + bar:
                fails.append("%s %s bf16: L2-rel %.3e vs fp64 oracle (emulation: %.3e)"
                             % (name, k, e64, floor))
            worst = max(worst, ee)
    if stats:
        print("PARITY composite %s bf16 L2-rel vs emulation: %s" % (
            name, ", ".join("%s %.2e" % kv for kv in sorted(stats.items()))))
    assert not fails, "\n".join(fails)
    return worst


def _grads(params, prefix=PFX + "."):
    return {k[len(prefix):]: v.grad.detach().float() for k, v in params.items()
            if v.is_leaf and v.grad is not None}


def _hip_run(F, module_fn, inputs, dy, dtype, params):
    """inputs: list of CPU NCHW tensors -> (y NCHW, [dx NCHW], {param: grad})."""
    xs = [to_dev_nhwc(x.detach(), dtype).detach().requires_grad_() for x in inputs]
    y = module_fn(*[F.Act(x) for x in xs])
    y.backward(to_dev_nhwc(dy, dtype))
    out = {"y": to_cpu_nchw(y)}
    for i, x in enumerate(xs):
        out["dx%d" % i] = to_cpu_nchw(x.grad)
    for k, p in params:
        assert p.grad is not None, k
        out["d:" + k] = p.grad.detach().float().cpu()
    return out


def _oracle_run(fn, inputs, dy, dtype_ref, dtype=None):
    """dy None: draw the output gradient (representable in `dtype`) once the shape is known."""
    xs = [x.detach().clone().to(dtype_ref).requires_grad_() for x in inputs]
    y = fn(*xs)
    if dy is None:
        dy = quant(rnd(tuple(y.shape), 9), dtype)
    y.backward(dy.to(y.dtype))
    out = {"y": y.detach().float(), "_dy": dy}
    for i, x in enumerate(xs):
        out["dx%d" % i] = x.grad.float()
    return out


# bf16 data / weight gradients vs the emulation: 3e-2 unless listed here with
# the measured reason
DXDW_BAR = {
    # three stacked separable convs (ReLU after each BN) resp. ASPP + decoder + classifier: every
    # intermediate gradient is stored in bf16 and passes 6-9 train-mode BatchNorm backwards; the
    # emulation's backward is exact arithmetic.  Measured r03: dx 3.7e-2 / 5.1e-2, every dW below
    # 3e-2 / 5.6e-2 (profiles/r03_parity.txt)
    "xception_exit_1536_2048": 5e-2,
    # (r05, image-pooling branch in float32 on both sides: 7.0e-2 .. 7.6e-2 on every tensor
    # alike, 8.6e-2 on the image-pooling weight — the forward agrees to 5.9e-3 and the emulation
    # itself sits 0.16-0.20 from the fp64 oracle on this random-init head: ReLU-mask noise of the
    # decoder, not a term of the backward; the ASPP alone agrees to 5e-3, aspp_2048 below)
    "deeplab_head": 1e-1,
    # two BasicBlocks per branch + the cross-resolution fuse: 16-channel tensors at 256x512, four
    # bf16-stored gradient hops per weight; measured 3.4e-2 on branches.0.0.conv2.weight
    "hr_module_4": 5e-2,
}
# Gradients that are residuals of an (almost) exact cancellation and are therefore measured
# against the norm of their siblings: PSP's bin-1 branch normalises N*1*1 = 2 samples per channel
# in training mode — BatchNorm of two samples is +-gamma/sqrt(1 + eps/var) + beta whatever the
# convolution computed, so d loss / d conv weight carries the factor eps / (delta^2 + eps) with
# delta the difference of two bf16-rounded pooled responses: one flipped rounding of a pooled
# input moves it by tens of percent (the emulation itself sits 0.39 from the fp64 oracle there by
# its own norm, 0.11 by its siblings').  Bar for it: 2 x that floor + the dx/dW bar.
SIBLINGS = {"psp_head_2048": {"d:psp.convs.0.conv.weight": "d:psp.convs.%d.conv.weight"}}
# The other pyramid branches normalise N*o*o = 8 / 18 / 72 samples per channel: the same
# mechanism, weaker (the bf16 emulation sits 0.27 / 0.24 from the fp64 oracle on bins 2 and 3;
# measured HIP-vs-emulation 0.16 / 0.10).  Same sensitivity-based bar, by their own norms.
SENSITIVE = {"psp_head_2048": ("d:psp.convs.1.conv.weight", "d:psp.convs.2.conv.weight",
                               "d:psp.convs.3.conv.weight")}

CASES = ["sep_relu_first_728", "sep_relu_last_1536", "sep_stride2_256_728",
         "xception_middle_728", "xception_entry_conv_256_728", "xception_exit_1536_2048",
         "inverted_residual_32", "aspp_2048", "deeplab_head",
         # the block types of C1 / C4 / C5 at BASELINE-size feature maps
         "bottleneck_dilated_2048", "psp_head_2048", "fcn_head_2048", "hr_module_4"]


@pytest.mark.parametrize("dtype", DTYPES, ids=IDS)
@pytest.mark.parametrize("case", CASES)
def test_composite_teacher_forced(case, dtype, c3_cfg):
    import segmentron_amd
    from segmentron_amd import functional as F
    from segmentron_amd.models.backbones.xception import XceptionBlock
    from segmentron_amd.models.deeplabv3_plus import _DeepLabHead
    from segmentron_amd.modules import InvertedResidual, SeparableConv2d, _ASPP
    segmentron_amd.set_compute_dtype(dtype)
    torch.manual_seed(0)
    N, H, W = 2, 65, 129       # C3's /16 feature map at 1025x2049: M = 16770 pixels
    # fp32 | bf16 y | bf16 gamma / beta / bias | bf16 dx / dW
    bars = (5e-4, 1e-2, 1.5e-1, DXDW_BAR.get(case, 3e-2))

    def act_in(shape, seed, relu_like=False):
        x = rnd(shape, seed) * 1.2 + 0.1
        return quant(x, dtype)

    if case.startswith("sep_"):
        cin, cout, stride, dil, relu_first = {
            "sep_relu_first_728": (728, 728, 1, 1, True),
            "sep_relu_last_1536": (1536, 1536, 1, 2, False),
            "sep_stride2_256_728": (256, 728, 2, 1, True)}[case]
        h, w = (H, W) if stride == 1 else (129, 257)
        mod, sd = _setup(SeparableConv2d(cin, cout, stride=stride, dilation=dil,
                                         relu_first=relu_first), 3)
        inputs = [act_in((N, cin, h, w), 1)]
        hip_fn = lambda a: F.materialize(mod(a))
        ora = lambda net: (lambda x: net.separable_conv(x, PFX, stride, dil, relu_first))
        emu = lambda net: (lambda x: r16(net.sep(_A(x), PFX, stride, dil, relu_first).val()))
    elif case.startswith("xception_"):
        ch, stride, dil, skip, relu_first, h, w = {
            "xception_middle_728": ([728] * 4, 1, 1, "sum", True, H, W),
            "xception_entry_conv_256_728": ([256, 728, 728, 728], 2, 1, "conv", True, 129, 257),
            "xception_exit_1536_2048": ([1024, 1536, 1536, 2048], 1, 2, "none", False, 33, 65),
        }[case]
        mod, sd = _setup(XceptionBlock(ch, stride=stride, dilation=dil,
                                       skip_connection_type=skip, relu_first=relu_first), 4)
        inputs = [act_in((N, ch[0], h, w), 2)]
        hip_fn = lambda a: F.materialize(mod(a))
        ora = lambda net: (lambda x: net.xception_block(x, PFX, stride, dil, skip, relu_first))
        emu = lambda net: (lambda x: r16(net.block(_A(x), PFX, stride, dil, skip, relu_first).val()))
    elif case == "inverted_residual_32":
        mod, sd = _setup(InvertedResidual(32, 32, 1, 6), 5)
        inputs = [act_in((N, 32, 129, 257), 3)]
        hip_fn = lambda a: F.materialize(mod(a))
        ora = lambda net: (lambda x: torch_ref._inverted_residual(net, x, PFX, 1, 1, True))
        emu = lambda net: (lambda x: r16(net.inverted_residual(_A(x), PFX).val()))
    elif case == "aspp_2048":
        mod, sd = _setup(_ASPP(2048, 256), 6)
        mod.dropout.p = 0.0
        inputs = [act_in((N, 2048, 33, 65), 4)]
        hip_fn = lambda a: F.materialize(mod(a)[0])
        ora = lambda net: (lambda x: net.aspp(x, PFX))
        emu = lambda net: (lambda x: r16(net.aspp(_A(x), PFX + ".").val()))
    elif case == "bottleneck_dilated_2048":
        # ResNet-101 layer4 block at OS8 (resnet.py:50-81): 2048 -> 512 -> 3x3 dil 4 -> 2048,
        # identity skip, at C4's 1025x2049 / 8 map (conv2 on the direct-to-LDS KxK kernels)
        from segmentron_amd.models.backbones.resnet import BottleneckV1b
        mod, sd = _setup(BottleneckV1b(2048, 512, 1, 4, None, 4), 8)
        inputs = [torch.relu(act_in((N, 2048, 65, 129), 7))]
        hip_fn = lambda a: F.materialize(mod(a))
        ora = lambda net: (lambda x: torch_ref._res_block(net, x, PFX, 1, 4, 4))
        emu = lambda net: (lambda x: net.res_block(_A(x), PFX, 1, 4, 4).val())
    elif case == "psp_head_2048":
        # PyramidPooling + _PSPHead (pspnet.py:44-58, module.py:82-97) on a 2 x 2048 x 65 x 129 c4
        from segmentron_amd.models.pspnet import _PSPHead
        mod, sd = _setup(_PSPHead(19), 9)
        mod.block[3].p = 0.0
        inputs = [torch.relu(act_in((N, 2048, 65, 129), 8))]
        hip_fn = lambda a: mod(a)
        ora = lambda net: (lambda x: torch_ref._psp_head(net, x, PFX))
        emu = lambda net: (lambda x: net.psp_head(_A(x), PFX))
    elif case == "fcn_head_2048":
        from segmentron_amd.modules import _FCNHead
        mod, sd = _setup(_FCNHead(2048, 19), 10)
        mod.block[3].p = 0.0
        inputs = [torch.relu(act_in((N, 2048, 65, 129), 9))]
        hip_fn = lambda a: mod(a)
        ora = lambda net: (lambda x: net.fcn_head(x, PFX))
        emu = lambda net: (lambda x: net.fcn_head(_A(x), PFX))
    elif case == "hr_module_4":
        # stage4 of hrnet_w18_small_v1 (4 branches 16/32/64/128, BASIC x 2) at C5's 1024x2048 / 4
        # top resolution; the four fused outputs are compared as one concatenated vector each
        from segmentron_amd.models.backbones import hrnet as HR
        ch = [16, 32, 64, 128]
        mod, sd = _setup(HR.HighResolutionModule(4, HR.BasicBlock, [2] * 4, list(ch), list(ch),
                                                 "SUM", True), 11)
        inputs = [torch.relu(act_in((N, c, 256 >> i, 512 >> i), 10 + i)) for i, c in enumerate(ch)]

        def flat(ts):  # one NCHW "image" [N, sum_i C_i*H_i*W_i, 1, 1]
            return torch.cat([t.reshape(t.shape[0], -1) for t in ts], 1)[:, :, None, None]

        class _DenseGrad(torch.autograd.Function):  # the kernels take dense NHWC gradients
            @staticmethod
            def forward(ctx, t):
                return t.view_as(t)

            @staticmethod
            def backward(ctx, g):
                return g.contiguous()

        def hip_fn(*acts):
            ys = mod(list(acts))
            return torch.cat([_DenseGrad.apply(F.materialize(y)).permute(0, 3, 1, 2).reshape(N, -1)
                              for y in ys], 1)[:, None, None, :]
        ora = lambda net: (lambda *xs: flat(torch_ref._hr_module(net, list(xs), PFX)))
        emu = lambda net: (lambda *xs: flat([y.val() for y in
                                             net.hr_module([_A(x) for x in xs], PFX)]))
    else:  # deeplab_head: ASPP + decoder + classifier at C3's c4 / c1 shapes (513x1025 input)
        mod, sd = _setup(_DeepLabHead(19, 256, 2048), 7)
        mod.aspp.dropout.p = 0.0
        inputs = [act_in((N, 2048, 33, 65), 5), act_in((N, 256, 129, 257), 6)]
        hip_fn = lambda a, c1: mod(a, c1)
        ora = lambda net: (lambda x, c1: net.deeplab_head(x, c1, PFX))
        emu = lambda net: (lambda x, c1: net.head(_A(x), _A(c1), PFX + "."))

    # float64 oracle (the reference graph)
    s64 = _ref_state(sd, torch.float64)
    net64 = torch_ref.OracleNet(s64, training=True, eps_encoder=1e-3, drop_p=0.0)
    ref = _oracle_run(ora(net64), inputs, None, torch.float64, dtype)
    dy = ref.pop("_dy")
    for k, v in _grads(s64).items():
        ref["d:" + k] = v
    emu_out = ref32 = None
    if dtype == torch.float32:
        s32 = _ref_state(sd, torch.float32)
        net32 = torch_ref.OracleNet(s32, training=True, eps_encoder=1e-3, drop_p=0.0)
        ref32 = _oracle_run(ora(net32), inputs, dy, torch.float32)
        ref32.pop("_dy")
        for k, v in _grads(s32).items():
            ref32["d:" + k] = v
    if dtype == torch.bfloat16:
        s32 = _ref_state(sd, torch.float32)
        enet = Bf16EmuNet(s32, training=True, eps_encoder=1e-3, eps_decoder=1e-3, accum64=True)
        emu_out = _oracle_run(emu(enet), inputs, dy, torch.float32)
        emu_out.pop("_dy")
        for k, v in _grads(s32).items():
            emu_out["d:" + k] = v
    got = _hip_run(F, hip_fn, inputs, dy, dtype, list(mod.named_parameters()))
    assert set(got) == set(ref), (sorted(set(got) ^ set(ref)))
    worst = _compare(case, dtype, got, ref, emu_out, bars, ref32)
    print("%s %s: worst %s %.3e over %d tensors (y, dx, parameter gradients)"
          % (case, IDS[DTYPES.index(dtype)], "ratio to the fp32 bound" if dtype == torch.float32
             else "L2-rel vs the bf16 emulation", worst, len(got)))
    # running statistics went through the same finalize
    msd = mod.state_dict()
    for k in sd:
        if k.endswith("running_var") or k.endswith("running_mean"):
            r = s64[PFX + "." + k].float()
            assert _l2(msd[k].cpu(), r) < (1e-4 if dtype == torch.float32 else 2e-2), k

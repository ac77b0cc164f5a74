#!/usr/bin/env python3
"""bench.py — images/sec, forward+backward, DeepLabv3+ xception65 @1025x2049 on N MI355X.

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

One step = what tools/train.py:135-146 does per iteration on one batch: model(images) ->
cross-entropy (ignore -1) -> backward (+ DDP gradient all-reduce and SyncBN statistics
all-reduces when N > 1) -> SGD step, on BASELINE.json configs[2]: batch 2 per GPU, synthetic
`randn(2,3,1025,2049)` images and `randint(0,19)` labels with 5 % ignore, random
init weights, bf16 compute path (fp32 accumulation / statistics / master weights / logits).
Inputs are resident in HBM before the timed region.  Prints ONE JSON line on rank 0.
`python bench.py --gpus N` without a torchrun environment re-launches itself under
`python -m torch.distributed.run` with N ranks (one per GPU, RCCL).

Launch path: the step issues ~1400 kernels, and issued one by one from Python the host is the
bottleneck (BENCH_r01: 39 ms/step against 33 ms of kernels).  With one GPU the whole step
(forward, loss, backward, SGD) is therefore captured ONCE into a HIP graph (torch.cuda.CUDAGraph
over the C-ABI launches — every entry point is capture-safe: no allocation, no host sync) and
the timed region replays it; `"launch": "hip_graph"` in the line says so, `--no-graph` /
a failed capture falls back to eager launches.  N > 1 runs eager under DDP.

roofline: the dominant kernel is the MFMA implicit-GEMM `conv_gemm_glds_kernel` (direct-to-LDS
256x256 / 192x256 tiles, eight waves; r06: + `conv_gemm_glds4_kernel`, 224x256 tiles on four waves,
for the launches of at most two rounds), csrc/conv_gemm_glds.hip / conv_gemm_glds4.hip: every 1x1 stride-1 convolution with O >= 384 whose input needs no prologue,
forward + data gradient: the Xception middle / exit flow — ~110 of the ~157 GEMM launches per
step).  `achieved` = algorithmic FLOPs
(2 * output pixels * K * O per launch — SURVEY.md §8d counts conv MACs only) summed over its
launches / summed launch durations measured with HIP events on the launch stream — in eager
steps run right after the timed region when that replays a graph (events cannot bracket a node
of a captured graph); the figure over ALL forward/dgrad GEMM launches is reported beside it
(`all_gemm_*`).  `gpu_busy_frac` = kernel durations / wall span of a rocprofv3 kernel trace of the
replayed step (profiles/rocprof_roofline.json, tools/rocprof_roofline.py; null without one);
`eager_profile_kernel_ms_per_step` = kernel sum of ONE eager step under torch.profiler.  `traffic` is filled from profiles/ (rocprofv3 PMC
pass of this build, see profiles/traffic.json "source") when available.

cpu_baseline: the CPU oracle (oracle/torch_ref.py — bit-identical to the reference's module graph
on CPU, see oracle/gen_golden.py; the reference tree itself is not on the GPU box) timed on the
host cores of rank 0 as BASELINE.md §3 plans it: the SAME train step at the full 1025x2049,
batch 2, fp32, 1 warm-up + 2 timed steps (<= 32 threads: oneDNN collapses when oversubscribed on
the 256-core host); `--cpu-baseline-size HxW` bounds it for quick runs.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

H, W, BATCH = 1025, 2049, 2
FLOP_FWD_BWD_PER_IMAGE = 2.50935e12   # conv FLOPs, 3x fwd - first-layer dgrad
MFMA_BF16_PEAK = 2.5e15               # MI355X dense bf16 (MI355X_MICROARCH.md)

C3 = ["DATASET.NAME", "cityscape", "TRAIN.BATCH_SIZE", str(BATCH), "TRAIN.CROP_SIZE", "769",
      "TEST.CROP_SIZE", "(1025, 2049)", "SOLVER.LR", "0.02", "MODEL.MODEL_NAME", "DeepLabV3_Plus",
      "MODEL.BACKBONE", "xception65", "MODEL.BN_EPS_FOR_ENCODER", "1e-3",
      "TRAIN.BACKBONE_PRETRAINED", "False"]
_COMMON = ["DATASET.NAME", "cityscape", "TRAIN.BACKBONE_PRETRAINED", "False"]
# `--config`: the default c3 is BASELINE.json's metric; c2 / c4 / c5 are its other GPU configs
# (configs[1], [3], [4]) emitted in the same JSON schema so that they can be driver-timed too.
# A "step" of an inference config is one forward pass over one batch (torch.no_grad, eval mode).
CONFIGS = {
    "c3": dict(over=C3, batch=BATCH, h=H, w=W, train=True, aux=False, oracle="deeplabv3_plus_xception65",
               metric="images/sec fwd+bwd DeepLabv3+_xception65 @1025x2049",
               workload="DeepLabv3+_xception65 train step (fwd + CE loss + bwd + SGD)",
               tag="BASELINE.json configs[2]"),
    "c4": dict(over=_COMMON + ["MODEL.MODEL_NAME", "PSPNet", "MODEL.BACKBONE", "resnet101",
                               "MODEL.OUTPUT_STRIDE", "8", "SOLVER.AUX", "True", "SOLVER.LR", "0.01"],
               batch=2, h=1025, w=2049, train=True, aux=True, oracle="pspnet_resnet", os=8,
               metric="images/sec fwd+bwd PSPNet_resnet101 @1025x2049",
               workload="PSPNet_resnet101 (OS8, aux head) train step (fwd + CE + 0.4 aux CE + bwd "
                        "+ SGD)", tag="BASELINE.json configs[3]"),
    "c2": dict(over=_COMMON + ["MODEL.MODEL_NAME", "DeepLabV3_Plus", "MODEL.BACKBONE", "mobilenet_v2",
                               "MODEL.DEEPLABV3_PLUS.USE_ASPP", "False",
                               "MODEL.DEEPLABV3_PLUS.ENABLE_DECODER", "False"],
               batch=1, h=1024, w=2048, train=False, aux=False, oracle="deeplab_mobilenet", os=16,
               metric="images/sec inference DeepLabv3_plus_mobilenetV2 @1024x2048",
               workload="DeepLabv3+_mobilenet_v2 inference (the reference's "
                        "cityscapes_deeplabv3_plus_mobilenet.yaml: USE_ASPP / ENABLE_DECODER False)",
               tag="BASELINE.json configs[1]"),
    # not a BASELINE config: the §8(f4) widening, README model-zoo row "Fast_SCNN ... 145.77 FPS"
    # (V100, 1024x2048); configs/cityscapes_fast_scnn.yaml values (SOLVER.AUX True: the reference
    # forward computes both aux heads in evaluation mode too)
    "c7": dict(over=_COMMON + ["MODEL.MODEL_NAME", "FastSCNN", "SOLVER.AUX", "True",
                               "MODEL.BN_MOMENTUM", "0.01"],
               batch=1, h=1024, w=2048, train=False, aux=True, oracle="fast_scnn", os=16,
               momentum=0.01, metric="images/sec inference Fast_SCNN @1024x2048",
               workload="Fast-SCNN inference (configs/cityscapes_fast_scnn.yaml, aux heads on)",
               tag="README model zoo: 145.77 FPS on V100"),
    "c5": dict(over=_COMMON, yaml="configs/cityscapes_hrnet_w18_small_v1.yaml", batch=16, h=1024,
               w=2048, train=False, aux=False, oracle="hrnet_seg", os=16, momentum=0.01,
               metric="images/sec inference HRNet_w18_small_v1 @1024x2048 batch 16",
               workload="HRNet_w18_small_v1 inference", tag="BASELINE.json configs[4]"),
}


class GemmTimer:
    """HIP-event pairs around every launch of the dominant kernel (on the launch stream)."""

    def __init__(self):
        from segmentron_amd import hip_ops
        self.K = hip_ops
        self.orig = hip_ops.conv_gemm
        self.events, self.flops, self.active = [], 0.0, False
        self.events_all, self.flops_all = [], 0.0
        hip_ops.conv_gemm = self._wrapped

    def _wrapped(self, x, w_packed, O, KH, KW, stride, pad, dil, pro=None, bias=None, out=None,
                 want_stats=False, scatter=None, ep=None, tconv_out_hw=None):
        args = (x, w_packed, O, KH, KW, stride, pad, dil, pro, bias, out, want_stats, scatter, ep,
                tconv_out_hw)
        if not self.active:
            return self.orig(*args)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        y, p = self.orig(*args)
        e1.record()
        n, hi, wi, c = x.shape
        ho = self.K.conv_out_size(hi, KH, stride, pad, dil)
        wo = self.K.conv_out_size(wi, KW, stride, pad, dil)
        fl = 2.0 * n * ho * wo * KH * KW * c * O
        self.flops_all += fl
        self.events_all.append((e0, e1))
        # the dispatch rule of seg_conv_gemm_fwd (csrc/conv_gemm_fwd.hip: gemm_use_px256 +
        # conv_gemm_glds_usable): the direct-to-LDS 256x256 kernel
        m = n * ho * wo
        if (KH * KW == 1 and stride == 1 and pad == 0
                and ((O >= 384 and m >= 4096) or (O >= 256 and m >= 65536))
                and scatter is None and tconv_out_hw is None and x.dtype == torch.bfloat16
                and (pro is None or pro[0] == 0) and bias is None and O % 8 == 0):
            self.flops += fl
            self.events.append((e0, e1))
        return y, p

    def result(self):
        ms = sum(a.elapsed_time(b) for a, b in self.events)
        return self.flops, ms * 1e-3, len(self.events)

    def result_all(self):
        ms = sum(a.elapsed_time(b) for a, b in self.events_all)
        return self.flops_all, ms * 1e-3, len(self.events_all)


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(size, conf="c3", keep=None):
    """`keep` (a dict, c3 train only): the oracle step's loss / logits sample / gradients of the
    LAST timed iteration are left there for the parity leg (oracle/parity.py) — the step runs on
    the well-conditioned synth state with dropout off in that case (same cost; the state only
    differs in BatchNorm affine parameters)."""
    from oracle import synth, torch_ref
    import segmentron_amd
    c = CONFIGS[conf]
    h, w = size
    batch = c["batch"] if c["train"] else 1  # inference: per-image rate of the CPU path
    threads = min(32, os.cpu_count() or 1)  # oneDNN collapses when oversubscribed (256-core host)
    torch.set_num_threads(threads)
    model = segmentron_amd.get_segmentation_model()
    parity = keep is not None and conf in ("c3", "c4") and c["train"]
    sd = synth.synth_like(model.state_dict(), seed=0, conditioned=parity)
    x = synth.synth_images(batch, h, w, seed=0)
    y = synth.synth_targets(batch, h, w, seed=0)
    times = []
    for it in range(3):  # 1 warm-up + 2 timed
        if parity:
            from oracle import parity as OP
            okw = {} if conf == "c3" else dict(output_stride=c["os"], aux=c["aux"], eps_encoder=None)
            res = OP.oracle_step(sd, x, y, torch.float32, c["oracle"], **okw)
            times.append(res["seconds"])
            if it == 2:
                keep.update(res, state=sd, x=x, y=y, conf=conf)
            del res
            continue
        osd = torch_ref.clone_state(sd, requires_grad=c["train"])
        kw = dict(eps_encoder=1e-3) if conf == "c3" else dict(
            output_stride=c["os"], aux=c["aux"], drop_p=0.0, momentum=c.get("momentum"))
        net = torch_ref.OracleNet(osd, training=c["train"], **kw)
        t0 = time.perf_counter()
        if c["train"]:
            loss = torch_ref.mix_softmax_ce(getattr(net, c["oracle"])(x), y)
            loss.backward()
        else:
            with torch.no_grad():
                loss = getattr(net, c["oracle"])(x)
        times.append(time.perf_counter() - t0)
        del osd, net, loss
    t = sum(times[1:]) / 2.0
    ratio = (h * w) / float(c["h"] * c["w"])
    full = (h, w) == (c["h"], c["w"])
    return {"value": batch * ratio / t, "unit": "images/sec" if full else
            "images/sec (%dx%d-equivalent)" % (c["h"], c["w"]), "cores": threads,
            "host_cores": os.cpu_count(),
            "cpu_model": _cpu_model(), "torch": torch.__version__, "kind": "port",
            "sample": "oracle (torch CPU fp32 restatement of the reference graph) %s, "
                      "batch %d @%dx%d, 1 warm-up + 2 timed (%.1f s, %.1f s)%s%s"
                      % ("train fwd+bwd" if c["train"] else "eval forward", batch, h, w, times[1],
                         times[2], "" if full else ", scaled by pixel ratio %.4f" % ratio,
                         ", conditioned synth state, dropout off (= the parity leg's step)"
                         if parity else "")}


def parity_leg(keep):
    """One EAGER full-size step of the HIP path in fp32 and one in bf16 on the state and input
    of the oracle step `cpu_baseline` just ran (the oracle is the checker here, never the thing
    measured): VERDICT r04 Missing #1, /root/reference/tools/train.py:135-146.  The bars live in
    tests/test_parity_conditioned.py::test_c3_train_full_size_1025x2049_matches_oracle; the
    yardstick of the bf16 figures is the oracle under CPU bf16 autocast at the same size
    (tests/golden/c3_autocast_sizes.json, oracle/gen_autocast_sizes.py)."""
    from oracle import parity as OP
    out = {"state": "oracle.synth conditioned, dropout off", "oracle": "CPU fp32",
           "logits_sample": "[::%d, ::%d] pixel grid" % (OP.SAMPLE, OP.SAMPLE)}
    for dt in ("fp32", "bf16"):
        try:
            got = OP.hip_step(dt, keep["state"], keep["x"], keep["y"],
                              eps_encoder=1e-3 if keep.get("conf", "c3") == "c3" else None)
            cmp = OP.compare(got, keep)
            del got
            out.update({"loss_rel_" + dt: cmp["loss_rel"], "logits_maxrel_" + dt: cmp["logits_maxrel"],
                        "logits_l2rel_" + dt: cmp["logits_l2rel"],
                        "argmax_agree_" + dt: cmp["argmax_agree"],
                        # pixels whose arg-max differs / of those, NOT oracle top-2 near-ties
                        "argmax_mismatch_" + dt: [cmp["argmax_mismatch"], cmp["argmax_pixels"]],
                        "argmax_unexplained_" + dt: cmp["argmax_unexplained"],
                        "grad_global_rel_" + dt: cmp["grad_global_rel"],
                        "grad_cosine_" + dt: cmp["grad_cosine"],
                        "grad_norm_ratio_" + dt: cmp["grad_norm_ratio"],
                        "grad_error_top_" + dt: cmp["grad_error_top"][:4]})
            if cmp["grad_tensors_missing"] or not cmp["finite"]:
                out["error_" + dt] = "missing %d gradient tensors, finite=%s" % (
                    cmp["grad_tensors_missing"], cmp["finite"])
        except Exception as e:  # noqa: BLE001 — report, never hide the bench line
            out["error_" + dt] = repr(e)[:300]
    try:  # the yardstick of the bf16 figures: the oracle under CPU bf16 autocast at THIS size
        if keep.get("conf", "c3") != "c3":
            raise KeyError("the autocast yardstick exists for C3 only")
        ac = json.load(open(os.path.join(ROOT, "tests", "golden", "c3_autocast_sizes.json")))
        ac = ac["%dx%d" % tuple(keep["x"].shape[2:])]
        out["oracle_cpu_autocast_bf16_same_size"] = {
            k: ac[k] for k in ("loss_rel", "logits_l2rel", "argmax_agree", "grad_cosine",
                               "grad_norm_ratio")}
    except Exception:  # noqa: BLE001
        pass
    out["pass_fp32_1e-3"] = bool(
        out.get("loss_rel_fp32", 1.0) < 1e-3 and out.get("logits_maxrel_fp32", 1.0) < 1e-3
        and out.get("grad_global_rel_fp32", 1.0) <= 1e-3 and out.get("argmax_unexplained_fp32", 1) == 0)
    return out


def self_launch(args):
    """`python bench.py --gpus N` outside torchrun: start N ranks the way the driver does."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
           "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def kernel_time_of_one_step(run_step):
    """Sum of device kernel durations of one step (ms) and the kernel count, from torch.profiler
    (roctracer); (None, None) if the profiler is unavailable."""
    try:
        from torch.profiler import ProfilerActivity, profile
        with profile(activities=[ProfilerActivity.CUDA]) as prof:
            run_step()
            torch.cuda.synchronize()
        tot_us, n = 0.0, 0
        per = {}
        for ev in prof.events():
            if getattr(ev, "device_type", None) is not None and "cuda" in str(ev.device_type).lower():
                dur = getattr(ev, "device_time", None)
                if dur is None:
                    dur = getattr(ev, "cuda_time", 0.0)
                name = ev.name.lower()
                if "memcpy" in name or "memset" in name:
                    continue
                tot_us += float(dur)
                n += 1
                key = ev.name.split("(")[0].replace("void ", "").replace("seg::", "")
                t = per.setdefault(key, [0.0, 0])
                t[0] += float(dur)
                t[1] += 1
        TOP_KERNELS[:] = [{"kernel": k[:80], "ms_per_step": round(v[0] * 1e-3, 4), "launches": v[1]}
                          for k, v in sorted(per.items(), key=lambda kv: -kv[1][0])[:6]]
        return (tot_us * 1e-3, n) if n else (None, None)
    except Exception as e:  # noqa: BLE001 — diagnostics only, never fail the bench for it
        sys.stderr.write("kernel_time_of_one_step: %r\n" % (e,))
        return None, None


TOP_KERNELS = []  # filled by kernel_time_of_one_step: the six largest kernels of one eager step


def _timed(fn, n, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def extra_legs(args, conf, model, opt, images, targets, step, loss_fn, dev, graph):
    """Untimed-by-the-contract extra figures of the same workload (VERDICT r03 #12c, next #4),
    single GPU, measured AFTER the timed region:
      eager_ms_per_step       the step as ~1100 eager launches (--no-graph)
      train_loop_ms_per_step  the reference's tools/train.py:135-146 statements, unchanged, with
                              SEGMENTRON_HIP_GRAPH=1 semantics (graph.TransparentTrainGraph: forward
                              and backward replay captured graphs, criterion + optimizer eager)
      fp32_ms_per_step        the exact-fp32 kernels (the path the 1e-3 parity claim is made on),
                              whole step in one HIP graph"""
    import segmentron_amd
    from segmentron_amd import functional as SF
    from segmentron_amd import graph as SG
    from segmentron_amd.solver.optimizer import FusedSGD
    out = {}
    try:
        out["eager_ms_per_step"] = _timed(step, 5)
    except Exception as e:  # noqa: BLE001 — diagnostics must not fail the bench line
        out["eager_ms_per_step_error"] = repr(e)[:200]
    try:
        tg = SG.TransparentTrainGraph.install(model, warmup=1)

        class Criterion(torch.nn.CrossEntropyLoss):  # shape of solver/loss.py:16-46
            def forward(self, preds, target):
                loss = super().forward(preds[0], target)
                for p in preds[1:]:
                    loss = loss + 0.4 * super().forward(p, target)
                return dict(loss=loss)
        criterion = Criterion(ignore_index=-1).to(dev)

        def loop_iteration():
            outputs = model(images)
            loss_dict = criterion(outputs, targets)
            losses = sum(loss for loss in loss_dict.values())
            opt.zero_grad()
            losses.backward()
            opt.step()
        out["train_loop_ms_per_step"] = _timed(loop_iteration, 20, warm=4)
        out["train_loop_launch"] = "hip_graph (forward + backward segments)" \
            if tg.segments and tg.disabled is None else "eager (%s)" % tg.disabled
        tg.uninstall()
        del tg
    except Exception as e:  # noqa: BLE001
        out["train_loop_ms_per_step_error"] = repr(e)[:200]
    if args.dtype == "bf16":
        try:
            torch.cuda.synchronize()
            SF.clear_weight_cache()
            segmentron_amd.set_compute_dtype("fp32")
            torch.manual_seed(0)
            m32 = segmentron_amd.get_segmentation_model().to(dev).train()
            o32 = FusedSGD([{"params": m32.parameters(), "lr": 0.02}], lr=0.02, momentum=0.9,
                           weight_decay=1e-4)
            g32 = SG.GraphedTrainStep(m32, o32, images, targets, loss_fn)
            out["fp32_ms_per_step"] = _timed(g32, 10, warm=3)
            del g32, o32, m32
        except Exception as e:  # noqa: BLE001
            out["fp32_ms_per_step_error"] = repr(e)[:200]
        finally:
            segmentron_amd.set_compute_dtype(args.dtype)
            SF.clear_weight_cache()
    return out


def _graph_busy():
    try:
        return json.load(open(os.path.join(ROOT, "profiles", "rocprof_roofline.json"))).get("graph_busy_frac")
    except Exception:  # noqa: BLE001
        return None


def rocprof_fraction(args, flops_per_step, peak_tflops):
    """`roofline.frac_rocprof`: the dominant kernel's algorithmic FLOPs per step (counted live,
    above) over its per-step time in the committed rocprofv3 --kernel-trace --stats summary of
    the same command (profiles/rocprof_roofline.json, written by tools/rocprof_roofline.py from
    the CSV named there)."""
    path = os.path.join(ROOT, "profiles", "rocprof_roofline.json")
    if args.config != "c3" or args.dtype != "bf16" or not os.path.exists(path) or flops_per_step <= 0:
        return None
    try:
        rj = json.load(open(path))
        ms = float(rj["glds_ms_per_step"])
        ach = flops_per_step / (ms * 1e-3) / 1e12
        return {"achieved_rocprof": ach, "frac_rocprof": ach / peak_tflops,
                "rocprof_kernel_ms_per_step": ms, "rocprof_source": rj.get("source")}
    except Exception as e:  # noqa: BLE001
        return {"frac_rocprof_error": repr(e)[:200]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="c3", choices=sorted(CONFIGS),
                    help="c3 = BASELINE.json's metric (default); c2 / c4 / c5 = its other configs; "
                         "c7 = Fast-SCNN inference (README model zoo)")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--no-graph", action="store_true", help="eager launches (no HIP graph)")
    ap.add_argument("--cpu-baseline-size", default=None)
    ap.add_argument("--no-extra-legs", action="store_true",
                    help="skip the untimed extra legs (eager / fp32 / reference-loop ms_per_step)")
    ap.add_argument("--no-parity", action="store_true",
                    help="skip the full-size parity leg (one fp32 + one bf16 eager step of the HIP "
                         "path against the oracle step the cpu_baseline leg runs)")
    ap.add_argument("--prewarm-seconds", type=float, default=10.0,
                    help="continuous untimed replay before the warm-up + timed steps")
    args = ap.parse_args()
    conf = CONFIGS[args.config]
    batch, train = conf["batch"], conf["train"]
    if args.height is None:
        args.height = conf["h"]
    if args.width is None:
        args.width = conf["w"]
    if args.cpu_baseline_size is None:
        # (c4's CPU train step at 1025x2049 takes ~40 s on 32 cores: 1 warm-up + 2 timed = 2 minutes,
        # paid since r06 because the same oracle step is the reference of the `parity` object;
        # `--cpu-baseline-size 513x1025` is the bounded sample of r02 - r05)
        args.cpu_baseline_size = "%dx%d" % (conf["h"], conf["w"])

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks"
                 % (args.gpus, world))
    # SEG_BENCH_ONE_DEVICE=1 (test plumbing): every rank on device 0 over gloo, to exercise the
    # N > 1 code path on a single-GPU box (RCCL refuses two ranks per device)
    one_dev = os.environ.get("SEG_BENCH_ONE_DEVICE") == "1"
    if one_dev:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    # SEG_BENCH_FORCE_DDP=1 (test plumbing): the N > 1 machinery — RCCL process group,
    # convert_sync_batchnorm with the statistics all-reduces forced on, gradient averaging — with
    # ONE rank, so that a single-GPU box exercises the RCCL calls (also inside the HIP graph)
    force_ddp = os.environ.get("SEG_BENCH_FORCE_DDP") == "1" and world == 1 and conf["train"]
    multi = world > 1 or force_ddp
    # N > 1 data-parallel mode:
    #   native (default) — SyncBN statistics and gradient averaging as direct RCCL calls
    #                      (segmentron_amd/rccl.py) on the compute stream, the WHOLE step incl. the
    #                      collectives captured into one HIP graph;
    #   ddp              — torch DistributedDataParallel + torch.distributed all-reduces, eager
    #                      launches (ProcessGroupNCCL cannot be captured on this stack)
    dp_mode = os.environ.get("SEG_BENCH_DP", "native") if (multi and conf["train"]) else None
    if one_dev and dp_mode == "native":
        dp_mode = "ddp"  # gloo ranks on one device: no RCCL
    comm = None
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        # RCCL prints its version banner on STDOUT at communicator creation: the one JSON line
        # of this program must stay the only thing there
        sys.stdout.flush()
        saved_out = os.dup(1)
        os.dup2(2, 1)
        try:
            if one_dev:
                dist.init_process_group("gloo", init_method="env://")
            else:
                dist.init_process_group("nccl", init_method="env://", device_id=dev)
                warm = torch.zeros(1, device=dev)
                dist.all_reduce(warm)  # creates torch's communicator (banner) now
                torch.cuda.synchronize()
            if dp_mode == "native":
                try:
                    from segmentron_amd import parallel as SP
                    from segmentron_amd import rccl as SR
                    comm = SR.communicator_from_torch_group()
                    if os.environ.get("SEG_BENCH_XGMI", "1") == "1":
                        # SyncBatchNorm statistics as one-hop xGMI peer writes, checked against
                        # this communicator first (falls back to it with a line on stderr)
                        from segmentron_amd import xgmi as SX
                        comm = SX.connect(comm, rank, world)
                    SP.use_native_rccl(comm)
                except Exception as e:  # noqa: BLE001 — every rank takes the same branch
                    sys.stderr.write("bench.py: native RCCL communicator unavailable (%r): "
                                     "torch DDP, eager\n" % (e,))
                    dp_mode, comm = "ddp", None
        finally:
            sys.stdout.flush()
            ctypes.CDLL(None).fflush(None)  # the banner sits in libc's stdout buffer
            os.dup2(saved_out, 1)
            os.close(saved_out)
        if force_ddp:
            os.environ["SEG_SYNC_FORCE"] = "1"  # parallel.sync_group: all-reduce with one rank

    import segmentron_amd
    from segmentron_amd.config import cfg, reset_cfg
    reset_cfg()
    if conf.get("yaml"):
        cfg.update_from_file(os.path.join(ROOT, conf["yaml"]))
    cfg.update_from_list(conf["over"])
    cfg.PHASE = "train" if train else "test"
    cfg.check_and_freeze()
    segmentron_amd.set_compute_dtype(args.dtype)
    torch.manual_seed(0)
    model = segmentron_amd.get_segmentation_model()
    if cfg.MODEL.BN_EPS_FOR_ENCODER and getattr(model, "encoder", None) is not None:
        for _, m in model.encoder.named_modules():  # solver/optimizer.py:18-20
            if isinstance(m, torch.nn.BatchNorm2d):
                m.eps = cfg.MODEL.BN_EPS_FOR_ENCODER
    model = model.to(dev).train(train)
    opt = None
    if train:
        if multi:
            model = torch.nn.SyncBatchNorm.convert_sync_batchnorm(model)  # tools/train.py:76
            if dp_mode == "ddp":
                model = torch.nn.parallel.DistributedDataParallel(model, device_ids=[local],
                                                                  output_device=local)
            elif world > 1:  # what DDP's constructor does: everybody starts from rank 0's state
                for t in model.state_dict().values():
                    dist.broadcast(t, src=0)
        params = [{"params": model.parameters(), "lr": cfg.SOLVER.LR}]
        sgd = dict(lr=cfg.SOLVER.LR, momentum=cfg.SOLVER.MOMENTUM,
                   weight_decay=cfg.SOLVER.WEIGHT_DECAY)  # solver/optimizer.py:45-66
        # the reference's optimizer (solver/optimizer.py:45-50) on the multi-tensor HIP kernel
        from segmentron_amd.solver.optimizer import FusedSGD
        opt = FusedSGD(params, **sgd)

    g = torch.Generator().manual_seed(rank)
    images = torch.randn(batch, 3, args.height, args.width, generator=g).to(dev)
    targets = torch.randint(0, 19, (batch, args.height, args.width), generator=g)
    targets[torch.rand(batch, args.height, args.width, generator=g) < 0.05] = -1
    targets = targets.to(dev)
    ce = torch.nn.functional.cross_entropy

    def loss_fn(out, tgt):  # solver/loss.py:16-46 MixSoftmaxCrossEntropyLoss (aux weight 0.4)
        loss = ce(out[0], tgt, ignore_index=-1)
        for o in out[1:]:
            loss = loss + cfg.SOLVER.AUX_WEIGHT * ce(o, tgt, ignore_index=-1)
        return loss

    def step():
        if not train:
            with torch.no_grad():
                return model(images)[0]
        loss = loss_fn(model(images), targets)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        if post_backward is not None:
            post_backward()
        opt.step()
        return loss

    post_backward = None
    grad_overlap = None
    if train and dp_mode == "native":
        from segmentron_amd import parallel as SP
        dp_params = list(model.parameters())
        post_backward = lambda: SP.average_gradients(dp_params)  # noqa: E731 — DDP's mean
        if os.environ.get("SEG_BENCH_GRAD_OVERLAP", "1") == "1":
            # DDP's other half: bucketed all-reduces overlapped with backward on a side stream,
            # over a communicator of their own (the SyncBN exchanges keep the first one busy)
            try:
                from segmentron_amd import rccl as SR
                sys.stdout.flush()
                saved_out = os.dup(1)
                os.dup2(2, 1)
                try:
                    comm_grad = SR.communicator_from_torch_group()
                finally:
                    ctypes.CDLL(None).fflush(None)
                    os.dup2(saved_out, 1)
                    os.close(saved_out)
                grad_overlap = SP.OverlappedGradientAverager(dp_params, comm_grad)
                post_backward = grad_overlap.finish
            except Exception as e:  # noqa: BLE001
                sys.stderr.write("bench.py: gradient overlap unavailable (%r): one grouped "
                                 "all-reduce after backward\n" % (e,))
                grad_overlap = None

    timer = GemmTimer()
    # ---- launch path: one HIP graph of the whole step (single GPU), else eager
    graph, graph_err = None, None
    use_graph = (not multi or not train or dp_mode == "native") and not args.no_graph \
        and os.environ.get("SEG_BENCH_GRAPH", "1") != "0"
    from segmentron_amd import functional as SF
    from segmentron_amd import graph as SG
    if use_graph:
        try:  # segmentron_amd/graph.py: eager warm-up on a side stream, then ONE capture
            if train:
                graph = SG.GraphedTrainStep(model, opt, images, targets, loss_fn,
                                            post_backward=post_backward,
                                            check=getattr(comm, "check", None))
            else:
                graph = SG.GraphedInference(model, images)
        except Exception as e:  # noqa: BLE001
            graph, graph_err = None, repr(e)[:300]
            sys.stderr.write("bench.py: HIP-graph capture failed, running eager: %s\n" % graph_err)
            if os.environ.get("SEG_BENCH_DEBUG") == "1":
                import traceback
                traceback.print_exc()
            torch.cuda.synchronize()
            SF.clear_weight_cache()
    if graph is None:
        # eager path (N > 1, --no-graph): warm up on the CURRENT stream — a side-stream warm-up
        # (the graph-capture recipe) leaves the 440 AccumulateGrad nodes bound to that stream and
        # every later step pays a cross-stream event pair per parameter (+13 ms/step of host time)
        for _ in range(3):
            step()
        torch.cuda.synchronize()

    def run_step():
        if graph is not None:
            out = graph()
            return out if train else out[0]
        return step()

    # Device pre-conditioning (untimed, reported as "prewarm_steps"): a fresh box occasionally ran
    # its first 0.5 s of steps ~25 % slow (clock / power-state ramp, lazy kernel-module loads,
    # allocator growth) — a whole default-length bench fits into that window.  Run 20 extra
    # steps first, THEN the W warm-up steps and the K timed steps of the contract.
    # r04: >= 10 s of CONTINUOUS replay (VERDICT r03 #12a): clocks / power state are settled and
    # an outside observer (the driver's SMI sampler, 5 s period) sees the GPU busy.  The count is
    # derived from 5 probe steps and agreed over the ranks (every rank issues the same
    # collectives).
    torch.cuda.synchronize()
    t_probe = time.perf_counter()
    for _ in range(5):
        run_step()
    torch.cuda.synchronize()
    per = max((time.perf_counter() - t_probe) / 5.0, 1e-4)
    prewarm = max(20, int(args.prewarm_seconds / per) + 1)
    if world > 1:
        tn = torch.tensor([prewarm], device=dev, dtype=torch.int64)
        dist.all_reduce(tn, op=dist.ReduceOp.MAX)
        prewarm = int(tn.item())
    for _ in range(prewarm):
        run_step()
    torch.cuda.synchronize()
    prewarm += 5
    for _ in range(args.warmup):
        run_step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    step_ends = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = run_step()
        step_ends.append(time.perf_counter())
    torch.cuda.synchronize()
    if os.environ.get("SEG_BENCH_HOST_PROFILE") == "1" and rank == 0:
        sys.stderr.write("host-side issue time per step (ms): %s; drain %.1f\n" % (
            " ".join("%.1f" % ((b - a) * 1e3) for a, b in zip([t0] + step_ends, step_ends)),
            (time.perf_counter() - step_ends[-1]) * 1e3))
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()
    loss_value = float(loss.item()) if train else float(loss.float().abs().mean().item())
    if hasattr(comm, "check"):
        comm.check()  # a peer that stopped publishing its statistics invalidates the run: raise
    kernel_ms, n_kernels = (None, None)
    # per-launch HIP events of the dominant kernel: three EAGER steps (same kernels, same shapes)
    # AFTER the timed region — event pairs around every launch slow the launch-bound eager path
    # itself down (2 x 157 timed events per step), so they must not sit in the timed steps of an
    # eager (N > 1) run either
    roofline_steps = 3
    if graph is not None:
        if train:
            graph.release()
        step()
    torch.cuda.synchronize()
    timer.active = True
    for _ in range(roofline_steps):
        # An eager step is launch-bound (the host needs longer to issue the ~1200 launches
        # than the GPU to run them), and an event pair around a launch then also measures
        # the queue running dry.  Give the host a head start: the GPU spins ~60 ms first, so
        # every kernel of the step is already queued when its turn comes and e0 -> e1 is the
        # kernel's duration alone.
        torch.cuda.synchronize()
        try:
            torch.cuda._sleep(int(1.5e8))
        except Exception:  # noqa: BLE001 — no spin kernel in this build: measure as is
            pass
        step()
    torch.cuda.synchronize()
    timer.active = False
    if os.environ.get("SEG_BENCH_HOST_PROFILE") == "1":
        # where the host time of an eager step goes (stderr; diagnostics for the N > 1 path).
        # EVERY rank runs the extra steps (they issue collectives); rank 0 prints.
        import cProfile
        import pstats
        pr = cProfile.Profile()
        pr.enable()
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        pr.disable()
        if rank == 0:
            pstats.Stats(pr, stream=sys.stderr).sort_stats("tottime").print_stats(25)
    # one EAGER step under the profiler (roctracer does not see the nodes of a replayed graph):
    # sum of its kernel durations / the timed ms_per_step -> gpu_busy_frac
    if rank == 0 and world == 1:
        kernel_ms, n_kernels = kernel_time_of_one_step(step)

    extra = {}
    if rank == 0 and world == 1 and train and not args.no_extra_legs:
        extra = extra_legs(args, conf, model, opt, images, targets, step, loss_fn, dev, graph)

    if rank == 0:
        ms = elapsed / args.steps * 1e3
        value = world * batch * args.steps / elapsed
        flops, secs, launches = timer.result()
        achieved = flops / secs / 1e12 if secs > 0 else 0.0
        traffic = traffic_src = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath) and args.config == "c3":
            tj = json.load(open(tpath))
            traffic = tj.get("conv_gemm_glds_bytes_per_launch")
            traffic_src = tj.get("source")
        fa, sa, la = timer.result_all()
        # the roofline kernel: C3's dominant kernel is the direct-to-LDS 1x1 GEMM; the other
        # configs are led by other members of the conv GEMM family (HRNet / MobileNetV2 barely run
        # the 256-wide kernel at all — r04's c5 line said frac 0.0 of a kernel it never launched):
        # there the figure is taken over ALL forward / data-gradient conv GEMM launches, and
        # `top_kernels` names what the step actually spends its time in
        roof_kernel = "conv_gemm_glds_kernel + conv_gemm_glds4_kernel (bf16, direct-to-LDS 1x1 GEMM)" if args.dtype == "bf16" \
            else "conv_gemm_px256_kernel<fp32>"
        if args.config != "c3" or launches == 0:
            roof_kernel = "conv_gemm_* (all forward / data-gradient convolution GEMM launches)"
            flops, secs, launches = fa, sa, la
            achieved = flops / secs / 1e12 if secs > 0 else 0.0
        full = (args.height, args.width) == (conf["h"], conf["w"])
        peak = MFMA_BF16_PEAK / 1e12 if args.dtype == "bf16" else 157.3
        line = {
            "metric": conf["metric"],
            "value": value, "unit": "images/sec", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "prewarm_steps": prewarm, "ms_per_step": ms,
            "higher_is_better": True, "scaling": "weak",
            "launch": "hip_graph" if graph is not None else "eager",
            # (an EAGER step under torch.profiler: kernel count and the sum of their durations —
            # not comparable with the replayed step's wall time, so no ratio is formed from it)
            "eager_profile_kernel_ms_per_step": kernel_ms, "kernels_per_step": n_kernels,
            # busy fraction of the REPLAYED step: kernel durations / wall span of one rocprofv3
            # kernel trace of this command (tools/rocprof_roofline.py), when profiles/ holds one
            "gpu_busy_frac": _graph_busy(),
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "%s @%dx%d, batch %d/GPU (%s)"
                                   % (conf["workload"], args.height, args.width, batch, conf["tag"]),
                       "global_batch": world * batch,
                       "bn": ("SyncBN" if (world > 1 or force_ddp) else "BN") if train
                       else "eval (running stats)",
                       "parallelism": ("dp%d" % world) if train else ("replicas x%d" % world),
                       "dp_mode": dp_mode,
                       "syncbn_exchange": (None if dp_mode != "native" else
                                           "xgmi peer mailbox (csrc/p2p.hip)"
                                           if hasattr(comm, "mailbox") else "rccl all-reduce"),
                       "grad_allreduce": (None if dp_mode != "native" else
                                          ("bucketed, overlapped with backward (side stream, "
                                           "%d buckets)" % len(grad_overlap.buckets))
                                          if grad_overlap is not None else "grouped, after backward"),
                       "full_size": full, "loss" if train else "mean_abs_logit": loss_value},
            "model_flop_fraction_of_bf16_mfma_peak":
                value / world * FLOP_FWD_BWD_PER_IMAGE / MFMA_BF16_PEAK
                if (full and args.config == "c3") else None,
            "roofline": {"bound": "mfma", "kernel": roof_kernel,
                         "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                         "frac": achieved / peak,
                         "traffic": traffic, "traffic_source": traffic_src,
                         "launches_per_step": launches / max(roofline_steps, 1),
                         "kernel_ms_per_step": secs * 1e3 / max(roofline_steps, 1),
                         "timing": "HIP events around each launch, %d eager steps with the launch queue kept "
                                   "full (GPU spin-wait head start) after the timed %s"
                                   % (roofline_steps, "graph replays" if graph is not None
                                      else "eager steps"),
                         "all_gemm_achieved": fa / sa / 1e12 if sa > 0 else 0.0,
                         "all_gemm_launches_per_step": la / max(roofline_steps, 1),
                         "all_gemm_ms_per_step": sa * 1e3 / max(roofline_steps, 1)},
        }
        if TOP_KERNELS:
            line["roofline"]["top_kernels"] = list(TOP_KERNELS)
        line.update(extra)
        rp = rocprof_fraction(args, flops / max(roofline_steps, 1), peak)
        if rp:
            line["roofline"].update(rp)
        if graph_err:
            line["hip_graph_error"] = graph_err
        if not args.no_cpu_baseline and world == 1:
            ch, cw = (int(v) for v in args.cpu_baseline_size.lower().split("x"))
            keep = {} if (args.config in ("c3", "c4") and train and (ch, cw) == (args.height, args.width)
                          and not args.no_parity) else None
            # (the timed model / graph / optimizer are dropped first: the parity leg builds two
            # fresh full-size models on the same device)
            line["cpu_baseline"] = cpu_baseline((ch, cw), args.config, keep)
            if keep:
                del graph, model, opt
                SF.clear_weight_cache()
                torch.cuda.empty_cache()
                line["parity"] = parity_leg(keep)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

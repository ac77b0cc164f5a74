"""`SegmentationMetric` of segmentron/utils/score.py:10-80 (pixAcc / mIoU; the class
tools/train.py:114 and tools/eval.py:60 construct) on ONE HIP kernel per update.

The reference takes two arg-max maps, builds three float maps, moves them to the CPU for
`torch.histc` and moves the histograms back (score.py:83-113), then — distributed — all-reduces
four tensors per batch.  Here an update is a single pass over the logits that bumps exact int64
counters on the device (csrc/metric.hip); a `LogitsView` (functional.py: the network's
low-resolution logits with the final bilinear resize pending) is consumed without materialising
the [N, C, H, W] tensor.  The distributed reduction happens once per `get()` on the 2 + 3*nclass
counters.  Per-update counts equal the reference's exactly (including its arg-max of the
integer-TRUNCATED logits for pixAcc, score.py:86); totals are exact integers where the reference
accumulates float32.
"""
import torch
from torch import distributed as dist

from .. import hip_ops as K

__all__ = ["SegmentationMetric"]

_EPS = 2.220446049250313e-16  # np.spacing(1), score.py:68-69


class SegmentationMetric(object):
    def __init__(self, nclass, distributed):
        self.nclass = nclass
        self.distributed = distributed
        self._cnt = None
        self.reset()

    def reset(self):
        self._cnt = None
        self._cache = None

    def _counters(self, device):
        if self._cnt is None:
            self._cnt = K.metric_counters(self.nclass, device)
        return self._cnt

    def update(self, preds, labels):
        from ..functional import LogitsView
        self._cache = None
        if isinstance(preds, (list, tuple)):
            for p, l in zip(preds, labels):
                self.update(p, l)
            return
        if isinstance(preds, LogitsView) and preds._full is None and labels.is_cuda \
                and tuple(labels.shape) == (preds.lo.shape[0],) + tuple(preds.out_hw):
            lo = preds.lo  # the resize is still pending: take the counts through it
            K.metric_update_upsample(lo, labels, preds.align_corners, self.nclass,
                                     self._counters(lo.device))
            return
        if isinstance(preds, LogitsView):
            preds = preds.materialize()
        if not isinstance(preds, torch.Tensor) or not preds.is_cuda:
            raise RuntimeError("SegmentationMetric (segmentron_amd) takes HIP device tensors: "
                               "there is no CPU fallback")
        K.metric_update_nchw(preds, labels.to(preds.device), self.nclass,
                             self._counters(preds.device))

    def _totals(self):
        """(correct, labelled, inter[nclass], union[nclass]) summed over ranks, on the host.

        Distributed: the FIRST read after an update is a collective (one all_reduce of the
        2 + 3*nclass counters) and must therefore happen on every rank — tools/train.py:196 and
        tools/eval.py:73 call `get()` on all ranks.  The reduced totals are cached until the next
        `update()` / `reset()`, so later rank-local reads (`total_*`, a rank-0-only `get()` for
        logging) issue no further collective and cannot deadlock."""
        if self._cache is not None:
            return self._cache
        n = self.nclass
        if self._cnt is None:
            z = torch.zeros(n, dtype=torch.float64)
            return 0, 0, z, z.clone()
        tot = self._cnt.clone()
        if self.distributed:
            dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        tot = tot.cpu()
        inter, pred, lab = tot[2:2 + n], tot[2 + n:2 + 2 * n], tot[2 + 2 * n:]
        self._cache = (int(tot[0]), int(tot[1]), inter.double(), (pred + lab - inter).double())
        return self._cache

    # the reference's public attributes (score.py:76-80)
    @property
    def total_correct(self):
        return self._totals()[0]

    @property
    def total_label(self):
        return self._totals()[1]

    @property
    def total_inter(self):
        return self._totals()[2].float()

    @property
    def total_union(self):
        return self._totals()[3].float()

    def get(self, return_category_iou=False):
        correct, labelled, inter, union = self._totals()
        assert correct <= labelled, "Correct area should be smaller than Labeled"
        assert bool((inter <= union).all()), "Intersection area should be smaller than Union area"
        pixAcc = 1.0 * correct / (_EPS + labelled)
        IoU = (1.0 * inter / (_EPS + union)).float()
        mIoU = IoU.mean().item()
        if return_category_iou:
            return pixAcc, mIoU, IoU.cpu().numpy()
        return pixAcc, mIoU

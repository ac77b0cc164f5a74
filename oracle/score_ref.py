"""TEST INFRASTRUCTURE (not product code): CPU restatement of the reference's per-batch metric
counts, segmentron/utils/score.py:83-113 (batch_pix_accuracy / batch_intersection_union), used
by tests/test_metric_gpu.py as the oracle of csrc/metric.hip.  `pin()` checks it against the
reference's own functions when /root/reference is importable (build container)."""
import torch


def batch_pix_accuracy(output, target):
    # score.py:83-92 — note the arg-max of the logits cast to int64 (truncation)
    predict = torch.argmax(output.long(), 1) + 1
    target = target.long() + 1
    labeled = torch.sum(target > 0)
    correct = torch.sum((predict == target) * (target > 0))
    return int(correct), int(labeled)


def batch_intersection_union(output, target, nclass):
    # score.py:95-113 — histc over [1, nclass] with nclass bins; zeros fall outside the range
    predict = torch.argmax(output, 1) + 1
    target = target.float() + 1
    predict = predict.float() * (target > 0).float()
    inter = predict * (predict == target).float()
    area_inter = torch.histc(inter.cpu(), bins=nclass, min=1, max=nclass)
    area_pred = torch.histc(predict.cpu(), bins=nclass, min=1, max=nclass)
    area_lab = torch.histc(target.cpu(), bins=nclass, min=1, max=nclass)
    return area_inter, area_pred, area_lab


def counters(output, target, nclass):
    """int64 [2 + 3*nclass] in csrc/metric.hip's layout."""
    c, l = batch_pix_accuracy(output, target)
    i, p, t = batch_intersection_union(output, target, nclass)
    return torch.cat([torch.tensor([c, l], dtype=torch.int64), i.long(), p.long(), t.long()])


def pin(ref_root="/root/reference"):
    """The restatement against the reference's own functions on a seeded batch."""
    import os
    import sys
    sys.path.insert(0, ref_root)
    sys.path.append(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                 "segmentron_amd", "shims"))
    from segmentron.utils import score as ref
    g = torch.Generator().manual_seed(0)
    out = torch.randn(2, 7, 33, 41, generator=g) * 3
    tgt = torch.randint(-1, 7, (2, 33, 41), generator=g)
    c, l = ref.batch_pix_accuracy(out, tgt)
    i, u = ref.batch_intersection_union(out, tgt, 7)
    mine = counters(out, tgt, 7)
    assert (int(c), int(l)) == (int(mine[0]), int(mine[1]))
    assert torch.equal(i.long(), mine[2:9])
    assert torch.equal(u.long(), mine[9:16] + mine[16:23] - mine[2:9])
    return True


if __name__ == "__main__":
    print("pinned against the reference:", pin())

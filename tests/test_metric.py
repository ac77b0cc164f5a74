"""SURVEY.md §8 f1 (metric tail): pixAcc / mIoU counters (csrc/metric.hip, utils/score.py).
CPU: the oracle restatement against counts produced by the reference's own functions
(tests/golden/score_counts.npz, oracle/gen_golden_score.py).  GPU: the kernels — fp32 NCHW and
the fused upsample variant — against the oracle, bit-exact (integer counts)."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import score_ref  # noqa: E402
from oracle.gen_golden_score import CASES, batch  # noqa: E402

GOLDEN = np.load(os.path.join(ROOT, "tests", "golden", "score_counts.npz"))


@pytest.mark.parametrize("i", range(len(CASES)))
def test_oracle_counts_match_reference_fixture(i):
    seed, n, c, h, w, q = CASES[i]
    assert list(GOLDEN["case%d" % i]) == [seed, n, c, h, w, int(q)]
    out, tgt = batch(seed, n, c, h, w, q)
    cnt = score_ref.counters(out, tgt, c)
    assert [int(cnt[0]), int(cnt[1])] == list(GOLDEN["pix%d" % i])
    inter, pred, lab = cnt[2:2 + c], cnt[2 + c:2 + 2 * c], cnt[2 + 2 * c:]
    assert np.array_equal(inter.numpy(), GOLDEN["inter%d" % i])
    assert np.array_equal((pred + lab - inter).numpy(), GOLDEN["union%d" % i])


def test_metric_refuses_cpu_tensors():
    from segmentron_amd.utils.score import SegmentationMetric
    m = SegmentationMetric(19, False)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m.update(torch.zeros(1, 19, 4, 4), torch.zeros(1, 4, 4, dtype=torch.int64))
    assert m.get() == (0.0, 0.0)


@pytest.mark.gpu
@pytest.mark.parametrize("i", range(len(CASES)))
def test_metric_kernel_counts_equal_oracle(i):
    from segmentron_amd import hip_ops as K
    seed, n, c, h, w, q = CASES[i]
    out, tgt = batch(seed, n, c, h, w, q)
    cnt = K.metric_counters(c, torch.device("cuda"))
    K.metric_update_nchw(out.cuda(), tgt.cuda(), c, cnt)
    want = score_ref.counters(out, tgt, c)
    assert torch.equal(cnt.cpu(), want)
    K.metric_update_nchw(out.cuda(), tgt.cuda(), c, cnt)  # counters accumulate
    assert torch.equal(cnt.cpu(), 2 * want)


@pytest.mark.gpu
def test_segmentation_metric_matches_reference_formulas_over_batches():
    from segmentron_amd.utils.score import SegmentationMetric
    m = SegmentationMetric(19, False)
    tot = torch.zeros(2 + 3 * 19, dtype=torch.int64)
    for seed in (5, 6, 7):
        out, tgt = batch(seed, 2, 19, 45, 77, seed == 6)
        m.update(out.cuda(), tgt.cuda())
        tot += score_ref.counters(out, tgt, 19)
    pix, miou, iou = m.get(return_category_iou=True)
    inter = tot[2:21].double()
    union = (tot[21:40] + tot[40:59] - tot[2:21]).double()
    eps = 2.220446049250313e-16
    assert pix == 1.0 * int(tot[0]) / (eps + int(tot[1]))
    want = (inter / (eps + union)).float()
    assert np.array_equal(iou, want.numpy()) and miou == want.mean().item()
    assert m.total_correct == int(tot[0]) and m.total_label == int(tot[1])
    m.reset()
    assert m.get() == (0.0, 0.0)
    m.update([out.cuda()], [tgt.cuda()])  # list form (score.py:54-56)
    assert m.total_label == int(score_ref.counters(out, tgt, 19)[1])


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["fp32", "bf16"])
@pytest.mark.parametrize("geom", [(2, 17, 33, 65, 129, 19, True), (1, 33, 65, 129, 257, 21, True),
                                  (1, 16, 32, 64, 128, 19, False)])
def test_fused_upsample_metric_equals_materialised_path(geom, dtype):
    """The counts through the pending resize are those of the materialised logits tensor, and a
    LogitsView goes down the fused path."""
    from segmentron_amd import functional as F, hip_ops as K
    from segmentron_amd.utils.score import SegmentationMetric
    N, Hi, Wi, H, W, C, align = geom
    g = torch.Generator().manual_seed(11)
    vec = 8 if dtype == torch.bfloat16 else 4
    pitch = (C + vec - 1) // vec * vec
    lo = torch.zeros(N, Hi, Wi, pitch, dtype=dtype, device="cuda")
    lo[..., :C] = (torch.randn(N, Hi, Wi, C, generator=g) * 3).to(dtype).cuda()
    lo = lo[..., :C]
    tgt = torch.randint(-1, C, (N, H, W), generator=g).cuda()
    full = K.upsample_to_nchw(lo, C, (H, W), align)
    a = K.metric_update_nchw(full, tgt, C, K.metric_counters(C, lo.device))
    b = K.metric_update_upsample(lo, tgt, align, C, K.metric_counters(C, lo.device))
    assert torch.equal(a, b)
    assert torch.equal(a.cpu(), score_ref.counters(full.cpu(), tgt.cpu(), C))
    view = F.LogitsView(lo, (H, W), align)
    m = SegmentationMetric(C, False)
    m.update(view, tgt)
    assert view._full is None  # never materialised
    assert torch.equal(m._cnt, a)

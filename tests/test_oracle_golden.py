"""CPU: the oracle restatement reproduces the fixtures generated from the reference itself
(oracle/gen_golden.py).  These fixtures are the pin for every parity claim."""
import json
import os

import numpy as np
import torch

from oracle import synth, torch_ref


def _load_state(golden_dir):
    keys = json.load(open(os.path.join(golden_dir, "c3_state_keys.json")))["keys"]
    sd = synth.synth_state_dict([(k, tuple(s)) for k, s in keys], seed=0)
    calib = np.load(os.path.join(golden_dir, "c3_bn_calib.npz"))
    for k in calib.files:
        sd[k] = torch.from_numpy(calib[k])
    for k in sd:
        if k.endswith("num_batches_tracked"):
            sd[k] = torch.zeros((), dtype=torch.long)
    return sd


def test_eval_logits_match_reference_fixture(golden_dir):
    sd = _load_state(golden_dir)
    g = np.load(os.path.join(golden_dir, "c3_eval_65x129.npz"))
    x = synth.synth_images(2, 65, 129, seed=0)
    net = torch_ref.OracleNet(torch_ref.clone_state(sd), training=False, eps_encoder=1e-3)
    with torch.no_grad():
        out = net.deeplabv3_plus_xception65(x)[0]
    ref = torch.from_numpy(g["logits"])
    # same torch build -> bit-identical; allow fp32 roundoff for a different CPU / thread count
    assert torch.allclose(out, ref, rtol=1e-4, atol=1e-4)
    assert (out.argmax(1).numpy() == g["argmax"]).mean() > 0.9999


def test_train_step_matches_reference_fixture(golden_dir):
    sd = _load_state(golden_dir)
    g = np.load(os.path.join(golden_dir, "c3_train_65x129.npz"))
    x = synth.synth_images(2, 65, 129, seed=0)
    y = synth.synth_targets(2, 65, 129, seed=0)
    osd = torch_ref.clone_state(sd, requires_grad=True)
    net = torch_ref.OracleNet(osd, training=True, eps_encoder=1e-3, drop_p=0.0)
    out = net.deeplabv3_plus_xception65(x)
    loss = torch_ref.mix_softmax_ce(out, y)
    loss.backward()
    assert abs(loss.item() - float(g["loss"])) < 1e-5
    assert torch.allclose(out[0].detach(), torch.from_numpy(g["logits"]), rtol=1e-4, atol=1e-4)
    names = [str(k) for k in g["grad_norm_keys"]]
    norms = g["grad_norms"]
    for k, n in zip(names, norms):
        got = float(osd[k].grad.double().norm())
        assert abs(got - n) <= 1e-3 * max(n, 1e-6) + 1e-9, k
    for k in g.files:
        if k.startswith("grad::"):
            ref = torch.from_numpy(g[k])
            got = osd[k[6:]].grad
            assert torch.allclose(got, ref, rtol=1e-3, atol=1e-6 + 1e-4 * ref.abs().max().item()), k
        if k.startswith("rv::"):
            assert torch.allclose(osd[k[4:] + ".running_var"], torch.from_numpy(g[k]), rtol=1e-5,
                                  atol=1e-7), k


def test_synth_is_deterministic():
    a = synth.synth_tensor("encoder.block4.sep_conv1.block.pointwise.weight", (728, 728, 1, 1))
    b = synth.synth_tensor("encoder.block4.sep_conv1.block.pointwise.weight", (728, 728, 1, 1))
    assert torch.equal(a, b)
    assert abs(a.std().item() - (2.0 / 728) ** 0.5) < 2e-3

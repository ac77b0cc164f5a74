"""Yardstick of the bf16 path at the LARGE sizes (TEST INFRASTRUCTURE; generated on the CPU of the
build container, committed as tests/golden/c3_autocast_sizes.json):

    python oracle/gen_autocast_sizes.py 513x1025 12
    python oracle/gen_autocast_sizes.py 1025x2049 0

One C3 train step (conditioned synth state, dropout off, batch 2) of the oracle — bit-identical
to the reference graph in fp32 (oracle/gen_golden*.py) — under torch's CPU bf16 autocast, against
the same step in fp32: how far does the REFERENCE's own mixed-precision path sit from fp32 at
this size?  tests/golden/c3_cond.npz holds the same figures at 65x129 (taken with the imported
reference itself); the arg-max agreement and the logits distance do not carry over to 1025x2049
(random-init logits are near-ties; the measured HIP bf16 path: 0.88 / 0.10 at full size, 0.97 /
0.04 at 65x129), so the full-size test takes its bars from this file.
"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import parity as OP, synth, torch_ref  # noqa: E402


def main():
    H, W = (int(v) for v in sys.argv[1].split("x"))
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    keys = json.load(open(os.path.join(ROOT, "tests", "golden", "c3_state_keys.json")))["keys"]
    sd = synth.synth_state_dict([(k, tuple(s)) for k, s in keys], seed=0, conditioned=True)
    x, y = OP.inputs(2, H, W, seed)
    torch.set_num_threads(os.cpu_count() or 1)
    t0 = time.time()
    ref = OP.oracle_step(sd, x, y, torch.float32)
    t1 = time.time()
    s = torch_ref.clone_state({k: v.clone() for k, v in sd.items()}, requires_grad=True)
    net = torch_ref.OracleNet(s, training=True, eps_encoder=1e-3, drop_p=0.0)
    with torch.autocast("cpu", dtype=torch.bfloat16):
        outs = net.deeplabv3_plus_xception65(x)
    loss = torch_ref.mix_softmax_ce([o.float() for o in outs], y)
    loss.backward()
    got = dict(loss=float(loss.item()),
               logits=outs[0].detach()[..., ::OP.SAMPLE, ::OP.SAMPLE].float(),
               grads={k: v.grad.detach().float() for k, v in s.items() if v.grad is not None})
    cmp = OP.compare(got, ref)
    cmp.update(size=[H, W], seed=seed, fp32_seconds=t1 - t0, autocast_seconds=time.time() - t1,
               threads=torch.get_num_threads(), torch=torch.__version__)
    print(json.dumps(cmp), flush=True)
    path = os.path.join(ROOT, "tests", "golden", "c3_autocast_sizes.json")
    data = json.load(open(path)) if os.path.exists(path) else {}
    data["%dx%d" % (H, W)] = cmp
    json.dump(data, open(path, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""One-node launcher for the reference's entry points on MI355X (replaces tools/dist_train.sh:8-9).

    python tools/launch.py --nproc 8 /path/to/SegmenTron/tools/train.py \
        --config-file configs/cityscapes_deeplabv3_plus.yaml TRAIN.BATCH_SIZE 2

The reference's own launcher calls `python -m torch.distributed.launch`, which current PyTorch
passes the rank as `--local-rank=N`; the reference's parser only knows `--local_rank`
(segmentron/utils/options.py:10), so the scripts die at argparse.  This launcher
starts one process per GPU with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT in the
environment (what default_setup.py:12-26 reads through init_method="env://") and appends
`--local_rank N` in the spelling the script expects.  It also puts THIS repository in front of
the reference checkout on PYTHONPATH, so `import segmentron` binds to the HIP hot path
(segmentron_amd/dropin.py), adds the torchvision / thop stand-ins when the image lacks the real
packages, and keeps HSA_ENABLE_IPC_MODE_LEGACY=0 (RCCL needs dmabuf IPC here).
Exit code: the first non-zero exit code of any rank (the others are terminated), else 0.
"""
import argparse
import importlib.util
import os
import signal
import socket
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def build_env(rank, nproc, port, base=None):
    env = dict(os.environ if base is None else base)
    env.update({"RANK": str(rank), "LOCAL_RANK": str(rank), "WORLD_SIZE": str(nproc),
                "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // max(nproc, 1) // 2)))
    parts = [REPO] + [p for p in env.get("PYTHONPATH", "").split(os.pathsep) if p and p != REPO]
    # tools/train.py:17 imports torchvision BEFORE segmentron: when the image has no torchvision /
    # thop, the stand-ins (segmentron_amd/shims) must already be importable
    shims = os.path.join(REPO, "segmentron_amd", "shims")
    if any(importlib.util.find_spec(n) is None for n in ("torchvision", "thop")) \
            and shims not in parts:
        parts.append(shims)
    env["PYTHONPATH"] = os.pathsep.join(parts)
    return env


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--nproc", type=int, default=1, help="processes (= GPUs) on this node")
    ap.add_argument("--master-port", type=int, default=0)
    ap.add_argument("--no-local-rank-arg", action="store_true",
                    help="do not append --local_rank N (scripts that read LOCAL_RANK themselves)")
    ap.add_argument("script")
    ap.add_argument("script_args", nargs=argparse.REMAINDER)
    a = ap.parse_args(argv)
    port = a.master_port or free_port()
    procs = []
    for r in range(a.nproc):
        cmd = [sys.executable, a.script] + list(a.script_args)
        if not a.no_local_rank_arg:
            # before the trailing KEY VALUE overrides: the reference's parser collects those
            # with nargs=REMAINDER (options.py:27-28), which would swallow a trailing option
            cmd = [sys.executable, a.script, "--local_rank", str(r)] + list(a.script_args)
        procs.append(subprocess.Popen(cmd, env=build_env(r, a.nproc, port)))
    rc = 0
    try:
        alive = set(range(a.nproc))
        while alive and rc == 0:
            for r in sorted(alive):
                code = procs[r].poll()
                if code is not None:
                    alive.discard(r)
                    if code != 0:
                        rc = code
                        break
            time.sleep(0.1)
    except KeyboardInterrupt:
        rc = 130
    finally:
        for p in procs:
            if p.poll() is None:
                p.send_signal(signal.SIGTERM)
        for p in procs:
            try:
                p.wait(timeout=10)
            except subprocess.TimeoutExpired:
                p.kill()
    return rc


if __name__ == "__main__":
    sys.exit(main())

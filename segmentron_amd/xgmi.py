"""SyncBatchNorm statistics exchange over one-hop xGMI peer writes (csrc/p2p.hip).

MI355X's eight GPUs are a full xGMI mesh: each pair has its own link.  A SyncBatchNorm exchange
is a 2C+1-double message (12 KB at C = 728) whose cost is pure latency — 584 of them sit on
the critical path of one DeepLabv3+/xception65 train step — and a ring all-reduce pays 2 (W - 1)
hops for it.  `PeerMailbox` maps every rank's mailbox into every process (hipIpc) and does the
exchange as ONE kernel: write my vector into my slot on each peer, raise a flag, wait for the W
flags here, add the W slots in rank order.  One hop; results are bit-identical on all ranks; the
launch is capturable (the exchange counter lives in device memory).

`StatsExchange` is what `parallel.use_native_rccl` takes: float64 / float32 sums that fit a mailbox
slot go through the mailbox, everything else (gradient buckets, other dtypes) through the wrapped
communicator (segmentron_amd.rccl.Communicator).  `connect()` verifies the mailbox against that
communicator before handing it out and returns the plain communicator if anything is off.

Reference call sites: tools/train.py:76 (convert_sync_batchnorm), torch's SyncBatchNorm
collectives torch/nn/modules/_functions.py:49-74,140.
"""
import ctypes
import os
import sys

import torch

from ._lib import LIB

_HANDLE_BYTES = 64
SLOT_BYTES = 128 * 1024  # one message: 16384 doubles; the in-kernel exchange of the finalize
#                          kernels needs 24 B per channel (forward) -> C up to 4096


def _gather_bytes_torch(payload):
    import torch.distributed as dist
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, payload)
    return out


class PeerMailbox:
    """In-place float64 / float32 SUM over the ranks of one node.  `gather(bytes) -> [bytes] * world`
    (rank order) carries the 64-byte IPC handles; default: torch.distributed.all_gather_object
    over the default group.  Every rank must construct it, and issue the same calls."""

    def __init__(self, rank, world, slot_bytes=SLOT_BYTES, gather=None, timeout_s=None):
        """timeout_s: bound of one in-kernel wait for a peer (default: SEG_P2P_TIMEOUT_S or the
        library's 600 s).  When it expires the exchange — and every later one, the error word is
        latched — returns NaN, so the step fails visibly; `check()` raises."""
        self.rank, self.world, self.slot_bytes = int(rank), int(world), int(slot_bytes)
        if timeout_s is None and os.environ.get("SEG_P2P_TIMEOUT_S"):
            timeout_s = float(os.environ["SEG_P2P_TIMEOUT_S"])
        self._h = self.handle = ctypes.c_void_p()
        err, mine = None, b""
        try:
            h = ctypes.c_void_p()
            LIB.call("seg_p2p_create", self.rank, self.world, self.slot_bytes, ctypes.byref(h))
            self._h = self.handle = h  # (`handle`: what the seg_*_sync entry points take)
            if timeout_s is not None:
                LIB.call("seg_p2p_set_timeout", self._h, float(timeout_s))
            buf = ctypes.create_string_buffer(_HANDLE_BYTES)
            LIB.call("seg_p2p_ipc_handle", self._h, buf)
            mine = buf.raw
        except Exception as e:  # noqa: BLE001 — the handle exchange below is collective: a rank
            err = e             # that failed still takes part (with an empty payload)
        every = [mine]
        if self.world > 1:
            every = (gather or _gather_bytes_torch)(mine)
        try:
            if err is not None:
                raise err
            if len(every) != self.world or any(len(b) != _HANDLE_BYTES for b in every):
                raise RuntimeError("xgmi.PeerMailbox: a peer has no mailbox to share")
            if self.world > 1:
                LIB.call("seg_p2p_connect", self._h, ctypes.create_string_buffer(b"".join(every)))
        except Exception:
            self.destroy()
            raise

    def fits(self, t):
        return (t.dtype in (torch.float64, torch.float32) and t.is_cuda and t.is_contiguous()
                and 0 < t.numel() * t.element_size() <= self.slot_bytes)

    def all_reduce(self, t):
        if not self.fits(t):
            raise RuntimeError("xgmi.PeerMailbox: contiguous float64 / float32 HIP tensor of at "
                               "most %d bytes required" % self.slot_bytes)
        LIB.call("seg_p2p_all_reduce_f64" if t.dtype == torch.float64 else "seg_p2p_all_reduce_f32",
                 self._h, t.data_ptr(), t.numel(), torch.cuda.current_stream(t.device).cuda_stream)
        return t

    def set_timeout(self, seconds):
        """Launches issued from now on wait at most `seconds` (a captured graph keeps its own)."""
        LIB.call("seg_p2p_set_timeout", self._h, float(seconds))

    def check(self):
        """Synchronises; raises if a peer failed to publish within the kernel's time limit (the
        results since then are NaN; the error is latched — build a new mailbox to go on)."""
        LIB.call("seg_p2p_status", self._h)

    def destroy(self):
        if self._h:
            LIB.query("seg_p2p_destroy", self._h)
            self._h = self.handle = ctypes.c_void_p()


class StatsExchange:
    """`rccl.Communicator`'s interface: mailbox for the statistics vectors, `comm` for the rest."""

    def __init__(self, mailbox, comm):
        self.mailbox, self.comm = mailbox, comm
        self.rank, self.world = mailbox.rank, mailbox.world

    def all_reduce(self, t, op="sum"):
        if op == "sum" and self.mailbox.fits(t):
            return self.mailbox.all_reduce(t)
        return self.comm.all_reduce(t, op)

    def all_reduce_many(self, tensors, op="sum"):
        return self.comm.all_reduce_many(tensors, op)

    def check(self):
        self.mailbox.check()

    def destroy(self):
        self.mailbox.destroy()
        if hasattr(self.comm, "destroy"):
            self.comm.destroy()


def connect(comm, rank, world, gather=None, rounds=4, log=sys.stderr):
    """-> StatsExchange(mailbox, comm) if the mailbox reproduces `comm.all_reduce` on this node,
    else `comm` itself (reason printed to `log`).  Collective: every rank calls it.  The check
    runs `rounds` exchanges of several sizes against the communicator, eagerly and from a
    replayed HIP graph, then the in-kernel exchange of a BatchNorm finalize (728 and 2048
    channels, eager and replayed) against the same sums, and all ranks agree on the verdict
    through `comm`."""
    box, why = None, None
    try:
        box = PeerMailbox(rank, world, gather=gather)
    except Exception as e:  # no uncached memory / IPC refused: the RCCL path stays
        why = "%s: %s" % (type(e).__name__, e)
    dev = torch.device("cuda", torch.cuda.current_device())
    ok = torch.tensor([0.0 if box is None else 1.0], dtype=torch.float64, device=dev)
    comm.all_reduce(ok, "sum")
    if ok.item() != world:  # some rank has no mailbox: nobody uses one
        if box is not None:
            box.destroy()
        if log is not None:
            log.write("[segmentron_amd.xgmi] peer mailbox unavailable (%s); statistics go "
                      "through RCCL\n" % (why or "another rank failed"))
        return comm
    # phase 1: the mailbox alone (a failure here must not change how many calls phase 2 makes)
    gen = torch.Generator(device=dev).manual_seed(1234 + rank)
    inputs = [torch.randn(n, dtype=dt, device=dev, generator=gen)
              for n, dt in ((3, torch.float64), (1457, torch.float64), (4097, torch.float64),
                            (SLOT_BYTES // 8, torch.float64), (1456, torch.float32),
                            (SLOT_BYTES // 4, torch.float32)) for _ in range(rounds)]
    src = torch.randn(1457, dtype=torch.float64, device=dev, generator=gen)
    got, a, bad = [], torch.zeros_like(src), 0.0
    from .graph import capture  # (torch.cuda.graph with the garbage collector held off)
    try:
        for x in inputs:
            got.append(box.all_reduce(x.clone()))
        # captured + replayed (the path the train step takes): two dependent exchanges per replay
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.stream(side):
            a.copy_(src)
            box.all_reduce(a)
            torch.cuda.synchronize()
            with capture(graph, stream=side):
                a.copy_(src)
                box.all_reduce(a)
                box.all_reduce(a)
        torch.cuda.current_stream().wait_stream(side)
        for _ in range(rounds):
            graph.replay()
        box.check()
    except Exception as e:
        why, bad = "%s: %s" % (type(e).__name__, e), 1.0
    # ... and the exchange INSIDE a finalize kernel (p2p.h p2p_block_exchange: what the train step
    # uses for every BatchNorm), 91 and 256 blocks, eager and replayed
    fin_in = [(torch.randn(8, 2 * C, dtype=torch.float32, device=dev, generator=gen), C)
              for C in (728, 2048)]  # (built outside the try: phase 2 walks this list on every rank)
    fin_got = []
    try:
        from . import hip_ops as K
        for part, C in fin_in:
            ones, zeros = torch.ones(C, device=dev), torch.zeros(C, device=dev)
            mean, _, _, _, cnt = K.bn_finalize_p_sync(box, part, 10.0 + rank, ones, zeros, 1e-5,
                                                      0.1, None, None)
            g2 = torch.cuda.CUDAGraph()
            side2 = torch.cuda.Stream()
            side2.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side2):
                torch.cuda.synchronize()
                with capture(g2, stream=side2):
                    mean_g, _, _, _, cnt_g = K.bn_finalize_p_sync(box, part, 10.0 + rank, ones,
                                                                  zeros, 1e-5, 0.1, None, None)
            torch.cuda.current_stream().wait_stream(side2)
            g2.replay()
            g2.replay()
            torch.cuda.synchronize()  # (the graph object goes away with this loop iteration)
            fin_got.append((mean, cnt, mean_g, cnt_g))
        box.check()
    except Exception as e:
        why, bad = "%s: %s" % (type(e).__name__, e), 1.0
    # phase 2: the same sums through the communicator
    for i, (part, C) in enumerate(fin_in):
        sums = comm.all_reduce(part.double().sum(0), "sum")
        cnt = comm.all_reduce(torch.tensor([10.0 + rank], dtype=torch.float64, device=dev), "sum")
        if bad == 0.0 and i < len(fin_got):
            want_mean = (sums[:C] / cnt).float()
            mean, c1, mean_g, c2 = fin_got[i]
            if not (torch.allclose(mean, want_mean, rtol=1e-5, atol=1e-6)
                    and torch.allclose(mean_g, want_mean, rtol=1e-5, atol=1e-6)
                    and float(c1) == float(cnt) and float(c2) == float(cnt)):
                bad = 1.0
        elif bad == 0.0:
            bad = 1.0
    for i, x in enumerate(inputs):
        want = comm.all_reduce(x.clone(), "sum")
        tol = 1e-12 if x.dtype == torch.float64 else 1e-5
        if bad == 0.0 and not torch.allclose(got[i], want, rtol=tol, atol=tol):
            bad = 1.0
    want = comm.all_reduce(comm.all_reduce(src.clone(), "sum"), "sum")
    if bad == 0.0 and not torch.allclose(a, want, rtol=1e-12, atol=1e-12):
        bad = 1.0
    verdict = torch.tensor([bad], dtype=torch.float64, device=dev)
    comm.all_reduce(verdict, "sum")
    if verdict.item() != 0.0:
        box.destroy()
        if log is not None:
            log.write("[segmentron_amd.xgmi] peer mailbox failed its check against RCCL (%s); "
                      "statistics go through RCCL\n" % (why or "mismatch on some rank"))
        return comm
    return StatsExchange(box, comm)

"""View-sharded data parallelism: one process per GPU, every rank renders its share of the step's
camera views of the SAME Gaussians, and the only exchange is the sum over views of the per-Gaussian attribute
gradients (and, once the networks are in the step, of the StyleUNet gradients) -- one all-reduce per step.

Backend-agnostic on purpose: ``nccl`` (= RCCL over xGMI) on the GPU node, ``gloo`` in the CPU tests.
"""
from __future__ import annotations

import os
from typing import List, Sequence

import torch
import torch.distributed as dist


def init_distributed(world: int, local_rank: int):
    """One rank per GPU over RCCL (backend "nccl").  With fewer GPUs than ranks RCCL cannot run (two ranks on one device): the ranks
    then share devices over gloo -- a functional run of the N > 1 control flow, reported as such in the JSON line and NEVER a
    measurement.  AG_DIST_BACKEND forces either.  Returns (backend, local device index)."""
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    n_dev = torch.cuda.device_count()
    backend = os.environ.get("AG_DIST_BACKEND") or ("nccl" if n_dev >= world else "gloo")
    if backend == "nccl" and n_dev < world:
        raise RuntimeError(f"--gpus {world} with {n_dev} visible GPU(s): RCCL needs one device per rank (AG_DIST_BACKEND=gloo for a functional run)")
    local = local_rank % max(n_dev, 1)
    torch.cuda.set_device(local)
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    else:
        dist.init_process_group(backend)
    return backend, local


def views_of_rank(n_views: int, rank: int, world: int) -> List[int]:
    """Round-robin assignment of the step's views; every view is rendered exactly once."""
    return list(range(rank, n_views, world))


class ViewGradExchange:
    """The exchange step of view-sharded rendering (``bench.py --gpus N``): every rank renders its own view per step; the
    per-Gaussian attribute gradients of a rank's views are summed on the device and the ranks all-reduce the sums once per iteration of
    ``every`` steps (BASELINE configs[3]: a 16-view iteration over N ranks = 16 // N steps of a rank).

    Pipeline (one stream per in-flight view = "slot", one communication stream)::

        slot stream k:  raster forward + backward of the view  ->  wait consumed[k]  ->  pack the gradient arrays into packs[k]
        comm stream:    wait for that pack  ->  acc = / += packs[k]  ->  record consumed[k]  ->  every `every`-th step: all-reduce(acc)

    A slot's raster kernels wait for nothing (the previous reader of its gradient arrays is the pack on its own stream); its pack waits only
    for the communication stream's copy of the slot's PREVIOUS pack, ``n_slots`` steps back -- so the views of the next steps run under an
    all-reduce, and the all-reduce of an iteration is ordered before the first addition of the next one by the communication stream
    itself.  (Rounds 1-4 made every step's kernels wait for the previous step's all-reduce: compute and exchange never overlapped.)

    ``acc`` holds the reduced sums of the last completed iteration once the communication stream has been joined (``join()``).  On a CPU
    device (gloo tests) the same control flow runs without streams."""

    def __init__(self, rows: int, cols: int, device, n_slots: int, every: int):
        self.dev = torch.device(device)
        self.cuda = self.dev.type == "cuda"
        self.every = max(1, int(every))
        self.packs = [torch.zeros((rows, cols), dtype=torch.float32, device=self.dev) for _ in range(max(1, n_slots))]
        self.acc = torch.zeros((rows, cols), dtype=torch.float32, device=self.dev)
        self.comm = torch.cuda.Stream(self.dev) if self.cuda else None
        self.consumed = [torch.cuda.Event() for _ in self.packs] if self.cuda else None
        self.released = torch.cuda.Event() if self.cuda else None      # recorded by release(): the reader of `acc` is done with the previous iteration
        self._released_pending = False
        if self.cuda:
            self.comm.wait_stream(torch.cuda.current_stream(self.dev))   # the zero fills above ran on the constructing stream
        self.count = 0                 # steps submitted since the last reset()
        self.reduced = 0               # all-reduces issued
        self.hold_back = False         # True: accumulate only, never all-reduce (the check of bench.py reduces by hand)
        self.checksums = None          # a 2-element float64 tensor: += (sum, sum of magnitudes) of every pack (the untimed check)

    def reset(self):
        """Start the next step on an iteration boundary."""
        self.count = 0

    def submit(self, slot: int, grads: Sequence[torch.Tensor], producer_stream=None):
        """Pack ``grads`` (arrays [rows, c_i], sum of c_i = cols) of the view just rendered on ``producer_stream`` and hand them to the
        communication stream.  Never blocks the host (RCCL) -- gloo's all-reduce is synchronous."""
        k = int(slot) % len(self.packs)
        j = self.count % self.every
        if self.cuda:
            if producer_stream is None:
                producer_stream = torch.cuda.current_stream(self.dev)
            with torch.cuda.stream(producer_stream):
                producer_stream.wait_event(self.consumed[k])          # (a no-op until the event has been recorded once)
                torch.cat(list(grads), dim=1, out=self.packs[k])
            self.comm.wait_stream(producer_stream)
            with torch.cuda.stream(self.comm):
                if j == 0 and self._released_pending:
                    # the first view of an iteration OVERWRITES acc: not before the stream that read the previous iteration's sum (join) is through
                    self.comm.wait_event(self.released)
                    self._released_pending = False
                self._accumulate(k, j)
                self.consumed[k].record(self.comm)
                self._reduce(j)
        else:
            torch.cat(list(grads), dim=1, out=self.packs[k])
            self._accumulate(k, j)
            self._reduce(j)
        self.count += 1

    def _accumulate(self, k, j):
        if j == 0:
            self.acc.copy_(self.packs[k])
        else:
            self.acc.add_(self.packs[k])
        if self.checksums is not None:
            self.checksums[0] += self.packs[k].sum(dtype=torch.float64)
            self.checksums[1] += self.packs[k].abs().sum(dtype=torch.float64)

    def _reduce(self, j):
        if j == self.every - 1 and not self.hold_back and dist.is_initialized() and dist.get_world_size() > 1:
            dist.all_reduce(self.acc)
            self.reduced += 1

    def join(self, stream=None):
        """``stream`` (default: the current one) waits for everything submitted so far; work queued on it AFTER this call (the reader of
        ``acc``) is what the next iteration's first overwrite of ``acc`` waits for -- call ``release(stream)`` once the reads are queued."""
        if self.cuda:
            (stream or torch.cuda.current_stream(self.dev)).wait_stream(self.comm)

    def release(self, stream=None):
        """The reads of ``acc`` queued on ``stream`` so far must finish before the next iteration's first view overwrites it."""
        if self.cuda:
            self.released.record(stream or torch.cuda.current_stream(self.dev))
            self._released_pending = True


class GradSync:
    """Packs a fixed list of gradient tensors into one flat buffer and all-reduces it (sum, optionally mean).

    The flat buffer is allocated once; ``start`` packs and launches the collective (async), ``finish`` waits and
    scatters the reduced values back into the ``.grad`` fields."""

    def __init__(self, params: Sequence[torch.Tensor], average: bool = False):
        self.params = list(params)
        self.average = average
        n = sum(p.numel() for p in self.params)
        self.flat = torch.zeros(n, dtype=torch.float32, device=self.params[0].device)
        self.work = None

    def start(self):
        off = 0
        for p in self.params:
            g = p.grad if p.grad is not None else torch.zeros_like(p)
            self.flat[off:off + p.numel()].copy_(g.reshape(-1))
            off += p.numel()
        if dist.is_initialized() and dist.get_world_size() > 1:
            self.work = dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, async_op=True)

    def finish(self):
        if self.work is not None:
            self.work.wait()
            self.work = None
        if self.average and dist.is_initialized():
            self.flat /= dist.get_world_size()
        off = 0
        for p in self.params:
            p.grad = self.flat[off:off + p.numel()].view_as(p).clone()
            off += p.numel()


class BucketedGradSync:
    """Data-parallel gradient exchange for the networks (223.7 M fp32 = 895 MB per step), overlapped with
    the backward pass.

    * All gradients live in ONE flat buffer; every ``param.grad`` is a view into it, so nothing is packed or unpacked.
    * The buffer is cut into buckets of ``bucket_bytes`` following the REVERSE registration order (the order autograd
      finishes them in); a post-accumulate hook counts finished parameters and launches the bucket's all-reduce
      (async, on the collective's own stream) the moment its last gradient lands -- the remaining backward keeps
      running on the compute stream.
    * xGMI is point-to-point (7 links x ~153 GB/s per GPU) and ring collectives are per-link bound, so buckets are few
      and large (default 128 MB: ~7 collectives per step) rather than DDP's 25 MB.

    * A bucket's gradients can come from several HIP streams; each hook records an event on its own stream.  The collective is
      issued under a dedicated COMMUNICATION stream that waits for those events: no compute stream is made to wait for the other
      branches' gradients (round 2 made the stream of whichever hook fired last wait for all of them, which serialised that
      backward branch behind the others exactly where the overlap was designed in).
    * One ``backward()`` per step.  A second gradient for the same parameter in one step raises (its bucket may already be
      reducing); ``defer_to_finish=True`` supports several backward calls per step by reducing everything in ``finish()``.

    Usage per step: ``sync.zero()`` -> forward/backward (hooks fire) -> ``sync.finish()`` -> optimizer step.
    Works on any backend (``nccl`` = RCCL on the GPU node, ``gloo`` in the CPU tests).

    World size 1 (``flat_when_single=False``, the default): nothing is exchanged, so nothing is packed either -- no flat buffer, no
    hooks, ``zero()`` drops the gradients (``.grad = None``: autograd then hands every parameter its gradient tensor as it is instead
    of adding it into a pre-existing one) and ``finish()`` is empty (the autograd engine itself joins the backward's streams with the
    caller's).  Per training step of the avatar that is 657 accumulate kernels, 657 Python hook calls with an event each and one
    895-MB memset less.  ``flat_when_single=True`` keeps the flat storage and the hooks (tests of the bucket logic on one rank)."""

    def __init__(self, params: Sequence[torch.nn.Parameter], bucket_bytes: int = 128 << 20, average: bool = True,
                 defer_to_finish: bool = False, flat_when_single: bool = False):
        self.params = [p for p in params if p.requires_grad]
        self.average = average
        self.defer_to_finish = defer_to_finish
        self._world = dist.get_world_size() if dist.is_initialized() else 1
        self._passthrough = self._world == 1 and not flat_when_single
        if self._passthrough:
            self._flat, self.buckets, self._handles, self._works = None, [], [], []
            return
        n = sum(p.numel() for p in self.params)
        dev = self.params[0].device
        self._cuda = dev.type == "cuda"
        self._flat = torch.zeros(n, dtype=torch.float32, device=dev)
        # reverse order: the last-registered parameters (decoder heads) get their gradients first
        order = list(reversed(self.params))
        self.buckets = []          # [start, end) element ranges of the flat buffer
        self._bucket_of = {}
        self._pending_init = []
        off, b_start, b_count = 0, 0, 0
        cap = max(1, bucket_bytes // 4)
        for p in order:
            if off - b_start >= cap:
                self.buckets.append((b_start, off))
                self._pending_init.append(b_count)
                b_start, b_count = off, 0
            p.grad = self._flat[off:off + p.numel()].view_as(p)
            self._bucket_of[p] = len(self.buckets)
            off += p.numel()
            b_count += 1
        self.buckets.append((b_start, off))
        self._pending_init.append(b_count)
        self._comm = torch.cuda.Stream(dev) if self._cuda else None      # collectives are ordered after the gradients' events HERE
        self._arm()
        self._handles = [p.register_post_accumulate_grad_hook(self._on_grad) for p in self.params]

    @property
    def flat(self) -> torch.Tensor:
        """The flat gradient buffer every ``param.grad`` is a view of.  It exists when gradients are exchanged (world size > 1, or
        ``flat_when_single=True``); the world-size-1 pass-through keeps no such buffer -- read the parameters' ``.grad`` there (they are
        ordinary tensors, set to None by ``zero()``), or construct with ``flat_when_single=True``."""
        if self._flat is None:
            raise AttributeError("BucketedGradSync is a pass-through at world size 1 (no flat gradient buffer, .grad are plain tensors); "
                                 "construct it with flat_when_single=True to keep the flat buffer on one rank")
        return self._flat

    def _arm(self):
        self._pending = list(self._pending_init)
        self._works = []
        self._fired = set()
        self._events = [[] for _ in self.buckets]      # per bucket: one event per gradient, on the stream that accumulated it
        self._launched = [False] * len(self.buckets)
        self._next = 0                                  # collectives are issued strictly in bucket order (see _on_grad)
        self.launch_order = []                          # the order they were issued in this step (the same on every rank by construction)

    def zero(self):
        """Start of a step: clear the flat gradient buffer and re-arm the buckets."""
        if self._passthrough:
            for p in self.params:
                p.grad = None
            return
        if any(w is not None for w in self._works):
            raise RuntimeError("BucketedGradSync.zero() while collectives of the previous step are in flight: call finish() first")
        self._flat.zero_()
        self._arm()

    def _launch(self, b):
        """All-reduce bucket b.  Its gradients may have been accumulated on several HIP streams (the networks run their decoder
        branches on side streams and autograd replays every node on its recording stream).  The communication stream waits for every
        gradient's own event and the collective is issued with that stream current, so RCCL orders itself after exactly the
        producers of this bucket and nothing else waits."""
        s, e = self.buckets[b]
        self._launched[b] = True
        self.launch_order.append(b)
        if self._cuda:
            for ev in self._events[b]:
                self._comm.wait_event(ev)
            self._events[b] = []
            with torch.cuda.stream(self._comm):
                self._works.append(dist.all_reduce(self._flat[s:e], op=dist.ReduceOp.SUM, async_op=True))
            return
        self._events[b] = []
        self._works.append(dist.all_reduce(self._flat[s:e], op=dist.ReduceOp.SUM, async_op=True))

    def _on_grad(self, p):
        # autograd accumulated into the existing .grad view in place; guard against it having been replaced
        if p.grad.data_ptr() < self._flat.data_ptr() or p.grad.data_ptr() >= self._flat.data_ptr() + self._flat.numel() * 4:
            raise RuntimeError("a parameter's .grad was re-allocated; use BucketedGradSync.zero(), not zero_grad(set_to_none=True)")
        b = self._bucket_of[p]
        if self._cuda:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self._flat.device))     # the hook runs on the stream that accumulated this gradient
            self._events[b].append(ev)
        if self.defer_to_finish:
            return
        if id(p) in self._fired:
            # a second backward() in the same step (one loss per view, gradient accumulation) would write into a bucket whose
            # all-reduce is already in flight or done: replicas diverge silently.  Refuse.
            raise RuntimeError("BucketedGradSync: a parameter received a second gradient in one step after its bucket was armed for "
                               "launch; call backward() once per step, or construct with defer_to_finish=True (all buckets are then "
                               "reduced in finish())")
        self._fired.add(id(p))
        self._pending[b] -= 1
        # Launch in BUCKET-INDEX order only: bucket b goes out when it and every bucket before it are complete.  The order in which the
        # gradients of different buckets land is whatever the autograd engine's queues yield (the backward runs on several HIP streams),
        # and it may differ between ranks; collectives of one communicator must be issued in the same order on every rank or they pair
        # up wrongly / hang (the reason DDP launches its buckets in order).  A bucket that completes early simply waits for its
        # predecessors; finish() flushes whatever is left, in order.
        if self._world > 1:
            while self._next < len(self.buckets) and self._pending[self._next] == 0:
                self._launch(self._next)
                self._next += 1

    def finish(self):
        """End of backward: launch the buckets that have not been launched (parameters without gradient this step, or
        ``defer_to_finish``), wait, average."""
        if self._passthrough:
            return
        if self._world > 1:
            for b in range(len(self.buckets)):           # ascending: the same order on every rank
                if not self._launched[b]:
                    self._launch(b)
            self._next = len(self.buckets)
            for w in self._works:
                w.wait()                                   # the CURRENT stream waits for the collective (and the host for gloo)
            if self._cuda:
                torch.cuda.current_stream(self._flat.device).wait_stream(self._comm)
            self._works = []
            if self.average:
                self._flat /= self._world
        elif self._cuda:
            cur = torch.cuda.current_stream(self._flat.device)
            for evs in self._events:
                for ev in evs:
                    cur.wait_event(ev)
        self._events = [[] for _ in self.buckets]

    def close(self):
        for h in self._handles:
            h.remove()

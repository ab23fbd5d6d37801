"""Batch data-parallel gradient exchange: one process per GPU, RCCL over xGMI.

The reference is single-GPU (config.py:70,76-77); SURVEY.md 8(e) defines the scale-out: samples are
independent (no batch statistics in the generator), so the only exchange per step is
all-reduce(sum) of the generator gradients, followed by a 1/world_size scale that is folded into the
Adam kernel (``grad_scale``).  Gradients live in ONE flat slab; it is cut into buckets at generator
block boundaries and each bucket's all-reduce is issued on a dedicated communication stream as soon
as autograd has accumulated its last gradient -- backward visits the top-resolution (most expensive)
block first, so every bucket but the last small one overlaps with the remaining backward compute.
xGMI is point-to-point (7 links x ~153 GB/s per GPU): buckets are a few MB, large enough to be
bandwidth- rather than latency-bound per link, small enough to start early.

Device-agnostic on purpose: the same class runs over ``gloo`` on CPU tensors in the world_size-2 tests.
"""
import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise torch.distributed from torchrun's env (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*)."""
    import os
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            # "nccl" IS RCCL on ROCm.  DF_DIST_BACKEND=gloo lets several ranks share one GPU (single-GPU test boxes)
            backend = os.environ.get("DF_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if torch.cuda.is_available():
            torch.cuda.set_device(local_rank % torch.cuda.device_count())
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


class GradSync(object):
    """Bucketed, backward-overlapped all-reduce of a flat gradient slab.

    ``buckets``: list of ``(offset, numel, [param tensors])`` covering ``flat_grad``; each param's ``.grad``
    must be a view into its bucket's range (Trainer arranges that)."""

    def __init__(self, flat_grad, buckets, group=None):
        self.flat_grad = flat_grad
        self.buckets = buckets
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.enabled = self.world > 1
        self._pending = [0] * len(buckets)
        self._works = []
        self._comm_stream = torch.cuda.Stream() if (self.enabled and flat_grad.is_cuda) else None
        if self.enabled:
            for bi, (_, _, params) in enumerate(buckets):
                for p in params:
                    p.register_post_accumulate_grad_hook(self._make_hook(bi))
        self.begin_step()

    def begin_step(self):
        self._pending = [len(b[2]) for b in self.buckets]
        self._works = []

    def _make_hook(self, bi):
        def hook(_p):
            self._pending[bi] -= 1
            if self._pending[bi] == 0:
                self._launch(bi)
        return hook

    def _launch(self, bi):
        off, n, _ = self.buckets[bi]
        chunk = self.flat_grad[off:off + n]
        if self._comm_stream is not None:
            self._comm_stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self._comm_stream):
                w = dist.all_reduce(chunk, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        else:
            w = dist.all_reduce(chunk, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self._works.append(w)

    def finish(self):
        """Block the compute stream until every bucket is reduced; returns the scale (1/world) the
        optimizer must apply to the summed gradients."""
        if not self.enabled:
            return 1.0
        for bi, left in enumerate(self._pending):
            if left > 0:                      # a bucket whose hooks did not all fire (unused params): reduce it now
                self._pending[bi] = 0
                self._launch(bi)
        for w in self._works:
            w.wait()
        if self._comm_stream is not None:
            torch.cuda.current_stream().wait_stream(self._comm_stream)
        return 1.0 / self.world


def shard_batch(global_batch, rank, world):
    """rank r of N takes samples [r*B/N, (r+1)*B/N); B must divide evenly."""
    if global_batch % world != 0:
        raise ValueError("global batch %d is not divisible by world size %d" % (global_batch, world))
    per = global_batch // world
    return rank * per, per

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

Device-agnostic on purpose: the same class runs over ``gloo`` on CPU tensors in the world_size-2 tests, and over
``gloo`` with several ranks sharing ONE GPU (``DF_DIST_BACKEND=gloo``; buckets are staged through host memory) in the
-m gpu test that runs the real ``Trainer`` with world > 1 on a single-GPU box.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise torch.distributed from torchrun's env (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        # dmabuf IPC (this driver has no legacy IPC: RCCL's hipIpcGetMemHandle fails without it); effective only if the HIP runtime has not
        # been initialised yet -- launchers (bench.py, the tests) also export it
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            # "nccl" IS RCCL on ROCm.  DF_DIST_BACKEND=gloo lets several ranks share one GPU (single-GPU test boxes)
            backend = os.environ.get("DF_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if torch.cuda.is_available():
            torch.cuda.set_device(local_rank % torch.cuda.device_count())
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def device_identity(index=None):
    """A string that is the same for two ranks iff they drive the same physical GPU: host name + PCI domain:bus:device (+ the UUID
    where the runtime reports one).  The host name keeps two nodes of a multi-node job apart on runtimes without UUIDs (identical PCI
    ids on every node)."""
    if not torch.cuda.is_available():
        return "cpu"
    index = torch.cuda.current_device() if index is None else index
    pr = torch.cuda.get_device_properties(index)
    parts = ["%s/%s:%s:%s" % (os.uname().nodename, getattr(pr, "pci_domain_id", "?"), getattr(pr, "pci_bus_id", "?"),
                              getattr(pr, "pci_device_id", "?"))]
    uuid = getattr(pr, "uuid", None)
    if uuid is not None:
        parts.append(str(uuid))
    if parts[0].endswith("/?:?:?") and uuid is None:      # nothing physical to go by: fall back to (host, visible index)
        parts = ["%s/%s/%d" % (os.uname().nodename, os.environ.get("HIP_VISIBLE_DEVICES", os.environ.get("ROCR_VISIBLE_DEVICES", "")), index)]
    return "|".join(parts)


def verify_world(group=None):
    """What the job REALLY runs on, asked of the communication backend itself (bench.py's `rccl_ranks`): an on-device all-reduce of
    ones over the production backend counts the participating ranks, an all-gather of `device_identity()` lists the physical GPUs.
    Raises when the RCCL ("nccl") backend is asked to run several ranks on ONE device -- such a job would print a multi-GPU line
    for single-GPU hardware (the sharing configuration exists for the tests only, over gloo with DF_DIST_BACKEND=gloo)."""
    if not dist.is_initialized():
        return {"ranks": 1, "backend": None, "devices": [device_identity()], "distinct_devices": 1}
    backend = dist.get_backend(group)
    world = dist.get_world_size(group)
    on_dev = torch.cuda.is_available() and backend != "gloo"
    one = torch.ones(1, dtype=torch.float32, device="cuda" if on_dev else "cpu")
    dist.all_reduce(one, op=dist.ReduceOp.SUM, group=group)
    counted = int(round(float(one.item())))
    ids = [None] * world
    dist.all_gather_object(ids, device_identity(), group=group)
    return check_world(backend, world, counted, ids)


def check_world(backend, world, counted, ids):
    """The verdict of :func:`verify_world` (separate so that the refusal rules are testable without a multi-GPU node)."""
    distinct = len(set(ids))
    if counted != world:
        raise RuntimeError("all-reduce of ones over %s returned %d, world size is %d" % (backend, counted, world))
    if backend == "nccl" and distinct != world:
        raise RuntimeError("RCCL job with %d ranks on %d distinct device(s) %s: one rank per GPU is required (set DF_DIST_BACKEND=gloo "
                           "for the single-GPU sharing test configuration)" % (world, distinct, sorted(set(ids))))
    return {"ranks": counted, "backend": backend, "devices": list(ids), "distinct_devices": distinct}


class GradSync(object):
    """Bucketed, backward-overlapped all-reduce of a flat gradient slab.

    ``buckets``: list of ``(offset, numel, [param tensors])`` covering ``flat_grad``; each param's ``.grad``
    must be a view into its bucket's range (Trainer arranges that).

    ``profile=True`` brackets the exchange with HIP events (bench.py): per step, ``comm_ms`` = time the communication stream
    spent on the all-reduces, ``exposed_ms`` = time the compute stream waited for them after backward finished;
    ``hidden_ms = comm_ms - exposed_ms`` ran under backward compute."""

    def __init__(self, flat_grad, buckets, group=None, profile=False, force=False):
        """``force=True`` runs the exchange even with world_size 1 (the -m gpu test that drives the RCCL path -- side stream,
        hooks, async all-reduce, event profile -- on a single-GPU box, where the all-reduce is the identity)."""
        self.flat_grad = flat_grad
        self.buckets = buckets
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.enabled = self.world > 1 or bool(force and dist.is_initialized())
        self.backend = dist.get_backend(group) if dist.is_initialized() else None
        # gloo with device buffers: stage each bucket through host memory (test-only configuration: ranks sharing one GPU)
        self._host_staged = bool(self.enabled and flat_grad.is_cuda and self.backend == "gloo")
        self._pending = [0] * len(buckets)
        self._works = []
        self._comm_stream = torch.cuda.Stream() if (self.enabled and flat_grad.is_cuda and not self._host_staged) else None
        self.profile = bool(profile and self._comm_stream is not None)
        self._ev = []                  # per step: (first-launch start, [bucket end events], backward-done, finish-done)
        self._cur = None
        self.launch_order = []         # bucket indices in the order their all-reduces were issued during the last step
        self.suspended = False         # True: the hooks do nothing (a hipGraph capture of forward + backward is in progress; the
        #                                exchange then is reduce_all() after the replay, Trainer._graph_step)
        if self.enabled:
            for bi, (_, _, params) in enumerate(buckets):
                for p in params:
                    p.register_post_accumulate_grad_hook(self._make_hook(bi))
        self.begin_step()

    def begin_step(self):
        self._pending = [len(b[2]) for b in self.buckets]
        self._works = []
        self.launch_order = []
        self._cur = {"start": None, "ends": []} if self.profile else None

    def _make_hook(self, bi):
        def hook(_p):
            if self.suspended:
                return
            self._pending[bi] -= 1
            if self._pending[bi] == 0:
                self._launch(bi)
        return hook

    def _launch(self, bi):
        off, n, _ = self.buckets[bi]
        chunk = self.flat_grad[off:off + n]
        self.launch_order.append(bi)
        if self._host_staged:
            host = chunk.cpu()                                       # synchronises with the compute stream
            dist.all_reduce(host, op=dist.ReduceOp.SUM, group=self.group)
            chunk.copy_(host)
            return
        if self._comm_stream is not None:
            # ONE schedule for production and profiled runs: the collective is enqueued from the communication stream (RCCL's internal
            # stream then waits for the gradients through it) and the communication stream is made to wait for the collective at once
            # (`w.wait()` is a device-side stream dependency, the host does not block).  finish() then only joins the two streams.
            # profile=True adds event records on the communication stream and nothing else.
            self._comm_stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self._comm_stream):
                if self._cur is not None and self._cur["start"] is None:
                    self._cur["start"] = torch.cuda.Event(enable_timing=True)
                    self._cur["start"].record()
                w = dist.all_reduce(chunk, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
                w.wait()
                if self._cur is not None:
                    e = torch.cuda.Event(enable_timing=True)
                    e.record()
                    self._cur["ends"].append(e)
            return
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
        if self._cur is not None:
            self._cur["bwd_done"] = torch.cuda.Event(enable_timing=True)
            self._cur["bwd_done"].record()
        for w in self._works:
            w.wait()
        if self._comm_stream is not None:
            torch.cuda.current_stream().wait_stream(self._comm_stream)
        if self._cur is not None:
            self._cur["done"] = torch.cuda.Event(enable_timing=True)
            self._cur["done"].record()
            self._ev.append(self._cur)
            self._cur = None
        return 1.0 / self.world

    def reduce_all(self):
        """The whole flat slab in ONE all-reduce on the current stream (no overlap with backward: the graph-replay schedule, where the
        backward pass is a single graph launch that hooks cannot interleave with); returns the 1/world scale like finish()."""
        if not self.enabled:
            return 1.0
        self.launch_order = ["all"]
        if self._host_staged:
            host = self.flat_grad.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM, group=self.group)
            self.flat_grad.copy_(host)
        else:
            dist.all_reduce(self.flat_grad, op=dist.ReduceOp.SUM, group=self.group)
        return 1.0 / self.world

    def timing(self, reset=True):
        """Mean per-step milliseconds over the profiled steps since the last reset (synchronises the device)."""
        if not self._ev:
            return None
        torch.cuda.synchronize()
        comm = exposed = 0.0
        for s in self._ev:
            if s["start"] is None or not s["ends"]:
                continue
            # time from the first all-reduce's start to the last one's end on the communication stream: an UPPER bound of
            # the busy time (the stream idles between buckets while it waits for the next gradients)
            comm += s["start"].elapsed_time(s["ends"][-1])
            exposed += max(s["bwd_done"].elapsed_time(s["done"]), 0.0)
        n = len(self._ev)
        out = {"steps": n, "comm_span_ms": comm / n, "exposed_ms": exposed / n, "hidden_ms": max(comm - exposed, 0.0) / n,
               "buckets": len(self.buckets), "bytes": int(self.flat_grad.numel()) * 4}
        if reset:
            self._ev = []
        return out


def broadcast_(t, src=0, group=None):
    """In-place broadcast of a tensor from rank ``src``; device tensors over gloo (the single-GPU sharing configuration) are staged
    through host memory.  No-op without an initialised process group."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return t
    if t.is_cuda and dist.get_backend(group) == "gloo":
        h = t.cpu()
        dist.broadcast(h, src=src, group=group)
        t.copy_(h)
    else:
        dist.broadcast(t, src=src, group=group)
    return t


def broadcast_trainer_state(slabs, scalars, src=0, group=None):
    """Make every rank start from rank ``src``'s training state: ``slabs`` = flat device tensors (parameters, Adam slots) broadcast in
    place; ``scalars`` = list of Python numbers (global step, learning rate, Adam step counts) returned as rank ``src`` holds them.
    Only rank 0 writes checkpoints (Trainer.train), so after a restore-on-start on a node without a shared file system -- or whenever
    one rank does not see the file -- the ranks would otherwise train divergent replicas with different loop lengths."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return list(scalars)
    for t in slabs:
        broadcast_(t, src, group)
    h = torch.tensor([float(v) for v in scalars], dtype=torch.float64)
    dev = slabs[0].device if (slabs and slabs[0].is_cuda and dist.get_backend(group) != "gloo") else torch.device("cpu")
    h = h.to(dev)
    dist.broadcast(h, src=src, group=group)
    return h.cpu().tolist()


def all_equal_across_ranks(value, group=None):
    """True iff the float64 scalar / small vector ``value`` is bit-identical on every rank (MIN == MAX over the group)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return True
    v = torch.as_tensor(value, dtype=torch.float64).reshape(-1).clone()
    on_dev = torch.cuda.is_available() and dist.get_backend(group) != "gloo"
    lo = v.cuda() if on_dev else v.cpu()
    hi = lo.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=group)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=group)
    return bool((lo == hi).all().item())


def shard_batch(global_batch, rank, world):
    """rank r of N takes samples [r*B/N, (r+1)*B/N); B must divide evenly."""
    if global_batch % world != 0:
        raise ValueError("global batch %d is not divisible by world size %d" % (global_batch, world))
    per = global_batch // world
    return rank * per, per

"""Counterpart of the reference's ``Trainer`` / ``Trainer3`` for the ``arch='de'`` velocity-field path.

Reproduces ``build_model`` (trainer.py:136-184; trainer3.py:14-63) and the hot loop of ``train_``
(trainer.py:265-288) on PyTorch-ROCm + libdeepfluids_hip.so:

    G_s, G_var = GeneratorBE|GeneratorBE3(y, filters, output_shape, num_conv, repeat)
    G_         = curl(G_s) | jacobian3(G_s)[1]
    G_jaco_, _ = jacobian(G_) | jacobian3(G_)
    g_loss     = w1*mean|G_ - x| + w2*mean|G_jaco_ - x_jaco|           (x_jaco = jacobian(x), trainer.py:29-32)
    Adam(g_lr, beta1=.5, beta2=.999).minimize(g_loss, var_list=G_var)  (TF1 "epsilon-hat" Adam)
    g_lr <- lr_min + .5 (lr_max - lr_min)(cos(step*pi/max_step) + 1)   (assigned AFTER each step)

Flag names and defaults are the reference's (config.py:14-70).  Parameters, gradients and Adam slots
each live in one flat slab (one fused optimizer launch, bucketed all-reduce for data parallelism).
"""
import json
import math
import os
import re
import time
from types import SimpleNamespace

import numpy as np
import torch

from . import ops
from . import _lib as _lib_mod
from .ops import curl, curl3, jacobian, jacobian3, l1_mean, mse_mean, get_conv_shape, _ptr, _stream, call
from .model import GeneratorBE, GeneratorBE3, AE, AE3, DiscriminatorPatch, DiscriminatorPatch3
from .ops import concat, kl_bernoulli
from .dist import GradSync, broadcast_trainer_state


def default_config(**over):
    """config.py:14-70 defaults for the fields the velocity-field path reads."""
    c = dict(is_3d=False, res_x=96, res_y=128, res_z=32, repeat=0, filters=128, num_conv=4, use_curl=True,
             w1=1.0, w2=1.0, arch="de", batch_size=8, max_epoch=100, lr_max=1e-4, lr_min=2.5e-6,
             optimizer="adam", beta1=0.5, beta2=0.999, lr_update="decay", lr_update_step=120000,
             start_step=0, random_seed=123, num_samples=21000, c_num=3, use_curl3_alias=True,
             z_num=16, use_sparse=False, sparsity=0.01, w4=1.0, w5=1.0, p_num=2, x_channels=None, w3=0.005,
             log_step=500, test_step=1000, test_batch_size=100, model_dir=None, load_path="", code_path="", save_sec=3600,   # config.py:62-68
             fused_tail=True, graph=False, direct_grads=True)
    c.update(over)
    return SimpleNamespace(**c)


_CKPT_RE = re.compile(r"^model\.ckpt-(\d+)\.npz$")


def latest_checkpoint(model_dir):
    """``tf.train.latest_checkpoint``: the ``model.ckpt-<global step>.npz`` with the largest step in ``model_dir`` (None if there is none)."""
    best = None
    if model_dir and os.path.isdir(model_dir):
        for f in os.listdir(model_dir):
            m = _CKPT_RE.match(f)
            if m and (best is None or int(m.group(1)) > best[0]):
                best = (int(m.group(1)), os.path.join(model_dir, f))
    return None if best is None else best[1]


def _detached(m):
    """The step's outputs without their autograd graph.  A graph-mode trainer never hands out tensors that keep the graph alive: the
    variables' AccumulateGrad nodes live as long as any graph that reaches them, remember the stream they were created on, and a later
    capture on another stream would have to synchronise with that stream -- which a stream capture cannot (hipStreamEndCapture dies)."""
    return SimpleNamespace(**{k: (v.detach() if isinstance(v, torch.Tensor) else v) for k, v in vars(m).items()})


class Trainer(object):
    _restore_in_base = True          # subclasses that add variables after Trainer.__init__ (GANTrainer's D slab) restore at THEIR end

    def __init__(self, config, device="cuda", name="G"):
        self.config = config
        self.device = torch.device(device)
        self.is_3d = bool(config.is_3d)
        self.name = name
        self.b_num = config.batch_size
        self.c_num = config.c_num
        spatial = [config.res_z, config.res_y, config.res_x] if self.is_3d else [config.res_y, config.res_x]
        if config.use_curl:                                     # trainer.py:48-53
            self.output_shape = spatial + [3 if self.is_3d else 1]
        else:
            self.output_shape = spatial + [3 if self.is_3d else 2]
        self.filters, self.num_conv, self.repeat = config.filters, config.num_conv, config.repeat
        self.w1, self.w2 = config.w1, config.w2
        self.fused_tail = bool(getattr(config, "fused_tail", False))    # Trainer ('de') only; the AE / GAN graphs need J(u) / vorticity tensors
        self.beta1, self.beta2, self.eps = config.beta1, config.beta2, 1e-8
        self.step = config.start_step                            # trainer.py:65
        epochs_per_step = config.batch_size / float(config.num_samples)          # data.py:50
        self.max_step = int(config.max_epoch // epochs_per_step)                # trainer.py:67
        self.lr_update = config.lr_update
        self.g_lr = config.lr_max                                # trainer.py:72
        self._adam_t = 0
        ops.set_random_seed(config.random_seed)
        self._build_variables()
        self.grad_sync = None
        # graph=True: the whole step -- forward, backward, optimizer -- is captured ONCE per input shape into a hipGraph and replayed (the
        # counterpart of `sess.run(g_optim)` executing a pre-built TF graph, trainer.py:265-269: no per-op host cost); see _graph_step
        self.use_graph = bool(getattr(config, "graph", False))
        self._graphs, self._graph_warm = {}, set()
        self.save_sec = getattr(config, "save_sec", 3600)      # config.py:68 -> Supervisor(save_model_secs=...), trainer.py:110-117
        if self._restore_in_base:
            self._auto_restore()

    # ---- variables: created by one shape-only pass through the generator, then moved into flat slabs ----
    def _create_variables(self):
        gen = GeneratorBE3 if self.is_3d else GeneratorBE
        z = torch.zeros((1, self.c_num), dtype=torch.float32, device=self.device)
        with torch.no_grad():
            gen(z, self.filters, self.output_shape, name=self.name, num_conv=self.num_conv, repeat=self.repeat)

    def _build_variables(self):
        prefix = self.name + "/"
        existing = {k for k in ops.all_variables() if k.startswith(prefix)}
        if not existing:
            self._create_variables()
        names = [k for k in ops.all_variables() if k.startswith(prefix)]
        vars_ = ops.all_variables()
        total = sum(vars_[k].numel() for k in names)
        self.flat_p = torch.empty(total, dtype=torch.float32, device=self.device)
        self.flat_g = torch.zeros(total, dtype=torch.float32, device=self.device)
        self.flat_m = torch.zeros(total, dtype=torch.float32, device=self.device)
        self.flat_v = torch.zeros(total, dtype=torch.float32, device=self.device)
        self.var_names, self.var_slices, self.G_var = names, {}, []
        off = 0
        for k in names:
            old = vars_[k]
            n = old.numel()
            self.flat_p[off:off + n].copy_(old.detach().reshape(-1))
            v = self.flat_p[off:off + n].view(old.shape).detach().requires_grad_(True)
            v.grad = self.flat_g[off:off + n].view(old.shape)
            ops._VARS[k] = v
            self.var_slices[k] = (off, n)
            self.G_var.append(v)
            off += n
        self.n_params = total
        self._register_direct_grads()

    _direct_grads_ok = True          # False in trainers whose variables receive more than one gradient per step (GANTrainer)

    def _register_direct_grads(self):
        """The gradient kernels write straight into the flat gradient slab (ops._DIRECT_GRADS) where every variable receives exactly one
        gradient per backward pass: the `de` / `ae` trainers of a single process.  Not under data parallelism (its post-accumulate hooks would
        never fire: enable_data_parallel un-registers) and not in the GAN trainer."""
        ops._DIRECT_GRADS.clear()      # the variable store (ops._VARS) has ONE owner at a time: targets of an earlier trainer are stale pointers
        if self._direct_grads_ok and bool(getattr(self.config, "direct_grads", True)):
            for v in self.G_var:
                ops._DIRECT_GRADS[v.data_ptr()] = v.grad

    def _unregister_direct_grads(self):
        for v in self.G_var:
            ops._DIRECT_GRADS.pop(v.data_ptr(), None)

    def load_variables(self, params):
        """Inject weights by slim name (dict name -> ndarray), e.g. from the oracle's generator_init."""
        for k, val in params.items():
            off, n = self.var_slices[k]
            self.flat_p[off:off + n].copy_(torch.from_numpy(np.asarray(val, np.float32).reshape(-1)))

    def variables_numpy(self):
        return {k: self.flat_p[o:o + n].view(ops._VARS[k].shape).cpu().numpy() for k, (o, n) in self.var_slices.items()}

    def grads_numpy(self):
        return {k: self.flat_g[o:o + n].view(ops._VARS[k].shape).cpu().numpy() for k, (o, n) in self.var_slices.items()}

    def _bucket_id(self, k):
        layer = int(k.split("/")[-2].split("_")[0])
        return 0 if layer == 0 else 1 + (layer - 1) // self.num_conv

    # ---- checkpoint / resume (reference: tf.train.Saver + Supervisor, trainer.py:107-123,291-292) ----
    def _ckpt_slabs(self):
        """[(name -> (offset, numel), params, adam m, adam v)] of every variable slab a checkpoint holds."""
        return [(self.var_slices, self.flat_p, self.flat_m, self.flat_v)]

    def save(self, path):
        """Own format (.npz): variables under their slim names (G/0_fc/weights ...), Adam slots as
        '<name>/Adam' and '<name>/Adam_1' (TF's slot names), 'step', 'g_lr', 'beta_power_t'.  Written to a temporary file and
        renamed, so an interrupted save never leaves a truncated ``model.ckpt-*.npz`` for the next start to restore."""
        out = {}
        for slices, fp, fm, fv in self._ckpt_slabs():
            for k, (o, n) in slices.items():
                shp = ops._VARS[k].shape
                out[k] = fp[o:o + n].view(shp).cpu().numpy()
                out[k + "/Adam"] = fm[o:o + n].view(shp).cpu().numpy()
                out[k + "/Adam_1"] = fv[o:o + n].view(shp).cpu().numpy()
        out["step"] = np.int64(self.step); out["g_lr"] = np.float64(self.g_lr); out["beta_power_t"] = np.int64(self._adam_t)
        out.update(self._ckpt_extra())
        tmp = path + ".tmp.npz"
        np.savez(tmp, **out)
        os.replace(tmp, path if path.endswith(".npz") else path + ".npz")

    def _ckpt_extra(self):
        return {}

    def _ckpt_load_extra(self, d):
        pass

    def load(self, path):
        with np.load(path) as d:
            # everything is checked BEFORE the first copy: a mismatching checkpoint must not leave the slabs half overwritten
            need = [k + sfx for slices, _, _, _ in self._ckpt_slabs() for k in slices for sfx in ("", "/Adam", "/Adam_1")]
            need += ["step", "g_lr", "beta_power_t"] + list(self._ckpt_extra())
            missing = [k for k in need if k not in d.files]
            if missing:      # (tf.train.Saver.restore fails the same way on a graph / checkpoint mismatch)
                raise ValueError("checkpoint %s does not hold this trainer's variables (e.g. %s): a different architecture / scope "
                                 "name was trained in that model_dir" % (path, missing[0]))
            arrs = {}      # every array is read (and decompressed) ONCE: NpzFile is lazy, d[k] loads it each time
            for slices, _, _, _ in self._ckpt_slabs():
                for k, (o, n) in slices.items():
                    for sfx in ("", "/Adam", "/Adam_1"):
                        a = arrs[k + sfx] = d[k + sfx]
                        if int(a.size) != n:
                            raise ValueError("checkpoint %s: variable %s has %d elements (shape %s), this trainer's has %d (shape %s): "
                                             "different filters / resolution / architecture" % (
                                                 path, k + sfx, a.size, tuple(a.shape), n, tuple(ops._VARS[k].shape)))
            for slices, fp, fm, fv in self._ckpt_slabs():
                for k, (o, n) in slices.items():
                    fp[o:o + n].copy_(torch.from_numpy(np.ascontiguousarray(arrs.pop(k), np.float32).reshape(-1)))
                    fm[o:o + n].copy_(torch.from_numpy(np.ascontiguousarray(arrs.pop(k + "/Adam"), np.float32).reshape(-1)))
                    fv[o:o + n].copy_(torch.from_numpy(np.ascontiguousarray(arrs.pop(k + "/Adam_1"), np.float32).reshape(-1)))
            self.step = int(d["step"]); self.g_lr = float(d["g_lr"]); self._adam_t = int(d["beta_power_t"])
            self._ckpt_load_extra(d)

    def effective_model_dir(self):
        """util.py:37-38: ``--load_path`` IS the model directory when given (restore from it AND keep writing into it)."""
        return getattr(self.config, "load_path", "") or getattr(self.config, "model_dir", None)

    def _auto_restore(self):
        """``sv.prepare_or_wait_for_session`` (trainer.py:107-123): a trainer started on a ``model_dir`` (= ``--load_path`` when given,
        util.py:37-38) that already holds checkpoints continues from the latest one -- variables, Adam slots, global step, g_lr."""
        self.restored_from = latest_checkpoint(self.effective_model_dir())
        if self.restored_from is not None:
            self.load(self.restored_from)
            print("[*] restored %s (step %d)" % (self.restored_from, self.step))

    def enable_data_parallel(self, group=None, profile=False, force=False):
        """Bucket the flat gradient slab per generator block (fc | 4 convs | ... | last conv)."""
        groups = {}
        for k in self.var_names:
            gid = (k.rsplit("/", 2)[0], self._bucket_id(k))
            groups.setdefault(gid, []).append(k)
        buckets = []
        for gid in groups:                                   # insertion order == slab order
            ks = groups[gid]
            off = self.var_slices[ks[0]][0]
            n = sum(self.var_slices[k][1] for k in ks)
            buckets.append((off, n, [ops._VARS[k] for k in ks]))
        self.grad_sync = GradSync(self.flat_g, buckets, group, profile=profile, force=force)
        if self.grad_sync.enabled:
            self._unregister_direct_grads()      # the exchange is driven by post-accumulate hooks: gradients must go through AccumulateGrad
        self.sync_state_from_rank0(group)
        return self.grad_sync

    def _state_scalars(self):
        return [self.step, self.g_lr, self._adam_t]

    def _set_state_scalars(self, v):
        self.step, self.g_lr, self._adam_t = int(round(v[0])), float(v[1]), int(round(v[2]))

    def sync_state_from_rank0(self, group=None):
        """Every rank continues from RANK 0's state (parameters, Adam slots, global step, g_lr, Adam step counts): only rank 0 writes
        checkpoints, so a restore-on-start that one rank could not see (no shared file system) must not leave replicas that differ --
        their ``range(step, max_step)`` loops would have different lengths and the gradient exchange would hang."""
        slabs = [t for _, fp, fm, fv in self._ckpt_slabs() for t in (fp, fm, fv)]
        self._set_state_scalars(broadcast_trainer_state(slabs, self._state_scalars(), 0, group))

    # ---- graph (trainer.py:136-172 / trainer3.py:14-51) -------------------------------------------------
    def build_model(self, x, y):
        gen = GeneratorBE3 if self.is_3d else GeneratorBE
        out, _ = gen(y, self.filters, self.output_shape, name=self.name, num_conv=self.num_conv, repeat=self.repeat,
                     reuse=True)
        if self.config.use_curl and self.fused_tail:
            # the whole tail -- curl, both Jacobians (the ground truth's, trainer.py:29-32, recomputed on the fly), both L1 means --
            # as one fused op: 36 B/voxel instead of 240, no 9-channel tensors (velocity_loss.hip); G_jaco_ / G_vort_ are not
            # materialised (the summaries that show them, trainer.py:186-189, are out of scope)
            g_loss_l1, g_loss_j_l1, G_ = ops.velocity_loss(out, x)
            g_loss = g_loss_l1 * self.w1 + g_loss_j_l1 * self.w2    # trainer.py:172
            return SimpleNamespace(G_s=out, G_=G_, G_jaco_=None, G_vort_=None, x_jaco=None, g_loss_l1=g_loss_l1,
                                   g_loss_j_l1=g_loss_j_l1, g_loss=g_loss)
        with torch.no_grad():                                   # trainer.py:29-32: Jacobian of the ground truth
            x_jaco = (jacobian3(x) if self.is_3d else jacobian(x))[0]
        if self.config.use_curl:
            if self.is_3d:
                # `_, self.G_ = jacobian3(self.G_s)` (trainer3.py:18); TF prunes the unused j, eager cannot -> curl3
                G_ = curl3(out) if self.config.use_curl3_alias else jacobian3(out)[1]
            else:
                G_ = curl(out)                                  # trainer.py:140
            G_s = out
        else:
            G_, G_s = out, None
        G_jaco_, G_vort_ = jacobian3(G_) if self.is_3d else jacobian(G_)         # trainer.py:146 / trainer3.py:24
        g_loss_l1 = l1_mean(G_, x)                              # trainer.py:170
        g_loss_j_l1 = l1_mean(G_jaco_, x_jaco)                  # trainer.py:171
        g_loss = g_loss_l1 * self.w1 + g_loss_j_l1 * self.w2    # trainer.py:172
        return SimpleNamespace(G_s=G_s, G_=G_, G_jaco_=G_jaco_, G_vort_=G_vort_, x_jaco=x_jaco,
                               g_loss_l1=g_loss_l1, g_loss_j_l1=g_loss_j_l1, g_loss=g_loss)

    def generate(self, z):
        """Inference graph of build_test_model (trainer.py:295-303), training definition of the 3-D curl."""
        gen = GeneratorBE3 if self.is_3d else GeneratorBE
        with torch.no_grad():
            out, _ = gen(z, self.filters, self.output_shape, name=self.name, num_conv=self.num_conv,
                         repeat=self.repeat, reuse=True)
            if self.config.use_curl:
                out = curl3(out) if self.is_3d else curl(out)
        return out

    # ---- one `sess.run(g_optim)` + `sess.run(g_lr_update)` (trainer.py:269, 284-288) -------------------------
    def forward_backward(self, x, y):
        """Forward + backward + (data parallel) the bucketed gradient exchange, WITHOUT the optimizer step: afterwards ``flat_g`` holds the
        gradient summed over the ranks; returns (graph, the scale 1/world the optimizer folds in)."""
        self.flat_g.zero_()
        if self.grad_sync is not None:
            self.grad_sync.begin_step()
        m = self.build_model(x, y)
        m.g_loss.backward()
        gscale = self.grad_sync.finish() if self.grad_sync is not None else 1.0
        return m, gscale

    def _step_body(self, x, y, dev=None):
        """Everything of one step that runs on the device.  ``dev`` = None: the optimizer's per-step scalars travel by value (eager);
        a device float[4]: the optimizer kernels read them from there (the body is being captured / replayed)."""
        m, gscale = self.forward_backward(x, y)
        self._apply_optimizer(None if dev is not None else self._optimizer_scalars(gscale), dev)
        return m

    def train_step(self, x, y):
        if self.use_graph:
            return self._graph_step(x, y)
        m = self._step_body(x, y)
        self._advance_lr()
        return m

    # ---- the step as a replayed hipGraph ------------------------------------------------------------------------------------------
    def _dp_active(self):
        return self.grad_sync is not None and self.grad_sync.enabled

    def _graph_step(self, x, y):
        """``train_step`` with the device work of the step replayed from a hipGraph captured on the second call with these input shapes
        (the first call runs eagerly: one-time initialisations -- kernel attributes, code-object loading -- must not be recorded).

        Captured: gradient-slab clear, generator / auto-encoder forward, loss tail, the whole autograd backward, the optimizer launch --
        every C-ABI call of the eager step, on the capture stream, with outputs and workspaces from the graph's private pool (so
        every pointer a kernel node holds stays valid).  What changes from step to step lives in device memory the graph reads: the
        inputs (static buffers, refreshed by a device copy) and the optimizer's scalars lr_t / grad_scale (``df_store_scalars`` before
        the launch; kernel arguments by value, so nothing the host may overwrite early).  The returned graph ``m`` is the SAME object
        every step: its tensors are overwritten by the next replay.  Bitwise equal to the eager step (tests/test_gpu_graph.py).

        Data parallel (world > 1): the capture holds forward + backward only, the exchange is ONE eager all-reduce of the flat slab
        after the replay, then the optimizer launch -- no overlap with backward, which is the right trade where a graph matters
        (small steps: the launch overhead saved exceeds the 0.1-0.3 ms exchange)."""
        key = (tuple(x.shape), tuple(y.shape), self._dp_active())
        st = self._graphs.get(key)
        if st is None and key not in self._graph_warm:
            self._graph_warm.add(key)
            m = _detached(self._step_body(x, y))
            self._advance_lr()
            return m
        if st is None:
            st = self._graphs[key] = self._capture(x, y, key[2])
        if x.data_ptr() != st.x.data_ptr():
            st.x.copy_(x)
        if y.data_ptr() != st.y.data_ptr():
            st.y.copy_(y)
        if st.dp:
            st.graph.replay()
            scal = self._optimizer_scalars(self._reduce_all())
            self._apply_optimizer(scal, None)
        else:
            scal = self._optimizer_scalars(1.0)
            call("df_store_scalars", _ptr(st.dev), len(scal), *(list(scal) + [0.0] * (4 - len(scal))), _stream())
            st.graph.replay()
        self._advance_lr()
        return st.m

    def _reduce_all(self):
        return self.grad_sync.reduce_all()

    def _capture(self, x, y, dp):
        st = SimpleNamespace(dp=dp)
        st.x, st.y = x.clone(), y.clone()
        st.dev = torch.zeros(4, dtype=torch.float32, device=self.device)
        st.graph = torch.cuda.CUDAGraph()
        if _lib_mod.TIMER is not None:
            raise RuntimeError("graph capture with a KernelTimer installed: event pairs cannot be recorded into a hipGraph")
        # host-side bookkeeping (dispatch counters, fetch lists) would only see the capture pass, not the replays: off
        with ops.options(dispatch_counts=None, activation_fetch=None, sign_bits_fetch=None):
            if dp:
                self.grad_sync.suspended = True
                sync_d = getattr(self, "grad_sync_d", None)
                if sync_d is not None:
                    sync_d.suspended = True
            try:
                with torch.cuda.graph(st.graph):
                    m = self._capture_body(st.x, st.y, st.dev, dp)
                st.m = _detached(m)
                del m
            finally:
                if dp:
                    self.grad_sync.suspended = False
                    if sync_d is not None:
                        sync_d.suspended = False
        return st

    def _capture_body(self, x, y, dev, dp):
        if not dp:
            return self._step_body(x, y, dev)
        self.flat_g.zero_()
        m = self.build_model(x, y)
        m.g_loss.backward()
        return m

    def _advance_lr(self):
        """``sess.run(g_lr_update)`` after the optimizer step, with the already-incremented global step
        (trainer.py:74-80, 284-288)."""
        self.step += 1
        if self.lr_update == "decay":
            self.g_lr = self.config.lr_min + 0.5 * (self.config.lr_max - self.config.lr_min) * (
                math.cos(self.step * math.pi / self.max_step) + 1.0)
        elif self.lr_update == "step":
            if (self.step - 1) % self.config.lr_update_step == self.config.lr_update_step - 1:
                self.g_lr = max(self.g_lr * 0.5, self.config.lr_min)

    # ---- `train_` (trainer.py:228-293): the loop, the scalar log, the NaN guard, the final checkpoint -------------------------
    def _scalars(self, m, ep):
        """The scalar summaries of trainer.py:190-199 (+ the 'dg' ones, :208-213).  Under data parallelism every loss term is a mean
        over the rank's shard; the logged value is the mean over ranks = the global-batch mean the single-process reference logs
        (equal shard sizes), so every rank sees the SAME numbers and the NaN guard fires on all ranks together or on none."""
        keys = ["g_loss", "g_loss_l1", "g_loss_j_l1"] + [k for k in ("g_loss_real", "d_loss_real", "d_loss_fake")
                                                        if getattr(m, k, None) is not None]
        vals = torch.stack([getattr(m, k).detach().reshape(()).float() for k in keys])
        gs = self.grad_sync
        if gs is not None and gs.enabled and gs.world > 1:
            import torch.distributed as dist
            if gs.backend == "gloo" and vals.is_cuda:
                h = vals.cpu()
                dist.all_reduce(h, op=dist.ReduceOp.SUM, group=gs.group)
                vals = h / gs.world
            else:
                dist.all_reduce(vals, op=dist.ReduceOp.SUM, group=gs.group)
                vals = vals / gs.world
        v = dict(zip(keys, vals.tolist()))
        out = {"loss/g_loss": v["g_loss"], "loss/g_loss_l1": v["g_loss_l1"], "loss/g_loss_j_l1": v["g_loss_j_l1"], "misc/epoch": ep,
               "misc/g_lr": self.g_lr}
        for k in keys[3:]:
            out["loss/" + k] = math.sqrt(max(v[k], 0.0))
        return out

    def train(self, batch_manager, max_step=None, model_dir=None, log_step=None, test_step=None, on_log=None):
        """``Trainer.train_``: steps ``start_step .. max_step-1`` (from the restored global step when the trainer was started on a
        ``model_dir`` with checkpoints) on batches dequeued from ``batch_manager``; a checkpoint ``model.ckpt-<global step>.npz`` every
        ``config.save_sec`` seconds of wall time (config.py:68; the Supervisor's timed saver).  Every ``log_step``
        steps (and at the last) the scalars of the reference's summary op are appended as one JSON line to
        ``<model_dir>/scalars.jsonl`` (the counterpart of the TensorBoard event file) and the loss is checked for NaN
        (trainer.py:271-276); every ``test_step`` steps the fixed parameter sweeps are generated (trainer.py:230-241, 281-282;
        stored as ``<model_dir>/<step>_G.npz`` instead of PNG sheets); the last checkpoint is written at the end
        (trainer.py:290-292).  Only rank 0 of a data-parallel job writes files.  Returns the logged records."""
        model_dir = model_dir or self.effective_model_dir()
        log_step = log_step or self.config.log_step
        test_step = test_step or self.config.test_step
        max_step = self.max_step if max_step is None else max_step
        rank0 = self.grad_sync is None or not self.grad_sync.enabled or self.grad_sync.rank == 0
        if model_dir and rank0:
            os.makedirs(model_dir, exist_ok=True)
            with open(os.path.join(model_dir, "params.json"), "w") as fp:        # util.py:52-59 save_config
                json.dump({k: v for k, v in sorted(vars(self.config).items())}, fp, indent=4, sort_keys=True, default=str)
        # test1: each parameter varied over [-1, 1] with the others at 0 (trainer.py:230-241)
        z_samples = []
        for i in range(self.c_num):
            zi = np.zeros((self.b_num, self.c_num), np.float32)
            zi[:, i] = np.linspace(-1, 1, num=self.b_num)
            z_samples.append(zi)
        records = []
        t0 = time.time()
        last_save = time.time()
        for step in range(self.step, max_step):
            x, y = batch_manager.batch()
            m = self.train_step(x, y)
            if model_dir and rank0 and self.save_sec is not None and time.time() - last_save >= self.save_sec:
                # the Supervisor's timed saver (save_model_secs = config.save_sec, trainer.py:110-117): model.ckpt-<global step>
                self.save(os.path.join(model_dir, "model.ckpt-%d.npz" % self.step))
                last_save = time.time()
            if step % log_step == 0 or step == max_step - 1:
                ep = step * batch_manager.epochs_per_step
                rec = self._scalars(m, ep)
                loss = rec["loss/g_loss"]
                assert not np.isnan(loss), "Model diverged with loss = NaN"            # trainer.py:275
                rec.update(step=step, wall_s=time.time() - t0)
                records.append(rec)
                if rank0:
                    print("[{}/{}/ep{:.2f}] Loss: {:.6f}".format(step, max_step, ep, loss))
                    if model_dir:
                        with open(os.path.join(model_dir, "scalars.jsonl"), "a") as f:
                            f.write(json.dumps(rec) + "\n")
                if on_log is not None:
                    on_log(rec)
            if model_dir and rank0 and (step % test_step == 0 or step == max_step - 1):
                G = np.stack([self.generate(torch.from_numpy(z).to(self.device)).cpu().numpy() for z in z_samples])
                np.savez_compressed(os.path.join(model_dir, "%d_G.npz" % step), G=G, z=np.stack(z_samples))
        if model_dir and rank0:
            self.save(os.path.join(model_dir, "model.ckpt-%d.npz" % self.step))
        if hasattr(batch_manager, "stop_thread"):
            batch_manager.stop_thread()
        return records

    # ---- `test_` (trainer.py:314-354): one parameter pair, every frame, de-normalised, one .npz per frame ----------------------
    def test_(self, batch_manager, model_dir=None, p1=10, p2=2, test_b_num=None):
        """The inference sweep of ``Trainer.test_``: fix the first two control parameters at grid indices (p1, p2) -> c = p/(num-1)*2-1,
        sweep the last one (the frame number) over its ``y_num[2]`` values in [-1, 1], run the inference graph in batches of
        ``test_batch_size``, de-normalise with the dataset's velocity range (``batch_manager.denorm``) and dump frame i to
        ``<model_dir>/<p1>_<p2>/<i>.npz`` under key ``x`` (np.savez_compressed) -- the files the reference's visualisation scripts read."""
        model_dir = model_dir or self.effective_model_dir()
        test_b_num = test_b_num or self.config.test_batch_size
        y1, y2, y3 = (int(v) for v in batch_manager.y_num[:3])
        if y3 % test_b_num != 0:                                     # trainer.py:324 asserts; the default 100 rarely divides test data
            raise ValueError("test_: the number of frames (%d) must be a multiple of test_batch_size (%d)" % (y3, test_b_num))
        niter = y3 // test_b_num
        c1 = p1 / float(y1 - 1) * 2 - 1
        c2 = p2 / float(y2 - 1) * 2 - 1
        z_c = np.zeros((y3, self.c_num), np.float32)
        z_c[:, 0] = c1
        z_c[:, 1] = c2
        z_c[:, -1] = np.linspace(-1, 1, num=y3)
        G = []
        for b in range(niter):
            z = torch.from_numpy(z_c[test_b_num * b:test_b_num * (b + 1)]).to(self.device)
            G_ = self.generate(z).cpu().numpy()
            G_, _ = batch_manager.denorm(x=G_)
            G.append(G_)
        G = np.concatenate(G, axis=0)
        out_dir = os.path.join(model_dir, "%d_%d" % (p1, p2))
        os.makedirs(out_dir, exist_ok=True)
        for i, G_ in enumerate(G):
            np.savez_compressed(os.path.join(out_dir, "%d.npz" % i), x=G_)
        return out_dir

    def _optimizer_scalars(self, grad_scale):
        """The optimizer's per-step host scalars ``[lr_t | lr, grad_scale]``; advances Adam's step count (beta powers)."""
        if self.config.optimizer == "gd":                       # trainer.py:163-165
            return [float(self.g_lr), float(grad_scale)]
        self._adam_t += 1
        t = self._adam_t
        return [float(self.g_lr * math.sqrt(1.0 - self.beta2 ** t) / (1.0 - self.beta1 ** t)), float(grad_scale)]

    def _apply_slab(self, p, g, m, v, n, lr, gscale, dev, dev_off=0):
        """One optimizer launch over a flat slab; scalars by value (``dev`` None) or the pair ``dev[dev_off], dev[dev_off + 1]``."""
        sp = None if dev is None else dev.data_ptr() + 4 * dev_off
        if self.config.optimizer == "gd":
            if dev is None:
                call("df_gd_step", _ptr(p), _ptr(g), n, lr, gscale, _stream())
            else:
                call("df_gd_step_dev", _ptr(p), _ptr(g), n, sp, _stream())
        elif dev is None:
            call("df_adam_tf1_step", _ptr(p), _ptr(g), _ptr(m), _ptr(v), n, lr, float(self.beta1), float(self.beta2), float(self.eps),
                 gscale, _stream())
        else:
            call("df_adam_tf1_step_dev", _ptr(p), _ptr(g), _ptr(m), _ptr(v), n, sp, float(self.beta1), float(self.beta2),
                 float(self.eps), _stream())

    def _apply_optimizer(self, scal, dev):
        lr, gscale = scal if scal is not None else (None, None)
        self._apply_slab(self.flat_p, self.flat_g, self.flat_m, self.flat_v, self.n_params, lr, gscale, dev)

    def _apply_adam(self, grad_scale):
        """(kept for callers of the pre-graph API) one eager optimizer step with the gradients scaled by ``grad_scale``."""
        self._apply_optimizer(self._optimizer_scalars(grad_scale), None)


class Trainer3(Trainer):
    """trainer3.py: the 3-D overrides are selected by ``config.is_3d``; kept as a name for call-site parity."""

    def __init__(self, config, device="cuda", name="G"):
        config.is_3d = True
        super(Trainer3, self).__init__(config, device, name)


class AETrainer(Trainer):
    """arch='ae' (SURVEY 8(f)-1): ``build_model_ae`` + the optimizer step of ``train_ae``
    (trainer.py:357-462; trainer3.py:240-345).  x -> AE|AE3 -> (curl) -> L1 + Jacobian-L1 + w4*loss_p."""

    def __init__(self, config, device="cuda", name="AE"):
        self.z_num = config.z_num
        self.p_num = config.p_num
        self.use_sparse = config.use_sparse
        self.sparsity = config.sparsity
        self.w5 = config.w5
        self.w4 = config.w4
        super(AETrainer, self).__init__(config, device, name)

    def _x_shape(self):
        spatial = ([self.config.res_z] if self.is_3d else []) + [self.config.res_y, self.config.res_x]
        ch = self.config.x_channels or (3 if self.is_3d else 2)
        return spatial + [ch]

    def _create_variables(self):
        ae = AE3 if self.is_3d else AE
        x = torch.zeros([1] + self._x_shape(), dtype=torch.float32, device=self.device)
        with torch.no_grad():      # one shape-only pass creates AE/enc/* and AE/dec/* (decoder output = x's own shape)
            ae(x, self.filters, self.z_num, name=self.name, num_conv=self.num_conv, repeat=self.repeat)

    def build_model(self, x, y):
        ae = AE3 if self.is_3d else AE
        with torch.no_grad():
            x_jaco = (jacobian3(x) if self.is_3d else jacobian(x))[0]
        out, z, _ = ae(x, self.filters, self.z_num, name=self.name, num_conv=self.num_conv, repeat=self.repeat,
                       use_sparse=self.use_sparse, reuse=True)
        if self.config.use_curl:
            x_ = (curl3(out) if self.config.use_curl3_alias else jacobian3(out)[1]) if self.is_3d else curl(out)
        else:
            x_ = out
        x_jaco_, x_vort_ = jacobian3(x_) if self.is_3d else jacobian(x_)
        loss_l1 = l1_mean(x_, x)                                   # trainer3.py:265
        loss_j_l1 = l1_mean(x_jaco_, x_jaco)                       # trainer3.py:266
        y_last = y[:, :, -1] if y.dim() == 3 else y                # trainer3.py:268
        loss_p = mse_mean(y_last.contiguous(), z[:, -self.p_num:].contiguous())   # trainer3.py:269-270
        loss = loss_l1 * self.w1 + loss_j_l1 * self.w2 + loss_p * self.w4
        loss_kl = None
        if self.use_sparse:                                        # trainer3.py:272-277 (z is sigmoid(enc) then, model.py:210)
            loss_kl = kl_bernoulli(z, self.z_num - self.p_num, self.sparsity)
            loss = loss + loss_kl * self.w5
        return SimpleNamespace(s=out, G_=x_, x_=x_, z=z, G_jaco_=x_jaco_, G_vort_=x_vort_, x_jaco=x_jaco,
                               g_loss_l1=loss_l1, g_loss_j_l1=loss_j_l1, loss_p=loss_p, loss_kl=loss_kl, g_loss=loss, loss=loss)


    # ---- `test_ae` (trainer.py:475-583, trainer3.py:311-367; `--arch=ae --is_train=False`) ------------------------------------
    def encode(self, x):
        """The code ``z`` of the inference graph (``build_test_model_ae``, trainer.py:464-473 / trainer3.py:311-320)."""
        ae = AE3 if self.is_3d else AE
        with torch.no_grad():
            _, z, _ = ae(x, self.filters, self.z_num, name=self.name, num_conv=self.num_conv, repeat=self.repeat,
                         use_sparse=self.use_sparse, reuse=True)
        return z

    def decode(self, z):
        """``sess.run(self.x_, {self.z: z})`` (trainer.py:541-542): feeding the code tensor bypasses the encoder AND the sigmoid of
        ``use_sparse``; the decoder 'dec' (+ curl when ``use_curl``) turns the fed code into the velocity field."""
        gen = GeneratorBE3 if self.is_3d else GeneratorBE
        with torch.no_grad():
            out, _ = gen(z, self.filters, self._x_shape(), name=self.name + "/dec", num_conv=self.num_conv, repeat=self.repeat, reuse=True)
            if self.config.use_curl:
                out = curl3(out) if self.is_3d else curl(out)
        return out

    def test_(self, batch_manager, model_dir=None, test_b_num=None, code_path=None):
        """Positional arguments as ``Trainer.test_`` up to ``model_dir``; the batch size and ``code_path`` by keyword (the base class has
        ``p1, p2`` in between, so a positional batch size could not mean the same thing on both)."""
        return self.test_ae(batch_manager, model_dir, code_path=code_path, test_b_num=test_b_num)

    def test_ae(self, batch_manager, model_dir=None, *, code_path=None, test_b_num=None):
        """``Trainer.test_ae``.  Without ``code_path`` (trainer.py:478-523): encode the whole dataset in file order
        (``batch_manager.batch_``) and write ``<model_dir>/code<z_num>.npz`` -- ``x`` = codes of frames 0..F-2 of every scene, ``y`` =
        codes of frames 1..F-1, ``p`` = per-frame source-position increments from ``<data root>/n.npz`` (nx [, nz]), ``s`` scenes,
        ``f`` frames: the training set of the latent-space integrator.  With ``code_path`` (trainer.py:524-571): read
        ``<code_path>/code_out.npz`` (``z_out``, ``z_gt`` [scenes, frames, z_num]), decode both in batches of ``test_batch_size``,
        de-normalise and write ``<model_dir>/v<scene>.npz`` with ``v`` / ``v_gt`` (the arrays the reference builds and whose save it
        leaves commented out, trainer.py:569-571; its PNG sheets are visualisation, out of scope).  Returns the written path(s)."""
        model_dir = model_dir or self.effective_model_dir()
        test_b_num = test_b_num or self.config.test_batch_size
        code_path = code_path if code_path is not None else getattr(self.config, "code_path", "")
        os.makedirs(model_dir, exist_ok=True)
        if not code_path:
            with np.load(os.path.join(batch_manager.root, "n.npz")) as data:
                nx = data["nx"]
                nz = data["nz"] if self.is_3d else None
            num_sims, num_frames = nx.shape[0], nx.shape[1]
            dx_list = (nx[:, 1:] - nx[:, :-1]).reshape([-1, 1])
            if self.is_3d:
                dz_list = (nz[:, 1:] - nz[:, :-1]).reshape([-1, 1])
                p_list = np.concatenate((dx_list, dz_list), axis=-1)
            else:
                p_list = dx_list
            c_list = []
            for x, _ in batch_manager.batch_(test_b_num):
                c_list.append(self.encode(torch.from_numpy(np.ascontiguousarray(x, np.float32)).to(self.device)).cpu().numpy())
            c_list = np.concatenate(c_list)
            x_list, y_list = [], []
            for i in range(num_sims):
                s1, s2 = i * num_frames, (i + 1) * num_frames
                x_list.append(c_list[s1:s2 - 1, :])
                y_list.append(c_list[s1 + 1:s2, :])
            x_list, y_list = np.concatenate(x_list), np.concatenate(y_list)
            out = os.path.join(model_dir, "code%d.npz" % self.z_num)
            np.savez_compressed(out, x=x_list, y=y_list, p=p_list, s=num_sims, f=num_frames)
            return out
        with np.load(os.path.join(code_path, "code_out.npz")) as data:
            z_, z_gt_ = data["z_out"], data["z_gt"]
        num_sims, num_frames = z_.shape[0], z_[0].shape[0]
        # (the reference runs int(num_frames / test_b_num) full batches and silently drops the rest, trainer.py:531-533 -- with fewer frames
        #  than test_batch_size it decodes nothing; here every frame is decoded: the last batch is the remainder)
        if num_frames <= 0:
            raise ValueError("test_ae: %s holds no frames" % os.path.join(code_path, "code_out.npz"))
        test_b_num = max(1, min(int(test_b_num), num_frames))
        num_iters = -(-num_frames // test_b_num)
        paths = []
        for s in range(num_sims):
            v, v_gt = [], []
            for i in range(num_iters):
                for src, dst in ((z_[s], v), (z_gt_[s], v_gt)):
                    zz = torch.from_numpy(np.ascontiguousarray(src[i * test_b_num:min((i + 1) * test_b_num, num_frames), :], np.float32)).to(self.device)
                    vv, _ = batch_manager.denorm(self.decode(zz).cpu().numpy())
                    dst.append(vv)
            out = os.path.join(model_dir, "v%d.npz" % s)
            np.savez_compressed(out, v=np.concatenate(v, axis=0), v_gt=np.concatenate(v_gt, axis=0))
            paths.append(out)
        return paths


class _Slab(object):
    """Flat parameter / gradient / Adam-slot slabs over a set of registry variables (views re-registered in place)."""

    def __init__(self, names, device):
        vars_ = ops.all_variables()
        total = sum(vars_[k].numel() for k in names)
        self.p = torch.empty(total, dtype=torch.float32, device=device)
        self.g = torch.zeros(total, dtype=torch.float32, device=device)
        self.m = torch.zeros(total, dtype=torch.float32, device=device)
        self.v = torch.zeros(total, dtype=torch.float32, device=device)
        self.names, self.slices, self.vars = list(names), {}, []
        off = 0
        for k in names:
            old = vars_[k]
            n = old.numel()
            self.p[off:off + n].copy_(old.detach().reshape(-1))
            v = self.p[off:off + n].view(old.shape).detach().requires_grad_(True)
            v.grad = self.g[off:off + n].view(old.shape)
            ops._VARS[k] = v
            self.slices[k] = (off, n)
            self.vars.append(v)
            off += n
        self.n = total


class GANTrainer(Trainer):
    """arch='dg' (SURVEY 8(f)-4): generator + PatchGAN discriminator with LSGAN terms, both updated from the same
    forward pass like ``sess.run([g_optim, d_optim])`` (trainer.py:149-156, 174-184, 265-267; trainer3.py:26-33,53-63)."""

    _restore_in_base = False
    _direct_grads_ok = False      # G's backward runs inside a graph that also reaches D's variables (applied twice): classic accumulation

    def __init__(self, config, device="cuda", name="G"):
        self.w3 = config.w3
        super(GANTrainer, self).__init__(config, device, name)
        self.fused_tail = False                                   # the discriminator reads the vorticity of G_ (trainer.py:155-156)
        disc = DiscriminatorPatch3 if self.is_3d else DiscriminatorPatch
        cin = 6 if self.is_3d else 3                              # concat([x, x_vort]): 3+3 | 2+1 channels
        spatial = ([config.res_z] if self.is_3d else []) + [config.res_y, config.res_x]
        if not any(k.startswith("D/") for k in ops.all_variables()):
            with torch.no_grad():
                disc(torch.zeros([1] + spatial + [cin], device=self.device), self.filters)
        self.D = _Slab([k for k in ops.all_variables() if k.startswith("D/")], self.device)
        self._adam_t_d = 0
        self.grad_sync_d = None
        self._auto_restore()

    def _ckpt_slabs(self):
        return super(GANTrainer, self)._ckpt_slabs() + [(self.D.slices, self.D.p, self.D.m, self.D.v)]

    def _ckpt_extra(self):
        return {"beta_power_t_d": np.int64(self._adam_t_d)}

    def _ckpt_load_extra(self, d):
        self._adam_t_d = int(d["beta_power_t_d"])

    def _state_scalars(self):
        return super(GANTrainer, self)._state_scalars() + [self._adam_t_d]

    def _set_state_scalars(self, v):
        super(GANTrainer, self)._set_state_scalars(v)
        self._adam_t_d = int(round(v[3]))

    def enable_data_parallel(self, group=None, profile=False, force=False):
        """Two gradient slabs -> two bucketed exchanges: G's per generator block (as in ``Trainer``), D's as one bucket."""
        gs = super(GANTrainer, self).enable_data_parallel(group, profile, force)
        self.grad_sync_d = GradSync(self.D.g, [(0, self.D.n, list(self.D.vars))], group, profile=profile, force=force)
        return gs

    def _optimizer_scalars(self, grad_scale):
        """[G's lr_t | lr, grad_scale, D's lr_t | lr, grad_scale]: two optimizers with their own beta powers (trainer.py:160-165,183-184)."""
        out = super(GANTrainer, self)._optimizer_scalars(grad_scale)
        if self.config.optimizer == "gd":
            return out + [float(self.g_lr), float(grad_scale)]
        self._adam_t_d += 1
        t = self._adam_t_d
        return out + [float(self.g_lr * math.sqrt(1.0 - self.beta2 ** t) / (1.0 - self.beta1 ** t)), float(grad_scale)]

    def _apply_optimizer(self, scal, dev):
        lg, sg, ld, sd = scal if scal is not None else (None,) * 4
        self._apply_slab(self.flat_p, self.flat_g, self.flat_m, self.flat_v, self.n_params, lg, sg, dev, 0)
        self._apply_slab(self.D.p, self.D.g, self.D.m, self.D.v, self.D.n, ld, sd, dev, 2)

    def _forward_backward_gan(self, x, y, dp):
        disc = DiscriminatorPatch3 if self.is_3d else DiscriminatorPatch
        self.flat_g.zero_(); self.D.g.zero_()
        if dp:
            self.grad_sync.begin_step(); self.grad_sync_d.begin_step()
        m = self.build_model(x, y)
        with torch.no_grad():
            x_vort = (jacobian3(x) if self.is_3d else jacobian(x))[1]                 # trainer.py:30,32
        D_x, _ = disc(concat([x, x_vort], axis=-1), self.filters, reuse=True)         # trainer.py:153-154
        D_G, _ = disc(concat([m.G_, m.G_vort_], axis=-1), self.filters, reuse=True)   # trainer.py:155-156
        ones_g, zeros_g, ones_x = torch.ones_like(D_G), torch.zeros_like(D_G), torch.ones_like(D_x)
        m.g_loss_real = ops.mse_mean(D_G, ones_g)                                      # trainer.py:175
        m.d_loss_fake = ops.mse_mean(D_G, zeros_g)                                     # trainer.py:176
        m.d_loss_real = ops.mse_mean(D_x, ones_x)                                      # trainer.py:177
        m.g_loss = m.g_loss + m.g_loss_real * self.w3                                  # trainer.py:179
        m.d_loss = m.d_loss_real + m.d_loss_fake                                       # trainer.py:181
        # two `minimize` calls on disjoint var_lists (trainer.py:183-184): each backward accumulates into its own slab only, so
        # the post-accumulate hooks of a slab fire during exactly one of the two passes and its buckets are reduced once
        m.g_loss.backward(inputs=self.G_var, retain_graph=True)
        m.d_loss.backward(inputs=self.D.vars)
        return m

    def _step_body(self, x, y, dev=None):
        dp = self.grad_sync is not None
        m = self._forward_backward_gan(x, y, dp)
        gscale = 1.0
        if dp:
            gscale = self.grad_sync.finish()
            self.grad_sync_d.finish()
        self._apply_optimizer(None if dev is not None else self._optimizer_scalars(gscale), dev)   # trainer.py:163-165 (both optimizers)
        return m

    def _capture_body(self, x, y, dev, dp):
        if not dp:
            return self._step_body(x, y, dev)
        return self._forward_backward_gan(x, y, False)

    def _reduce_all(self):
        s = self.grad_sync.reduce_all()
        self.grad_sync_d.reduce_all()
        return s

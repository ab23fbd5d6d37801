"""Dataset reader for the reference's on-disk format: counterpart of ``data.py::BatchManager``.

On-disk format (written by the reference's ``scene/*.py``, e.g. scene/smoke_pos_size.py:121-126,215-218,230-234):
  <root>/args.txt            "key: value" lines (num_param, p0.., min_/max_/num_<pname>, num_frames, num_dof, path_format ...)
  <root>/v/%d_%d_%d.npz      x: [Y,X,2] | [Z,Y,X,3] float32 velocity, y: [c_num] parameters  (AE sets: y [dof, frames])
  <root>/v_range.txt         two numbers; x is normalised by max(|r0|, |r1|)     (data.py:87-88, 329)
Labels are mapped to [-1,1] with min_/max_<pname> (data.py:331-332).

The reference feeds a tf.FIFOQueue from N Python threads that share one RandomState (data.py:116-144) and dequeues
``batch_size`` samples in-graph.  Here N worker threads fill PINNED host batches and a double-buffered queue hands
them to the GPU with an async copy on a side stream, so the H2D transfer (75 MB per 16 samples at cfg3 = 1.2 ms at
63 GB/s) overlaps the ~0.6 s train step.  Sampling is i.i.d. with replacement per sample, as in the reference.
"""
import os
import queue
import threading
from glob import glob

import numpy as np
import torch


def preprocess(file_path, data_type, x_range, y_range):
    """data.py:311-333."""
    with np.load(file_path) as data:
        x = data["x"].astype(np.float32)
        y = data["y"].astype(np.float32)
    if data_type[0] == "d":
        x = x * 2 - 1
    else:
        x = x / x_range
    for i, ri in enumerate(y_range):
        y[i] = (y[i] - ri[0]) / (ri[1] - ri[0]) * 2 - 1
    return x, y


class BatchManager(object):
    def __init__(self, config, device="cuda", prefetch=2):
        self.rng = np.random.RandomState(config.random_seed)        # data.py:18
        self.root = config.data_path
        self.args = {}
        with open(os.path.join(self.root, "args.txt"), "r") as f:   # data.py:22-29
            for line in f:
                if not line.strip():
                    continue
                arg, arg_value = line.rstrip("\n").split(": ")
                self.args[arg] = arg_value
        self.is_3d = config.is_3d
        self.data_type = config.data_type
        pattern = "{}/{}/*".format(self.root, config.data_type[0])
        if "ae" in config.arch:                                      # data.py:32-38: sort by (sim, frame)
            nf = int(self.args["num_frames"])

            def sortf(p):
                n = os.path.basename(p)[:-4].split("_")
                return int(n[0]) * nf + int(n[1])
            self.paths = sorted(glob(pattern), key=sortf)
        else:
            self.paths = sorted(glob(pattern))
        self.num_samples = len(self.paths)
        assert self.num_samples > 0                                  # data.py:48
        self.batch_size = config.batch_size
        self.epochs_per_step = self.batch_size / float(self.num_samples)
        self.depth = (3 if self.is_3d else 2) if self.data_type == "velocity" else 1
        self.res_x, self.res_y, self.res_z = config.res_x, config.res_y, config.res_z
        self.c_num = int(self.args["num_param"])
        self.feature_dim = ([self.res_z] if self.is_3d else []) + [self.res_y, self.res_x, self.depth]
        if "ae" in config.arch:
            self.dof = int(self.args["num_dof"])
            self.label_dim = [self.dof, int(self.args["num_frames"])]
        else:
            self.label_dim = [self.c_num]
        r = np.loadtxt(os.path.join(self.root, self.data_type[0] + "_range.txt"))
        self.x_range = max(abs(r[0]), abs(r[1]))                     # data.py:87-88
        self.y_range, self.y_num = [], []
        for i in range(self.c_num):                                  # data.py:92-108
            p_name = self.args["p%d" % i]
            self.y_num.append(int(self.args["num_{}".format(p_name)]))
            if "ae" not in config.arch:
                self.y_range.append([float(self.args["min_{}".format(p_name)]), float(self.args["max_{}".format(p_name)])])
        if "ae" in config.arch:
            self.y_range = [[-1, 1] for _ in range(self.label_dim[0])]
        self.num_threads = int(np.amin([getattr(config, "num_worker", 2), os.cpu_count() or 1, self.batch_size]))
        self.device = torch.device(device) if device is not None else None
        self._q = queue.Queue(maxsize=prefetch)
        self._stop = threading.Event()
        self._threads = []
        self._lock = threading.Lock()
        self._copy_stream = None

    # ---- producer side -------------------------------------------------------------------------------------
    def _make_batch(self):
        pin = self.device is not None and self.device.type == "cuda"
        xb = torch.empty([self.batch_size] + self.feature_dim, dtype=torch.float32, pin_memory=pin)
        yb = torch.empty([self.batch_size] + self.label_dim, dtype=torch.float32, pin_memory=pin)
        for i in range(self.batch_size):
            with self._lock:                                          # the reference shares one RandomState unlocked (benign race)
                idx = self.rng.randint(len(self.paths))
            x_, y_ = preprocess(self.paths[idx], self.data_type, self.x_range, self.y_range)
            xb[i].copy_(torch.from_numpy(np.ascontiguousarray(x_)))
            yb[i].copy_(torch.from_numpy(np.ascontiguousarray(y_)))
        return xb, yb

    def _worker(self):
        while not self._stop.is_set():
            item = self._make_batch()
            while not self._stop.is_set():
                try:
                    self._q.put(item, timeout=0.1)
                    break
                except queue.Full:
                    continue

    def start_thread(self, sess=None):
        """data.py:116-159 (``sess`` is accepted for call-site parity and ignored)."""
        if self._threads:
            return
        self._stop.clear()
        self._threads = [threading.Thread(target=self._worker, daemon=True) for _ in range(self.num_threads)]
        for t in self._threads:
            t.start()

    def stop_thread(self):
        self._stop.set()
        for t in self._threads:
            t.join(timeout=5)
        self._threads = []

    def __del__(self):
        try:
            self.stop_thread()
        except Exception:
            pass

    # ---- consumer side -------------------------------------------------------------------------------------
    def batch(self):
        """One normalised batch (x [B,(Z,)Y,X,C], y [B,c_num] | [B,dof,frames]) on the device (data.py:170-171)."""
        if not self._threads:
            self.start_thread()
        xb, yb = self._q.get()
        if self.device is None or self.device.type != "cuda":
            return xb, yb
        if self._copy_stream is None:
            self._copy_stream = torch.cuda.Stream()
        with torch.cuda.stream(self._copy_stream):
            xd = xb.to(self.device, non_blocking=True)
            yd = yb.to(self.device, non_blocking=True)
        torch.cuda.current_stream().wait_stream(self._copy_stream)
        xd.record_stream(torch.cuda.current_stream()); yd.record_stream(torch.cuda.current_stream())
        return xd, yd

    def batch_(self, b_num):
        """Sequential pass over the whole set (data.py:173-184)."""
        assert len(self.paths) % b_num == 0
        x_batch = []
        for i, filepath in enumerate(self.paths):
            x, _ = preprocess(filepath, self.data_type, self.x_range, self.y_range)
            x_batch.append(x)
            if (i + 1) % b_num == 0:
                yield np.array(x_batch), []
                x_batch = []

    def denorm(self, x=None, y=None):
        """[-1,1] -> original range (data.py:186-195)."""
        if x is not None:
            x *= self.x_range
        if y is not None:
            for i, ri in enumerate(self.y_range):
                y[:, i] = (y[:, i] + 1) * 0.5 * (ri[1] - ri[0]) + ri[0]
        return x, y


def write_synthetic_dataset(root, spatial, num_p=(3, 2), num_frames=4, seed=0, ae=False):
    """Write a tiny dataset in the reference's on-disk format (tests / demos; there is no network for real data)."""
    rng = np.random.RandomState(seed)
    os.makedirs(os.path.join(root, "v"), exist_ok=True)
    is_3d = len(spatial) == 3
    lines = ["num_param: 3", "p0: src_x_pos", "p1: src_radius", "p2: frames",
             "min_src_x_pos: 0.2", "max_src_x_pos: 0.8", "num_src_x_pos: %d" % num_p[0],
             "min_src_radius: 0.04", "max_src_radius: 0.12", "num_src_radius: %d" % num_p[1],
             "min_frames: 0", "max_frames: %d" % (num_frames - 1), "num_frames: %d" % num_frames,
             "num_dof: 2", "path_format: %d_%d_%d.npz"]
    with open(os.path.join(root, "args.txt"), "w") as f:
        f.write("\n".join(lines) + "\n")
    vmax = 0.0
    for i in range(num_p[0]):
        for j in range(num_p[1]):
            for t in range(num_frames):
                x = rng.uniform(-2, 2, size=list(spatial) + [3 if is_3d else 2]).astype(np.float32)
                vmax = max(vmax, float(np.abs(x).max()))
                px = 0.2 + 0.6 * i / max(num_p[0] - 1, 1); pr = 0.04 + 0.08 * j / max(num_p[1] - 1, 1)
                y = rng.uniform(-1, 1, size=(2, num_frames)).astype(np.float32) if ae else np.array([px, pr, t], np.float32)
                np.savez_compressed(os.path.join(root, "v", "%d_%d_%d.npz" % (i, j, t)), x=x, y=y)
    with open(os.path.join(root, "v_range.txt"), "w") as f:
        f.write("%.3f\n%.3f\n" % (-vmax, vmax))
    return num_p[0] * num_p[1] * num_frames


def write_synthetic_ae_dataset(root, spatial, num_scenes=2, num_frames=4, seed=0):
    """A tiny dataset in the on-disk format of the reference's moving-source scenes (scene/smoke3_mov.py:18-38,176-182,286-326 --
    what ``--arch=ae`` trains on, run.bat:56,73): ``v/<scene>_<frame>.npz`` with x = velocity [(Z,)Y,X,2|3] and y = source positions
    [dof, frames]; ``n.npz`` with the noise tracks ``nx`` (and ``nz`` in 3-D) [scenes, frames]; ``args.txt``; ``v_range.txt``."""
    rng = np.random.RandomState(seed)
    os.makedirs(os.path.join(root, "v"), exist_ok=True)
    is_3d = len(spatial) == 3
    dof = 2 if is_3d else 1
    lines = ["num_param: 2", "path_format: %d_%d.npz", "p0: scenes", "p1: frames", "min_scenes: 0", "max_scenes: %d" % (num_scenes - 1),
             "num_scenes: %d" % num_scenes, "min_frames: 0", "max_frames: %d" % (num_frames - 1), "num_frames: %d" % num_frames,
             "num_simulations: %d" % (num_scenes * num_frames), "num_dof: %d" % dof]
    with open(os.path.join(root, "args.txt"), "w") as f:
        f.write("\n".join(lines) + "\n")
    nx = rng.uniform(-1, 1, (num_scenes, num_frames)).astype(np.float32)
    nz = rng.uniform(-1, 1, (num_scenes, num_frames)).astype(np.float32)
    vmax = 0.0
    for i in range(num_scenes):
        for t in range(num_frames):
            x = rng.uniform(-2, 2, size=list(spatial) + [3 if is_3d else 2]).astype(np.float32)
            vmax = max(vmax, float(np.abs(x).max()))
            y = np.stack([nx[i], nz[i]])[:dof].astype(np.float32)           # [dof, frames]; the trainer reads y[:, :, -1]
            np.savez_compressed(os.path.join(root, "v", "%d_%d.npz" % (i, t)), x=x, y=y)
    if is_3d:
        np.savez_compressed(os.path.join(root, "n.npz"), nx=nx, nz=nz)
    else:
        np.savez_compressed(os.path.join(root, "n.npz"), nx=nx)
    with open(os.path.join(root, "v_range.txt"), "w") as f:
        f.write("%.3f\n%.3f\n" % (-vmax, vmax))
    return num_scenes * num_frames

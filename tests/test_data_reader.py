"""CPU: the dataset reader (deep_fluids_amd/data.py) on a tiny dataset written in the reference's on-disk format."""
import os
from types import SimpleNamespace

import numpy as np

from deep_fluids_amd.data import BatchManager, preprocess, write_synthetic_dataset


def test_batch_manager_reads_reference_format(tmp_path):
    root = str(tmp_path / "smoke_tiny")
    n = write_synthetic_dataset(root, (8, 6), num_p=(3, 2), num_frames=4)
    cfg = SimpleNamespace(random_seed=123, data_path=root, is_3d=False, arch="de", data_type="velocity", batch_size=4,
                          res_x=6, res_y=8, res_z=1, num_worker=2)
    bm = BatchManager(cfg, device=None)
    assert bm.num_samples == n == 24 and bm.c_num == 3 and bm.epochs_per_step == 4 / 24.0
    assert bm.y_range[0] == [0.2, 0.8] and bm.y_num == [3, 2, 4]
    x, y = bm.batch()
    bm.stop_thread()
    assert tuple(x.shape) == (4, 8, 6, 2) and tuple(y.shape) == (4, 3)
    assert float(x.abs().max()) <= 1.0 + 1e-6 and float(y.abs().max()) <= 1.0 + 1e-6     # normalised to [-1,1]
    # preprocess == the reference's arithmetic on one file
    xr, yr = preprocess(bm.paths[5], "velocity", bm.x_range, bm.y_range)
    with np.load(bm.paths[5]) as d:
        np.testing.assert_allclose(xr, d["x"] / bm.x_range, rtol=1e-6)
        np.testing.assert_allclose(yr[0], (d["y"][0] - 0.2) / 0.6 * 2 - 1, rtol=1e-5)
    xs = list(bm.batch_(8))
    assert len(xs) == 3 and xs[0][0].shape == (8, 8, 6, 2)
    x0, y0 = bm.denorm(np.ones((1, 2)), np.array([[-1.0, 1.0, 0.0]]))
    assert np.allclose(y0, [[0.2, 0.12, 1.5]]) and np.allclose(x0, bm.x_range)


def test_batch_manager_ae_layout(tmp_path):
    root = str(tmp_path / "smoke3_mov_tiny")
    write_synthetic_dataset(root, (4, 6, 4), num_p=(2, 1), num_frames=3, ae=True)
    cfg = SimpleNamespace(random_seed=1, data_path=root, is_3d=True, arch="ae", data_type="velocity", batch_size=2,
                          res_x=4, res_y=6, res_z=4, num_worker=1)
    bm = BatchManager(cfg, device=None)
    x, y = bm.batch()
    bm.stop_thread()
    assert tuple(x.shape) == (2, 4, 6, 4, 3) and tuple(y.shape) == (2, 2, 3) and bm.dof == 2


def test_synthetic_ae_dataset_has_the_moving_source_layout(tmp_path):
    """scene/smoke3_mov.py:18-38,286-326: v/<scene>_<frame>.npz (x, y [dof, frames]), n.npz (nx, nz [scenes, frames]), args.txt with
    p0 = scenes, p1 = frames; BatchManager(arch='ae') sorts by (scene, frame) and batch_ walks that order."""
    from deep_fluids_amd.data import write_synthetic_ae_dataset
    root = str(tmp_path / "d")
    n = write_synthetic_ae_dataset(root, (4, 6, 4), num_scenes=3, num_frames=4, seed=1)
    assert n == 12
    cfg = SimpleNamespace(random_seed=1, data_path=root, is_3d=True, arch="ae", data_type="velocity", batch_size=2,
                          res_x=4, res_y=6, res_z=4, num_worker=1)
    bm = BatchManager(cfg, device=None)
    assert [os.path.basename(q) for q in bm.paths[:5]] == ["0_0.npz", "0_1.npz", "0_2.npz", "0_3.npz", "1_0.npz"]
    assert bm.label_dim == [2, 4] and bm.c_num == 2 and bm.y_num == [3, 4]
    nn = np.load(os.path.join(root, "n.npz"))
    assert nn["nx"].shape == (3, 4) and nn["nz"].shape == (3, 4)
    got = list(bm.batch_(4))
    assert len(got) == 3 and got[0][0].shape == (4, 4, 6, 4, 3) and np.abs(got[0][0]).max() <= 1.0
    root2 = str(tmp_path / "d2")
    write_synthetic_ae_dataset(root2, (6, 4), num_scenes=2, num_frames=3)
    assert "nz" not in np.load(os.path.join(root2, "n.npz")).files

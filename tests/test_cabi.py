"""CPU-side checks of the C-ABI boundary: the library loads, exports every symbol the public header declares,
and rejects bad arguments with the documented status codes BEFORE touching the device (no GPU needed)."""
import ctypes
import os
import subprocess

import pytest

from deep_fluids_amd import _lib


def test_library_present_and_exports_every_declared_symbol():
    assert os.path.exists(_lib.LIB_PATH), "build it first: python -c 'import __graft_entry__ as g; g.build()'"
    declared = _lib.declared_symbols()
    assert len(declared) >= 28
    out = subprocess.check_output(["nm", "-D", "--defined-only", _lib.LIB_PATH]).decode()
    exported = {line.split()[-1] for line in out.splitlines() if " T " in line}
    missing = [s for s in declared if s not in exported]
    assert not missing, missing
    assert sorted(_lib.SIGNATURES) == declared          # the ctypes table binds exactly the header
    # ... and the release library exports nothing BUT the header: no debug / tuning entry points, no mutable switches
    extra = sorted(s for s in exported if s.startswith("df_") and s not in declared)
    assert not extra, extra


def test_release_library_holds_only_the_production_kernel_instantiations():
    """`wino3d_kernel<DBG, FL, MODE, PREC, XS>` carries diagnosis / experiment switches for the -DDF_TUNING library (tools/); the shipped
    library must hold exactly the production instantiations -- DBG = PREC = XS = 0, MODE in {0 plain, 2 pooled adjoint, 3 up-sampling-aware
    forward on the coarse halo block} -- with these epilogues (FL): -1 run-time flags, 0 none, 2 residual, 4 mask, 9 bias + lrelu,
    25 ... + add-up, 73 ... + sign words, 132 mask from sign words, 345 add-up + sign words without the primary output."""
    out = subprocess.check_output(["nm", "-C", _lib.LIB_PATH]).decode()
    import re
    inst = sorted(set(re.findall(r"wino3d_kernel<([-0-9, ]+)>", out)))
    got = sorted(tuple(int(v) for v in i.split(",")) for i in inst)
    assert got, "no wino3d_kernel symbols in the library (stripped?)"
    assert all(g[0] == 0 and g[3] == 0 and g[4] == 0 and g[2] in (0, 2, 3) for g in got), got
    want = sorted([(0, f, 0, 0, 0) for f in (-1, 0, 2, 4, 9, 25, 73, 132, 345)] + [(0, 0, 2, 0, 0), (0, 9, 3, 0, 0), (0, 73, 3, 0, 0)])
    assert got == want, got
    # the 2-D twin and the weight-gradient kernels have no experiment template switches; the tuning knobs are entry points the release
    # library must not export (checked above by name: no df_debug_*)
    assert "df_debug_" not in out


def test_version_and_error_convention():
    h = _lib.lib()
    assert h.df_version() == 207
    # null pointers / bad extents are argument errors (< 0) caught on the host, with a message
    assert h.df_jacobian3d_fwd(None, None, None, 1, 4, 4, 4, None) == -1
    assert b"null input" in h.df_last_error()
    buf = ctypes.create_string_buffer(64)
    addr = ctypes.addressof(buf)
    addr16 = (addr + 15) & ~15
    assert h.df_jacobian3d_fwd(addr16, addr16, addr16, 1, 1, 4, 4, None) == -2        # extent 1 < 2: DF_ESHAPE
    assert b">= 2" in h.df_last_error()
    assert h.df_jacobian3d_fwd(addr16, addr16 + 4, None, 1, 2, 2, 2, None) == -3      # misaligned output: DF_EALIGN
    assert h.df_conv_fwd(addr16, addr16, None, None, None, addr16, 1, 1, 4, 4, 16, 16, 2, 0, 0.0, None) == -2
    assert h.df_conv_fwd(addr16, addr16, None, None, None, addr16, 1, 1, 4, 4, 16, 16, 1, 8, 0.0, None) == -1  # BIAS flag, no bias
    assert h.df_l1_mean_fwd(addr16, addr16, 16, addr16, addr16, 8, None) == -4        # workspace too small
    assert h.df_conv_packed_elems(27, 128, 128, 0) == 27 * 128 * 128
    assert h.df_conv_packed_elems(27, 128, 3, 0) == 27 * 128 * 32                      # N padded to the 32-wide tile
    assert h.df_conv_packed_elems(27, 128, 3, 1) == 27 * 16 * 128                      # dgrad: K = 3 -> 16
    assert h.df_conv_wgrad_workspace_bytes(16, 64, 96, 64, 128, 128, 3) > 0


def test_python_surface_fails_loudly_without_gpu():
    import torch
    from deep_fluids_amd import ops
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(_lib.DeepFluidsHipError):
        ops.curl3(torch.zeros((1, 4, 4, 4, 3)))
    with pytest.raises(_lib.DeepFluidsHipError):
        ops.l1_mean(torch.zeros(8), torch.zeros(8))


def test_sign_word_layout_decoder_roundtrip():
    """ops.sign_bits_to_mask documents the sign-word layout of conv_wino.hip (kSignBits): encode a random mask with an index-form restatement
    of the kernel's epilogue addressing (byte = (tile block, cout slice, wave = (th, xi_z), cout 16-block, lane = (kq, tl)), bit s =
    (dz, dy, dx)) and decode it back -- ragged extents included (bits of outputs outside the tensor are don't-care)."""
    import numpy as np
    import torch
    from deep_fluids_amd import ops
    rng = np.random.RandomState(0)
    for (B, D, H, W, C) in ((1, 4, 8, 8, 32), (2, 6, 10, 12, 64), (1, 5, 7, 9, 32)):
        m = rng.rand(B, D, H, W, C) > 0.5
        nbz, nby, nbx, ncs = -(-D // 4), -(-H // 8), -(-W // 8), C // 32
        by = np.zeros(B * nbz * nby * nbx * ncs * 1024, np.uint8)
        for b, z, y, x, c in zip(*np.nonzero(m)):
            bz, by_, bx = z // 4, y // 8, x // 8
            th, kq, xz = (z % 4) // 2, (y % 8) // 2, (x % 8) // 2
            s = (z % 2) * 4 + (y % 2) * 2 + (x % 2)
            cs, nb, tl = c // 32, (c // 16) % 2, c % 16
            blk = ((b * nbz + bz) * nby + by_) * nbx + bx
            by[(blk * ncs + cs) * 1024 + (th * 4 + xz) * 128 + nb * 64 + kq * 16 + tl] |= 1 << s
        pad = (-by.size) % 8
        words = torch.from_numpy(np.concatenate([by, np.zeros(pad, np.uint8)])).view(torch.int64)
        got = ops.sign_bits_to_mask(words, (B, D, H, W), C).numpy()
        np.testing.assert_array_equal(got, m)

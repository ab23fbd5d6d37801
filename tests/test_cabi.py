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


def test_version_and_error_convention():
    h = _lib.lib()
    assert h.df_version() == 203
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

#!/usr/bin/env python3
"""Timing probe of the individual HIP kernels at BASELINE cfg3 shapes (run on the GPU box via gpurun).
Prints achieved GB/s (stencils, algorithmic bytes) and TFLOP/s (convs) -- tuning aid, not the bench."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deep_fluids_amd import ops  # noqa: E402
from deep_fluids_amd._lib import call, query  # noqa: E402
from deep_fluids_amd.ops import _ptr, _stream, _pack  # noqa: E402


def timeit(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def main():
    torch.manual_seed(0)
    B, Z, Y, X = 16, 64, 96, 64
    nvox = B * Z * Y * X
    x = torch.rand((B, Z, Y, X, 3), device="cuda") * 2 - 1
    j = torch.empty((B, Z, Y, X, 9), device="cuda"); c = torch.empty((B, Z, Y, X, 3), device="cuda")
    s = _stream()
    t = timeit(lambda: call("df_jacobian3d_fwd", _ptr(x), _ptr(j), _ptr(c), B, Z, Y, X, s))
    print("jacobian3 fwd  j+c : %8.1f us  %7.1f GB/s (60 B/vox)" % (t * 1e6, nvox * 60 / t / 1e9))
    t = timeit(lambda: call("df_jacobian3d_fwd", _ptr(x), None, _ptr(c), B, Z, Y, X, s))
    print("curl3 fwd      c   : %8.1f us  %7.1f GB/s (24 B/vox)" % (t * 1e6, nvox * 24 / t / 1e9))
    t = timeit(lambda: call("df_jacobian3d_fwd", _ptr(x), _ptr(j), None, B, Z, Y, X, s))
    print("jacobian3 fwd  j   : %8.1f us  %7.1f GB/s (48 B/vox)" % (t * 1e6, nvox * 48 / t / 1e9))
    gx = torch.empty_like(x)
    t = timeit(lambda: call("df_jacobian3d_bwd", _ptr(j), None, _ptr(gx), B, Z, Y, X, s))
    print("jacobian3 bwd  gj  : %8.1f us  %7.1f GB/s (48 B/vox)" % (t * 1e6, nvox * 48 / t / 1e9))
    t = timeit(lambda: call("df_jacobian3d_bwd", None, _ptr(c), _ptr(gx), B, Z, Y, X, s))
    print("curl3 bwd      gc  : %8.1f us  %7.1f GB/s (24 B/vox)" % (t * 1e6, nvox * 24 / t / 1e9))
    t = timeit(lambda: gx.copy_(x))
    print("torch copy 75 MB   : %8.1f us  %7.1f GB/s" % (t * 1e6, nvox * 24 / t / 1e9))
    big = torch.empty((B, Z, Y, X, 9), device="cuda")
    t = timeit(lambda: big.copy_(j))
    print("torch copy 226 MB  : %8.1f us  %7.1f GB/s" % (t * 1e6, nvox * 72 / t / 1e9))
    out = torch.empty((), device="cuda")
    nb = query("df_l1_mean_workspace_bytes", j.numel())
    ws = torch.empty(nb // 4, device="cuda")
    t = timeit(lambda: call("df_l1_mean_fwd", _ptr(j), _ptr(big), j.numel(), _ptr(out), _ptr(ws), nb, s))
    print("l1_mean 2x226 MB   : %8.1f us  %7.1f GB/s" % (t * 1e6, nvox * 72 / t / 1e9))
    del j, big, c, gx

    # convs: per level of the cfg3 generator
    F = 128
    for (bz, d, h, w) in [(16, 64, 96, 64), (16, 32, 48, 32), (16, 16, 24, 16), (16, 8, 12, 8), (2, 64, 96, 64)]:
        xin = torch.rand((bz, d, h, w, F), device="cuda") - 0.5
        wt = (torch.rand((3, 3, 3, F, F), device="cuda") - 0.5) * 0.05
        bias = torch.zeros(F, device="cuda")
        y = torch.empty_like(xin)
        wp = _pack(wt, 27, F, F, 0)
        flops = 2.0 * 27 * F * F * bz * d * h * w
        t = timeit(lambda: call("df_conv_fwd", _ptr(xin), _ptr(wp), _ptr(bias), None, None, _ptr(y), bz, d, h, w, F, F, 3,
                                8 | 1, 0.2, s), iters=5, warm=2)
        print("conv3d fwd  %2dx%3dx%3dx%3d F128: %9.3f ms  %6.1f TFLOP/s" % (bz, d, h, w, t * 1e3, flops / t / 1e12))
        gw = torch.empty_like(wt); gb = torch.empty(F, device="cuda")
        nb = query("df_conv_wgrad_workspace_bytes", bz, d, h, w, F, F, 3)
        ws = torch.empty(nb // 4 + 1, device="cuda")
        t = timeit(lambda: call("df_conv_wgrad", _ptr(xin), _ptr(y), _ptr(gw), _ptr(gb), bz, d, h, w, F, F, 3, _ptr(ws),
                                nb, s), iters=5, warm=2)
        print("conv3d wgrad %2dx%3dx%3dx%3d F128: %9.3f ms  %6.1f TFLOP/s" % (bz, d, h, w, t * 1e3, flops / t / 1e12))
        del xin, y, ws
    # last conv 128 -> 3 and its dgrad
    bz, d, h, w = 16, 64, 96, 64
    xin = torch.rand((bz, d, h, w, F), device="cuda") - 0.5
    wt = (torch.rand((3, 3, 3, F, 3), device="cuda") - 0.5) * 0.05
    y3 = torch.empty((bz, d, h, w, 3), device="cuda")
    wp = _pack(wt, 27, F, 3, 0)
    t = timeit(lambda: call("df_conv_fwd", _ptr(xin), _ptr(wp), None, None, None, _ptr(y3), bz, d, h, w, F, 3, 3, 0, 0.0, s),
               iters=5, warm=2)
    print("conv3d fwd 128->3  : %9.3f ms" % (t * 1e3))
    wpd = _pack(wt, 27, F, 3, 1)
    t = timeit(lambda: call("df_conv_fwd", _ptr(y3), _ptr(wpd), None, None, None, _ptr(xin), bz, d, h, w, 3, F, 3, 0, 0.0, s),
               iters=5, warm=2)
    print("conv3d dgrad 3->128: %9.3f ms" % (t * 1e3))
    gw = torch.empty_like(wt)
    nb = query("df_conv_wgrad_workspace_bytes", bz, d, h, w, F, 3, 3)
    ws = torch.empty(nb // 4 + 1, device="cuda")
    t = timeit(lambda: call("df_conv_wgrad", _ptr(xin), _ptr(y3), _ptr(gw), None, bz, d, h, w, F, 3, 3, _ptr(ws), nb, s),
               iters=5, warm=2)
    print("conv3d wgrad 128x3 : %9.3f ms" % (t * 1e3))
    up = torch.empty((bz, d, h, w, F), device="cuda")
    lo = torch.rand((bz, d // 2, h // 2, w // 2, F), device="cuda")
    t = timeit(lambda: call("df_upsample2x_fwd", _ptr(lo), _ptr(up), bz, d // 2, h // 2, w // 2, F, 1, s), iters=5, warm=2)
    print("upsample3 fwd ->3GiB: %9.3f ms  %7.1f GB/s" % (t * 1e3, up.numel() * 4 * 1.125 / t / 1e9))
    t = timeit(lambda: call("df_upsample2x_bwd", _ptr(up), _ptr(lo), bz, d // 2, h // 2, w // 2, F, 1, s), iters=5, warm=2)
    print("upsample3 bwd       : %9.3f ms  %7.1f GB/s" % (t * 1e3, up.numel() * 4 * 1.125 / t / 1e9))
    t = timeit(lambda: call("df_lrelu_bwd", _ptr(up), _ptr(xin), _ptr(up), 0.2, up.numel(), s), iters=5, warm=2)
    print("lrelu_bwd 3x3GiB    : %9.3f ms  %7.1f GB/s" % (t * 1e3, up.numel() * 12 / t / 1e9))


if __name__ == "__main__" and len(sys.argv) == 1:
    main()


def probe_bf16x3():
    """Kernel-level timing of the opt-in bf16x3 mode at the top-resolution shape."""
    from deep_fluids_amd import ops as _ops
    s = _stream()
    F = 128
    bz, d, h, w = 16, 64, 96, 64
    xin = torch.rand((bz, d, h, w, F), device="cuda") - 0.5
    wt = (torch.rand((3, 3, 3, F, F), device="cuda") - 0.5) * 0.05
    bias = torch.zeros(F, device="cuda")
    y = torch.empty_like(xin)
    n = query("df_conv_packed_elems_bf16x3", 27, F, F, 0)
    wp = torch.empty(n, device="cuda")
    call("df_conv_pack_weights_bf16x3", _ptr(wt), _ptr(wp), 27, F, F, 0, s)
    flops = 2.0 * 27 * F * F * bz * d * h * w
    t = timeit(lambda: call("df_conv_fwd_bf16x3", _ptr(xin), _ptr(wp), _ptr(bias), None, None, _ptr(y), bz, d, h, w, F, F, 3,
                            9, 0.2, s), iters=5, warm=2)
    print("bf16x3 conv3d fwd  top-res: %8.3f ms  %6.1f TFLOP/s fp32-equivalent (%.0f bf16 TFLOP/s on the matrix pipe)" % (t * 1e3, flops / t / 1e12, 3 * flops / t / 1e12))
    gw = torch.empty_like(wt); gb = torch.empty(F, device="cuda")
    nb = query("df_conv_wgrad_workspace_bytes", bz, d, h, w, F, F, 3)
    ws = torch.empty(nb // 4 + 1, device="cuda")
    t = timeit(lambda: call("df_conv_wgrad_bf16x3", _ptr(xin), _ptr(y), _ptr(gw), _ptr(gb), bz, d, h, w, F, F, 3, _ptr(ws), nb, s),
               iters=5, warm=2)
    print("bf16x3 conv3d wgrad top-res: %8.3f ms  %6.1f TFLOP/s fp32-equivalent (%.0f bf16 TFLOP/s on the matrix pipe)" % (t * 1e3, flops / t / 1e12, 3 * flops / t / 1e12))


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "bf16x3":
    probe_bf16x3()

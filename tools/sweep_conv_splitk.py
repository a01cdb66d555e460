"""Sweep of the split-K thresholds of the conv launcher on the UNet's low-resolution layers (tools build).
   python tools/sweep_conv_splitk.py [B]"""
import sys
import _toolslib  # noqa: F401
import ctypes as C
import torch
import torch.nn.functional as F
from audiopure_amd import _native as N

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = torch.device("cuda:0")
lib = N.lib()
lib.ap_debug_conv_splitk.argtypes = [C.c_int, C.c_int]
N.use_conv_workspace(dev)
shapes = [(256, 8, 256, 3), (512, 8, 256, 3), (256, 4, 256, 3), (512, 4, 256, 3), (256, 8, 768, 1), (256, 8, 256, 1), (512, 8, 256, 1), (512, 4, 256, 1),
          (256, 16, 256, 1), (256, 16, 768, 1)]
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for (cin, hw, cout, k) in shapes:
    x = torch.randn(B, cin, hw, hw, device=dev)
    w = torch.randn(cout, cin, k, k, device=dev) * 0.05
    b = torch.randn(cout, device=dev)
    wT = torch.empty(lib.ap_conv2d_packed_elems(cout, cin, k, k, 1), device=dev)
    N.check(lib.ap_conv2d_pack(N.ptr(w), None, N.ptr(wT), cout, cin, k, k, 1, N.stream()))
    out = torch.empty(B, cout, hw, hw, device=dev)
    ref = F.conv2d(x, w, b, padding=k // 2)
    fl = 2.0 * B * hw * hw * cout * cin * k * k
    row = []
    for (t2, cap) in [(384, 768), (640, 768), (640, 1536), (1100, 1536), (1100, 3072), (2200, 3072)]:
        lib.ap_debug_conv_splitk(t2, cap)
        for _ in range(3):
            N.check(lib.ap_conv2d_fwd(N.ptr(x), N.ptr(wT), N.ptr(b), None, N.ptr(out), B, cin, hw, hw, cout, k, k, 1, k // 2, 1, 0, cin, 0, N.stream()))
        e0.record()
        n = 10
        for _ in range(n):
            N.check(lib.ap_conv2d_fwd(N.ptr(x), N.ptr(wT), N.ptr(b), None, N.ptr(out), B, cin, hw, hw, cout, k, k, 1, k // 2, 1, 0, cin, 0, N.stream()))
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / n
        err = float((out - ref).abs().max() / ref.abs().max())
        row.append(f"({t2},{cap}): {ms * 1e3:7.1f} us {fl / ms / 1e9 / 157.3:.3f}" + ("" if err < 1e-5 else f" ERR {err:.1e}"))
    print(f"{cin:4d}->{cout:4d} {k}x{k} @{hw:2d}^2  " + "  ".join(row), flush=True)
lib.ap_debug_conv_splitk(384, 768)

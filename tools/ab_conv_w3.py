"""A/B of the F(2,3) conv kernel's staging forms and against the direct kernels (tools build).  python tools/ab_conv_w3.py [B]"""
import sys
import _toolslib  # noqa: F401
import ctypes as C
import torch
import torch.nn.functional as F
from audiopure_amd import _native as N

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = torch.device("cuda:0")
lib = N.lib()
lib.ap_debug_conv_w3.argtypes = [C.c_int, C.c_int]
lib.ap_debug_conv_w3_nb.argtypes = [C.c_int]
N.use_conv_workspace(dev)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for (cin, hw, cout) in [(128, 32, 128), (256, 32, 128), (256, 16, 256), (512, 16, 256), (256, 8, 256), (512, 8, 256), (256, 4, 256)]:
    x = torch.randn(B, cin, hw, hw, device=dev)
    w = torch.randn(cout, cin, 3, 3, device=dev) * 0.05
    b = torch.randn(cout, device=dev)
    wT = torch.empty(lib.ap_conv2d_packed_elems(cout, cin, 3, 3, 1), device=dev)
    N.check(lib.ap_conv2d_pack(N.ptr(w), None, N.ptr(wT), cout, cin, 3, 3, 1, N.stream()))
    out = torch.empty(B, cout, hw, hw, device=dev)
    ref = F.conv2d(x, w, b, padding=1)
    fl = 2.0 * B * hw * hw * cout * cin * 9
    row = []
    for name, on, nb in (("direct", 0, 0), ("w3 gather", 1, 0), ("w3 neighbour", 1, 1)):
        lib.ap_debug_conv_w3(on, 0)
        lib.ap_debug_conv_w3_nb(nb)
        for _ in range(3):
            N.check(lib.ap_conv2d_fwd(N.ptr(x), N.ptr(wT), N.ptr(b), None, N.ptr(out), B, cin, hw, hw, cout, 3, 3, 1, 1, 1, 0, cin, 0, N.stream()))
        e0.record()
        n = 10
        for _ in range(n):
            N.check(lib.ap_conv2d_fwd(N.ptr(x), N.ptr(wT), N.ptr(b), None, N.ptr(out), B, cin, hw, hw, cout, 3, 3, 1, 1, 1, 0, cin, 0, N.stream()))
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / n
        err = float((out - ref).abs().max() / ref.abs().max())
        row.append(f"{name}: {ms * 1e3:7.1f} us {fl / ms / 1e9 / 157.3:.3f}" + ("" if err < 1e-5 else f" ERR {err:.1e}"))
    print(f"{cin:4d}->{cout:4d} 3x3 @{hw:2d}^2  " + "   ".join(row), flush=True)
lib.ap_debug_conv_w3(1, 0); lib.ap_debug_conv_w3_nb(1)

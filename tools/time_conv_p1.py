"""One pointwise layer through ap_conv2d_fwd, timed with HIP events: python tools/time_conv_p1.py B Cin HW Cout [reps]   (AUDIOPURE_HIP_LIB selects a variant build)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audiopure_amd import _native as N
lib = N.lib()
dev = torch.device("cuda:0")
B, Cin, HW, Cout = (int(a) for a in sys.argv[1:5])
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 20
H = W = int(HW ** 0.5)
x = torch.randn(B, Cin, H, W, device=dev); w = torch.randn(Cout, Cin, 1, 1, device=dev) * 0.05
wT = torch.empty(lib.ap_conv2d_packed_elems(Cout, Cin, 1, 1, 1), device=dev)
N.check(lib.ap_conv2d_pack(N.ptr(w), None, N.ptr(wT), Cout, Cin, 1, 1, 1, N.stream()))
out = torch.empty(B, Cout, H, W, device=dev)
def run():
    N.check(lib.ap_conv2d_fwd(N.ptr(x), N.ptr(wT), None, None, N.ptr(out), B, Cin, H, W, Cout, 1, 1, 1, 0, 1, 0, Cin, 0, N.stream()))
for _ in range(3): run()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps): run()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
print(f"{os.environ.get('AUDIOPURE_HIP_LIB', 'product')}: B={B} {Cin}->{Cout} @{H}x{W}: {ms * 1e3:.1f} us = {2.0 * B * H * W * Cin * Cout / (ms * 1e-3) / 1e12:.1f} TFLOP/s ({2.0 * B * H * W * Cin * Cout / (ms * 1e-3) / 157.3e12:.3f} of peak)")

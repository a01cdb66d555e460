"""One conv shape alone, `reps` launches (for rocprofv3 passes / quick timing):
python tools/run_conv.py B Cin H W Cout k [stride] [groups] [reps]"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audiopure_amd import _native as N
a = [int(v) for v in sys.argv[1:]]
B, Cin, H, W, Cout, k = a[:6]
stride = a[6] if len(a) > 6 else 1
groups = a[7] if len(a) > 7 else 1
reps = a[8] if len(a) > 8 else 5
dev = torch.device("cuda:0"); lib = N.lib()
torch.manual_seed(0)
x = torch.randn(B, Cin, H, W, device=dev)
w = torch.randn(Cout, Cin // groups, k, k, device=dev) * 0.05
bias = torch.randn(Cout, device=dev)
wT = torch.empty(lib.ap_conv2d_packed_elems(Cout, Cin // groups, k, k, groups), device=dev)
N.check(lib.ap_conv2d_pack(N.ptr(w), None, N.ptr(wT), Cout, Cin // groups, k, k, groups, N.stream()))
pad = k // 2
Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
out = torch.empty(B, Cout, Ho, Wo, device=dev)
def run():
    N.check(lib.ap_conv2d_fwd(N.ptr(x), N.ptr(wT), N.ptr(bias), None, N.ptr(out), B, Cin, H, W, Cout, k, k, stride, pad, groups, 0, Cin, 0, N.stream()))
N.use_conv_workspace(dev) if (len(a) <= 9 or a[9]) else None
run(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps): run()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
fl = 2.0 * B * Ho * Wo * Cout * (Cin // groups) * k * k
ref = torch.nn.functional.conv2d(x, w, bias, stride=stride, padding=pad, groups=groups)
print(f"conv B{B} Cin{Cin} {H}x{W} Cout{Cout} k{k} s{stride} g{groups}: {ms:.3f} ms  {fl / ms / 1e9:.1f} TFLOP/s = {fl / ms / 1e9 / 157.3:.3f}  max err vs torch {float((out - ref).abs().max()):.2e}")

"""Run one conv layer repeatedly (for rocprofv3 passes): python tools/run_conv.py B Cin H Cout k reps"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audiopure_amd import _native as N
B, Cin, H, Cout, k, reps = [int(v) for v in sys.argv[1:7]]
dev = torch.device("cuda:0")
lib = N.lib()
N.use_conv_workspace(dev)
x = torch.randn(B, Cin, H, H, device=dev)
w = torch.randn(Cout, Cin, k, k, device=dev) * 0.05
b = torch.randn(Cout, device=dev)
wT = torch.empty(lib.ap_conv2d_packed_elems(Cout, Cin, k, k, 1), device=dev)
N.check(lib.ap_conv2d_pack(N.ptr(w), None, N.ptr(wT), Cout, Cin, k, k, 1, N.stream()))
out = torch.empty(B, Cout, H, H, device=dev)
for _ in range(reps):
    N.check(lib.ap_conv2d_fwd(N.ptr(x), N.ptr(wT), N.ptr(b), None, N.ptr(out), B, Cin, H, H, Cout, k, k, 1, k // 2, 1, 0, Cin, 0, N.stream()))
torch.cuda.synchronize()
print("done")

"""Board power and shader clock while one conv-as-GEMM shape runs back to back: python tools/power_check_conv.py [seconds] B Cin H W Cout k"""
import os, re, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from audiopure_amd import _native as N
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
shapes = [tuple(int(v) for v in sys.argv[2:8])] if len(sys.argv) >= 8 else [(256, 128, 32, 32, 128, 3), (256, 256, 16, 16, 256, 3), (256, 256, 32, 32, 128, 3), (256, 256, 16, 16, 768, 1)]
dev = torch.device("cuda:0"); lib = N.lib()
def smi():
    out = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True).stdout
    p = re.search(r"Power \(W\): ([0-9.]+)", out); s = re.search(r"sclk clock level: \S+ \((\d+)Mhz\)", out)
    return (float(p.group(1)) if p else None, int(s.group(1)) if s else None)
print("idle:", smi())
N.use_conv_workspace(dev)
for (B, Cin, H, W, Cout, k) in shapes:
    torch.manual_seed(0)
    x = torch.randn(B, Cin, H, W, device=dev); w = torch.randn(Cout, Cin, k, k, device=dev) * 0.05; bias = torch.randn(Cout, device=dev)
    wT = torch.empty(lib.ap_conv2d_packed_elems(Cout, Cin, k, k, 1), device=dev)
    N.check(lib.ap_conv2d_pack(N.ptr(w), None, N.ptr(wT), Cout, Cin, k, k, 1, N.stream()))
    out = torch.empty(B, Cout, H, W, device=dev)
    def launch(n):
        for _ in range(n):
            N.check(lib.ap_conv2d_fwd(N.ptr(x), N.ptr(wT), N.ptr(bias), None, N.ptr(out), B, Cin, H, W, Cout, k, k, 1, k // 2, 1, 0, Cin, 0, N.stream()))
    launch(3); torch.cuda.synchronize()
    samples, stop = [], [False]
    def sampler():
        while not stop[0]:
            samples.append(smi()); time.sleep(0.2)
    th = threading.Thread(target=sampler); th.start()
    t0 = time.time(); n = 0
    while time.time() - t0 < secs:
        launch(100); torch.cuda.synchronize(); n += 100
    el = time.time() - t0; stop[0] = True; th.join()
    ps = [p for p, _ in samples[2:] if p]; cs = [c for _, c in samples[2:] if c]
    ms = el / n * 1e3; fl = 2.0 * B * H * W * Cout * Cin * k * k
    print(f"conv B{B} {Cin}->{Cout} {H}x{W} k{k}: {ms:.3f} ms = {fl / ms / 1e9:.1f} TFLOP/s ({fl / ms / 1e9 / 157.3:.3f} of the 2.4 GHz peak); power {sum(ps) / len(ps):.0f} W, sclk {sum(cs) // len(cs)} MHz "
          f"-> {fl / ms / 1e9 / (157.3 * (sum(cs) / len(cs)) / 2400):.3f} of the peak at that clock")

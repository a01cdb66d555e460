"""Board power and shader clock while one residual-block kernel runs back to back (rocm-smi sampled from a side thread):
is the clock a kernel holds the board's power cap at work?   python tools/power_check.py [seconds per mode] [B]"""
import os, re, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from audiopure_amd import synth, _native as N
from audiopure_amd.diffusion_models.DiffWave_Unconditional.WaveNet import WaveNet_Speech_Commands

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 6.0
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
dev = torch.device("cuda:0")
L = 16000


def smi():
    out = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True).stdout
    p = re.search(r"Power \(W\): ([0-9.]+)", out)
    s = re.search(r"sclk clock level: \S+ \((\d+)Mhz\)", out)
    return (float(p.group(1)) if p else None, int(s.group(1)) if s else None)


cap = subprocess.run(["rocm-smi", "--showmaxpower"], capture_output=True, text=True).stdout
m = re.search(r"Power \(W\): ([0-9.]+)", cap)
print("power cap (W):", m.group(1) if m else "?", " idle:", smi())
torch.manual_seed(0)
h = torch.randn(B, 256, L, device=dev); ho = torch.empty_like(h); sk = torch.zeros_like(h); pt = torch.randn(256, device=dev)
zero = os.environ.get("AP_ZERO") == "1"
if zero:
    h.zero_(); pt.zero_()
for prec in ("bf16", "f32s", "f32d", "f32"):
    net = WaveNet_Speech_Commands(**dict(synth.FULL_WAVENET_CONFIG)).to(dev).set_precision(prec)
    eng = net.engine(); lib = eng.lib
    for layer in (9,):
        samples, stop = [], False

        def sampler():
            while not stop:
                samples.append(smi())
                time.sleep(0.25)
        def launch(n):
            for _ in range(n):
                N.check(lib.ap_resblock_fwd(eng.ctx, layer, N.ptr(h), N.ptr(pt), N.ptr(ho), N.ptr(sk), 1, B, L, N.stream()))
        launch(3); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); launch(10); e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        th = threading.Thread(target=sampler); th.start()
        t0 = time.time(); n = 0
        while time.time() - t0 < secs:
            launch(20); torch.cuda.synchronize(); n += 20
        el = time.time() - t0
        stop = True; th.join()
        ps = [p for p, _ in samples[2:] if p]; cs = [c for _, c in samples[2:] if c]
        print(f"{prec:5s} layer {layer} B={B}{' zero data' if zero else ''}: {ms:.3f} ms cold-ish, {el / n * 1e3:.3f} ms sustained over {el:.1f} s; "
              f"power W min/mean/max {min(ps):.0f}/{sum(ps) / len(ps):.0f}/{max(ps):.0f}; sclk MHz min/mean/max {min(cs)}/{sum(cs) // len(cs)}/{max(cs)} ({len(ps)} samples)")
    del net, eng

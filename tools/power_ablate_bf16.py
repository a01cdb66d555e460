"""Energy accounting of the bf16 residual block (tools build): board power x time per launch for the timing-only ablations
(outputs wrong by construction), layer 9 (d = 512), sustained for a few seconds each with rocm-smi sampled from a side thread.
python tools/power_ablate_bf16.py [B] [seconds] [dbg bits ...]"""
import os as _os, sys as _sys; _sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__))); import _toolslib  # noqa: E401,E702
import re, subprocess, sys, threading, time, ctypes as C, torch
from audiopure_amd import synth, _native as N
from audiopure_amd.diffusion_models.DiffWave_Unconditional.WaveNet import WaveNet_Speech_Commands
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 3.0
variants = [int(a, 0) for a in sys.argv[3:]] or [0, 1024, 1, 2, 4, 8, 16, 32, 64, 96, 128, 256, 384, 31, 384 + 31 + 96]
NAMES = {0: "product kernel", 0x1000000: "stores: default policy (exact)", 0x2000000: "stores: sc0 + nt (exact)", 0x3000000: "stores: sc1 + nt (exact)", 0x5000000: "stores: sc1 (exact)", 0x4000: "default policy everywhere (exact)", 1024: "X from cache-resident lines (no HBM reads of h)", 96: "no gate math, no GEMM2 MFMA", 1: "no weight requests (GEMM1)", 2: "no X requests", 4: "no pack", 8: "no GEMM1 MFMA", 16: "no B-fragment LDS reads",
         32: "no gate math", 64: "no GEMM2 MFMA", 128: "no read-modify-write loads", 256: "no stores", 384: "no RMW loads, no stores",
         31: "GEMM1 emptied (1+2+4+8+16)", 384 + 31 + 96: "everything off but the loop"}
L = 16000
def smi():
    out = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True).stdout
    p = re.search(r"Power \(W\): ([0-9.]+)", out); s = re.search(r"sclk clock level: \S+ \((\d+)Mhz\)", out)
    return (float(p.group(1)) if p else None, int(s.group(1)) if s else None)
idle = smi()
print("idle:", idle)
torch.manual_seed(0)
h = torch.randn(B, 256, L, device=dev); ho = torch.empty_like(h); sk = torch.zeros_like(h); pt = torch.randn(256, device=dev)
net = WaveNet_Speech_Commands(**dict(synth.FULL_WAVENET_CONFIG)).to(dev).set_precision("bf16")
eng = net.engine(); lib = C.CDLL(N.LIB_PATH); lib.ap_debug_bf16_dbg.argtypes = [C.c_int]
layer = 9
def launch(n):
    for _ in range(n):
        N.check(eng.lib.ap_resblock_fwd(eng.ctx, layer, N.ptr(h), N.ptr(pt), N.ptr(ho), N.ptr(sk), 1, B, L, N.stream()))
print(f"{'variant':50s} {'ms':>7s} {'W':>6s} {'MHz':>5s} {'J/launch':>9s} {'dyn J':>7s}")
for v in variants:
    lib.ap_debug_bf16_dbg(v)
    launch(3); torch.cuda.synchronize()
    samples, stop = [], [False]
    def sampler():
        while not stop[0]:
            samples.append(smi()); time.sleep(0.2)
    th = threading.Thread(target=sampler); th.start()
    t0 = time.time(); n = 0
    while time.time() - t0 < secs:
        launch(20); torch.cuda.synchronize(); n += 20
    el = time.time() - t0
    stop[0] = True; th.join()
    ps = [p for p, _ in samples[2:] if p]; cs = [c for _, c in samples[2:] if c]
    ms = el / n * 1e3; pw = sum(ps) / len(ps)
    print(f"{NAMES.get(v, hex(v)):50s} {ms:7.3f} {pw:6.0f} {sum(cs) // len(cs):5d} {pw * ms / 1e3:9.3f} {(pw - idle[0]) * ms / 1e3:7.3f}", flush=True)
lib.ap_debug_bf16_dbg(0)

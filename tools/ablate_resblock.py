import os as _os, sys as _sys; _sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__))); import _toolslib  # noqa: E401,E702  (-DAP_TOOLS library)
import sys, ctypes as C, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from audiopure_amd import synth, _native as N
from audiopure_amd.diffusion_models.DiffWave_Unconditional.WaveNet import WaveNet_Speech_Commands
dev = torch.device("cuda:0")
cfg = dict(synth.FULL_WAVENET_CONFIG)
net = WaveNet_Speech_Commands(**cfg).to(dev)
if len(sys.argv) > 3: net.set_precision(sys.argv[3])
eng = net.engine()
lib = eng.lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
L = int(sys.argv[4]) if len(sys.argv) > 4 else 16000
h = torch.randn(B, 256, L, device=dev); ho = torch.empty_like(h); sk = torch.zeros_like(h)
pt = torch.randn(256, device=dev)
lib.ap_debug_ablate.argtypes = [C.c_int]
if len(sys.argv) > 2: print('tile', sys.argv[2], lib.ap_debug_tile(int(sys.argv[2])))
def run(mask, layer, reps=5):
    lib.ap_debug_ablate(mask)
    for _ in range(2):
        lib.ap_resblock_fwd(eng.ctx, layer, N.ptr(h), N.ptr(pt), N.ptr(ho), N.ptr(sk), 1, B, L, N.stream())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        lib.ap_resblock_fwd(eng.ctx, layer, N.ptr(h), N.ptr(pt), N.ptr(ho), N.ptr(sk), 1, B, L, N.stream())
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    return ms, 16.777e9 * B / ms / 1e9
if len(sys.argv) > 5: lib.ap_debug_stagger(int(sys.argv[5]))
for layer in (5,):
    for mask in (0, 1):
        ms, tf = run(mask, layer)
        print(f"layer {layer:2d} mask {mask:2d}: {ms:8.3f} ms  {tf:7.1f} TFLOP/s-equivalent", flush=True)

"""Time the residual-block kernel per precision mode: python tools/time_resblock.py [B] [modes...]"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audiopure_amd import synth, _native as N
from audiopure_amd.diffusion_models.DiffWave_Unconditional.WaveNet import WaveNet_Speech_Commands
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
modes = sys.argv[2:] or ["f32", "f32d", "f32s", "bf16"]
L = 16000
h = torch.randn(B, 256, L, device=dev); ho = torch.empty_like(h); sk = torch.zeros_like(h)
pt = torch.randn(256, device=dev)
for mode in modes:
    net = WaveNet_Speech_Commands(**dict(synth.FULL_WAVENET_CONFIG)).to(dev)
    net.set_precision(mode)
    eng = net.engine(); lib = eng.lib
    for layer in (5, 9):
        for _ in range(2):
            lib.ap_resblock_fwd(eng.ctx, layer, N.ptr(h), N.ptr(pt), N.ptr(ho), N.ptr(sk), 1, B, L, N.stream())
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            lib.ap_resblock_fwd(eng.ctx, layer, N.ptr(h), N.ptr(pt), N.ptr(ho), N.ptr(sk), 1, B, L, N.stream())
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        print(f"{mode:5s} layer {layer} B={B}: {ms:8.3f} ms  {B * 16.777216 / ms:8.1f} TFLOP/s fp32-equivalent  {B * 65.536e-3 / ms:6.2f} TB/s algorithmic")

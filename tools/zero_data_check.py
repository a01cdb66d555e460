"""Is the bf16 block limited by the clock the chip holds under load?  Same launch on random and on all-zero activations
(identical instruction stream; zero operands toggle less and let the chip clock higher): python tools/zero_data_check.py [B]"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audiopure_amd import synth, _native as N
from audiopure_amd.diffusion_models.DiffWave_Unconditional.WaveNet import WaveNet_Speech_Commands
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
L = 16000
net = WaveNet_Speech_Commands(**dict(synth.FULL_WAVENET_CONFIG)).to(dev)
def t(mode, h, pt, layer=9, reps=6):
    net.set_precision(mode)
    eng = net.engine(); lib = eng.lib
    ho = torch.empty_like(h); sk = torch.zeros_like(h)
    for _ in range(3):
        lib.ap_resblock_fwd(eng.ctx, layer, N.ptr(h), N.ptr(pt), N.ptr(ho), N.ptr(sk), 1, B, L, N.stream())
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        lib.ap_resblock_fwd(eng.ctx, layer, N.ptr(h), N.ptr(pt), N.ptr(ho), N.ptr(sk), 1, B, L, N.stream())
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
hr, pr = torch.randn(B, 256, L, device=dev), torch.randn(256, device=dev)
hz, pz = torch.zeros(B, 256, L, device=dev), torch.zeros(256, device=dev)
for mode in ("bf16", "f32"):
    for rnd in range(2):
        a, b = t(mode, hr, pr), t(mode, hz, pz)
        print(f"{mode}: random activations {a:7.3f} ms   all-zero activations {b:7.3f} ms   ratio {a / b:.3f}", flush=True)

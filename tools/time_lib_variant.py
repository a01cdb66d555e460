"""Time the bf16 residual block of whatever library AUDIOPURE_HIP_LIB names (A/B of compiler flags or source variants built
side by side): AUDIOPURE_HIP_LIB=/path/lib.so python tools/time_lib_variant.py"""
import sys, os, ctypes as C, torch
sys.path.insert(0, os.getcwd())
from audiopure_amd import synth, _native as N
from audiopure_amd.diffusion_models.DiffWave_Unconditional.WaveNet import WaveNet_Speech_Commands
dev = torch.device("cuda:0")
B, L = 256, 16000
h = torch.randn(B, 256, L, device=dev); ho = torch.empty_like(h); sk = torch.zeros_like(h); pt = torch.randn(256, device=dev)
net = WaveNet_Speech_Commands(**dict(synth.FULL_WAVENET_CONFIG)).to(dev); net.set_precision("bf16"); eng = net.engine()
def t(layer, reps=8):
    for _ in range(2): N.check(eng.lib.ap_resblock_fwd(eng.ctx, layer, N.ptr(h), N.ptr(pt), N.ptr(ho), N.ptr(sk), 1, B, L, N.stream()))
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(reps): eng.lib.ap_resblock_fwd(eng.ctx, layer, N.ptr(h), N.ptr(pt), N.ptr(ho), N.ptr(sk), 1, B, L, N.stream())
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / reps
for r in range(2): print(os.environ.get("AUDIOPURE_HIP_LIB","default"), " ".join(f"L{l}:{t(l):.3f}" for l in (1, 5, 9, 11)), flush=True)

"""Run only the residual-block kernel (for rocprofv3 --pmc passes): python tools/run_resblock.py B precision reps [layer]"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audiopure_amd import synth, _native as N
from audiopure_amd.diffusion_models.DiffWave_Unconditional.WaveNet import WaveNet_Speech_Commands
dev = torch.device("cuda:0")
net = WaveNet_Speech_Commands(**dict(synth.FULL_WAVENET_CONFIG)).to(dev)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
net.set_precision(sys.argv[2] if len(sys.argv) > 2 else "f32")
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
layer = int(sys.argv[4]) if len(sys.argv) > 4 else 5
eng = net.engine(); lib = eng.lib
L = 16000
h = torch.randn(B, 256, L, device=dev); ho = torch.empty_like(h); sk = torch.zeros_like(h)
pt = torch.randn(256, device=dev)
for _ in range(reps):
    lib.ap_resblock_fwd(eng.ctx, layer, N.ptr(h), N.ptr(pt), N.ptr(ho), N.ptr(sk), 1, B, L, N.stream())
torch.cuda.synchronize()
print("done")

"""Run only the residual-block kernel, `reps` launches per listed layer in order (for rocprofv3 passes that are split by dispatch
order afterwards): python tools/run_resblock_layers.py B precision reps layer [layer ...]"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audiopure_amd import synth, _native as N
from audiopure_amd.diffusion_models.DiffWave_Unconditional.WaveNet import WaveNet_Speech_Commands
dev = torch.device("cuda:0")
net = WaveNet_Speech_Commands(**dict(synth.FULL_WAVENET_CONFIG)).to(dev)
B, prec, reps = int(sys.argv[1]), sys.argv[2], int(sys.argv[3])
layers = [int(a) for a in sys.argv[4:]]
net.set_precision(prec)
eng = net.engine(); lib = eng.lib
L = 16000
torch.manual_seed(0)
h = torch.randn(B, 256, L, device=dev); ho = torch.empty_like(h); sk = torch.zeros_like(h)
pt = torch.randn(256, device=dev)
for layer in layers:
    for _ in range(reps):
        N.check(lib.ap_resblock_fwd(eng.ctx, layer, N.ptr(h), N.ptr(pt), N.ptr(ho), N.ptr(sk), 1, B, L, N.stream()))
torch.cuda.synchronize()
print("done")

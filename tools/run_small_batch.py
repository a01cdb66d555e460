"""Three purifications of a small batch (the chain replays a captured HIP graph from the second call on): python tools/run_small_batch.py [B] [mode]"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audiopure_amd import synth
from audiopure_amd.diffusion_models.DiffWave_Unconditional.WaveNet import WaveNet_Speech_Commands
from audiopure_amd.diffusion_models.DiffWave_Unconditional.util import calc_diffusion_hyperparams
from audiopure_amd.diffusion_models.diffwave_ddpm import DiffWave
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
mode = sys.argv[2] if len(sys.argv) > 2 else "bf16"
dev = torch.device("cuda:0")
cfg = dict(synth.FULL_WAVENET_CONFIG)
net = WaveNet_Speech_Commands(**cfg)
net.load_state_dict({k: torch.from_numpy(v) for k, v in synth.wavenet_state_dict(cfg, 0).items()})
net = net.to(dev).set_precision(mode)
dw = DiffWave(model=net, diffusion_hyperparams=calc_diffusion_hyperparams(**synth.DIFFUSION_CONFIG), reverse_timestep=5)
dw.set_noise_source(("philox", 1234, 0))
x = torch.rand(B, 1, 16000, device=dev) - 0.5
with torch.no_grad():
    for _ in range(3):
        y = dw(x)
        torch.cuda.synchronize()
print(float(y.abs().max()))

"""one white-box gradient step through the purifier (RevDiffWave Euler chain, shipped config):
forward + backward w.r.t. the audio.  python tools/bench_whitebox.py [B] [t*] [f32|f32s|bf16]"""
import sys, os, time, types, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audiopure_amd import synth
from audiopure_amd.diffusion_models.DiffWave_Unconditional.WaveNet import WaveNet_Speech_Commands
from audiopure_amd.diffusion_models.DiffWave_Unconditional.util import calc_diffusion_hyperparams
from audiopure_amd.diffusion_models.diffwave_ddpm import DiffWave
from audiopure_amd.diffusion_models.diffwave_sde import RevDiffWave
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
t_star = int(sys.argv[2]) if len(sys.argv) > 2 else 5
mode = sys.argv[3] if len(sys.argv) > 3 else "f32"
cfg = dict(synth.FULL_WAVENET_CONFIG)
net = WaveNet_Speech_Commands(**cfg)
net.load_state_dict({k: torch.from_numpy(v) for k, v in synth.wavenet_state_dict(cfg, 0).items()})
net = net.to(dev).set_precision(mode)
dw = DiffWave(model=net, diffusion_hyperparams=calc_diffusion_hyperparams(**synth.DIFFUSION_CONFIG), reverse_timestep=t_star)
args = types.SimpleNamespace(t=t_star, rand_t=False, t_delta=0, use_bm=False, sample_step=1, score_type="guided_diffusion")
runner = RevDiffWave.from_model(dw, args)
x = (torch.rand(B, 1, 16000, device=dev) - 0.5)
w = torch.randn(B, 1, 16000, device=dev)
def step():
    xg = x.clone().requires_grad_(True)
    out = runner(xg)
    (out * w).sum().backward()
    return xg.grad
g = step(); torch.cuda.synchronize()
t0 = time.perf_counter(); R = 2
for _ in range(R): g = step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / R
with torch.no_grad():                                           # (warm first: the no-grad chain's first call captures its graph / sizes its workspace)
    runner(x); torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(R): runner(x)
    torch.cuda.synchronize(); tf = (time.perf_counter() - t1) / R
print(f"white-box gradient step [{mode}]: B={B} t*={t_star}: {dt*1e3:.1f} ms (forward + backward) = {B/dt:.2f} clips/s; "
      f"forward-only purify {tf*1e3:.1f} ms; grad finite={bool(torch.isfinite(g).all())} |g|max={float(g.abs().max()):.3e}")

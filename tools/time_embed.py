import sys, os, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from audiopure_amd import synth, _native as N
from audiopure_amd.diffusion_models.DiffWave_Unconditional.WaveNet import WaveNet_Speech_Commands
dev = torch.device("cuda:0")
cfg = dict(synth.FULL_WAVENET_CONFIG)
net = WaveNet_Speech_Commands(**cfg).to(dev)
eng = net.engine(); lib = eng.lib
buf = torch.empty(36 * 256 + 512, device=dev)
big = torch.empty(1 << 28, device=dev)   # 1 GB to flush caches
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for _ in range(3): lib.ap_embed(eng.ctx, 3.0, N.ptr(buf), N.stream())
torch.cuda.synchronize(); e0.record()
for _ in range(50): lib.ap_embed(eng.ctx, 3.0, N.ptr(buf), N.stream())
e1.record(); torch.cuda.synchronize()
print("warm: %.1f us per ap_embed (embed_mlp + fct)" % (e0.elapsed_time(e1) / 50 * 1e3))
ts = []
for _ in range(10):
    big.fill_(1.0); torch.cuda.synchronize()
    e0.record(); lib.ap_embed(eng.ctx, 3.0, N.ptr(buf), N.stream()); e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) * 1e3)
print("cold (after a 1 GB fill): %.1f us" % (sum(ts) / len(ts)))

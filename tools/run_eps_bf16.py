"""One eps evaluation of the shipped net in bf16 mode, `reps` times (for rocprofv3 passes summed per kernel afterwards): the
deferred-skip form runs 36 resblock_bf16p_kernel<DS> launches + one skipgemm_bf16_kernel per evaluation; mode bf16s
(AP_PREC_BF16_STORE) runs init_conv_u + 36 resblock_bf16u_kernel launches + the same skip GEMM.
python tools/run_eps_bf16.py B reps [skip_group] [bf16|bf16s]"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audiopure_amd import synth
from audiopure_amd.diffusion_models.DiffWave_Unconditional.WaveNet import WaveNet_Speech_Commands
dev = torch.device("cuda:0")
B, reps = int(sys.argv[1]), int(sys.argv[2])
cfg = dict(synth.FULL_WAVENET_CONFIG)
net = WaveNet_Speech_Commands(**cfg)
net.load_state_dict({k: torch.from_numpy(v) for k, v in synth.wavenet_state_dict(cfg, 0).items()})
net = net.to(dev).set_precision(sys.argv[4] if len(sys.argv) > 4 else "bf16")
if len(sys.argv) > 3 and int(sys.argv[3]) >= 0:
    net.engine().skip_group = int(sys.argv[3])
x = torch.from_numpy(synth.waveforms(B, 16000, seed=6)).to(dev).reshape(B, 1, 16000)
with torch.no_grad():
    for _ in range(reps):
        net.eps(x, 3.0)
torch.cuda.synchronize()
print("done", net.engine().skip_group)

"""One BASELINE configs[4] step (mel-dB -> Improved-Diffusion UNet DDPM n = 5 -> ResNeXt-29) after a warm-up, for
`rocprofv3 --kernel-trace --stats -- python3 tools/run_cfg4_step.py [B] [steps]`: the per-kernel split of the whole step
(conv-as-GEMM family, GroupNorm, attention, mel, sampler updates)."""
import os
os.environ.setdefault("AUDIOPURE_STRICT", "1")
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(1, os.path.join(ROOT, "tools"))
import torch  # noqa: E402
from synth_convnets import CifarResNeXt, synth_init  # noqa: E402
from audiopure_amd.acoustic_system import AcousticSystem  # noqa: E402
from audiopure_amd.diffusion_models.improved_diffusion_ddpm import ImprovedDiffusionDDPM  # noqa: E402
from audiopure_amd.diffusion_models.improved_diffusion_unet import create_model, model_and_diffusion_defaults  # noqa: E402
from audiopure_amd.transforms import MelSpecDB  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
dev = torch.device("cuda:0")
unet = synth_init(create_model(**model_and_diffusion_defaults()), 0).to(dev)
clf = synth_init(CifarResNeXt(10), 0).to(dev)
system = AcousticSystem(classifier=clf, transform=MelSpecDB(32), defender=ImprovedDiffusionDDPM(unet, reverse_timestep=5),
                        defense_type="spec").eval()
g = torch.Generator(device=dev)
g.manual_seed(4321)
x = (torch.rand((B, 1, 16000), device=dev, generator=g) - 0.5).contiguous()
with torch.no_grad():
    system(x, True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        y = system(x, True)
    torch.cuda.synchronize()
print(f"B={B}: {(time.perf_counter() - t0) * 1e3 / steps:.2f} ms per step, {B * steps / (time.perf_counter() - t0):.1f} utt/s")

"""Which torch-level operators a configs[4] step still launches (copies, fills, elementwise): torch.profiler over one step.
   python tools/prof_cfg4_aten.py [B]"""
import os, sys, runpy
os.environ.setdefault("AUDIOPURE_STRICT", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(1, os.path.join(ROOT, "tools"))
import torch
from torch.profiler import profile, ProfilerActivity
from synth_convnets import CifarResNeXt, synth_init
from audiopure_amd.acoustic_system import AcousticSystem
from audiopure_amd.diffusion_models.improved_diffusion_ddpm import ImprovedDiffusionDDPM
from audiopure_amd.diffusion_models.improved_diffusion_unet import create_model, model_and_diffusion_defaults
from audiopure_amd.transforms import MelSpecDB
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device("cuda:0")
unet = synth_init(create_model(**model_and_diffusion_defaults()), 0).to(dev)
clf = synth_init(CifarResNeXt(10), 0).to(dev)
system = AcousticSystem(classifier=clf, transform=MelSpecDB(32), defender=ImprovedDiffusionDDPM(unet, reverse_timestep=5), defense_type="spec").eval()
x = (torch.rand((B, 1, 16000), device=dev) - 0.5).contiguous()
with torch.no_grad():
    system(x, True); torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        system(x, True); torch.cuda.synchronize()
rows = [e for e in prof.key_averages(group_by_stack_n=6) if e.key.startswith("aten::") and e.count >= 5]
rows.sort(key=lambda e: -e.count)
for e in rows[:25]:
    st = [s for s in e.stack if "audiopure_amd" in s][:2]
    print(f"{e.count:5d} x {e.key:28s} dev {e.device_time_total / 1e3:8.3f} ms   " + " <- ".join(s.split('/')[-1] for s in st))

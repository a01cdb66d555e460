"""BASELINE configs[4]: Improved-Diffusion UNet n=5 (spectrogram SDE purifier) + ResNeXt29 classifier, batch=256."""
import os as _os, sys as _sys; _sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__))); import _toolslib  # noqa: E401,E702  (-DAP_TOOLS library)
import sys, time, types, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from synth_convnets import CifarResNeXt, synth_init
from audiopure_amd.convnet import NativeConvNet
from audiopure_amd.acoustic_system import AcousticSystem
from audiopure_amd.transforms import MelSpecDB
from audiopure_amd.diffusion_models.improved_diffusion_unet import create_model, model_and_diffusion_defaults
from audiopure_amd.diffusion_models.improved_diffusion_sde import RevImprovedDiffusion
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
mode = "f32"
if len(sys.argv) > 2 and sys.argv[2] in ("f32", "f32s", "f32h"):   # arithmetic of the conv layers
    mode = sys.argv[2]
elif len(sys.argv) > 2:                                # A/B switch of the conv dispatch (ap_debug_conv_path)
    from audiopure_amd import _native as N
    N.lib().ap_debug_conv_path(int(sys.argv[2]))
unet = synth_init(create_model(**model_and_diffusion_defaults()), 0).to(dev)
args = types.SimpleNamespace(t=5, rand_t=False, t_delta=0, use_bm=False, sample_step=1, score_type="guided_diffusion")
unet.set_precision(mode)
defender = RevImprovedDiffusion.from_model(unet, args)
clf = NativeConvNet(synth_init(CifarResNeXt(10), 0)).eval().set_precision(mode)
system = AcousticSystem(classifier=clf, transform=MelSpecDB(32), defender=defender, defense_type="spec")
x = (torch.rand(B, 1, 16000, device=dev) - 0.5)
for _ in range(1): y = system(x, True)
torch.cuda.synchronize(); t0 = time.perf_counter()
R = 3
for _ in range(R): y = system(x, True)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / R
gflop = 5 * 16.76 + 10.77
print(f"config5[{mode}]: B={B} {dt*1e3:.1f} ms/step  {B/dt:.1f} samples/s  {gflop*B/dt/1e3:.1f} TFLOP/s (fp32 MFMA peak 157.3)  logits {tuple(y.shape)} finite={bool(torch.isfinite(y).all())}")

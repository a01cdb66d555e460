"""BASELINE configs[4], per conv shape: one step (UNet DDPM n = 5 + ResNeXt-29, batch 256) with every ap_conv2d_fwd launch
bracketed by HIP events on its stream (the product library's ap_conv_profile_* hook), summed by (shape, kernel class).
python tools/conv_by_shape.py [B] [mode]  ->  table sorted by time (profiles/r3_cfg4_conv_by_shape.txt)"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(1, os.path.dirname(os.path.abspath(__file__)))
import torch
from synth_convnets import CifarResNeXt, synth_init
from audiopure_amd import _native as N
from audiopure_amd.acoustic_system import AcousticSystem
from audiopure_amd.diffusion_models.improved_diffusion_ddpm import ImprovedDiffusionDDPM
from audiopure_amd.diffusion_models.improved_diffusion_unet import create_model, model_and_diffusion_defaults
from audiopure_amd.transforms import MelSpecDB
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
unet = synth_init(create_model(**model_and_diffusion_defaults()), 0).to(dev)
clf = synth_init(CifarResNeXt(10), 0).to(dev)
system = AcousticSystem(classifier=clf, transform=MelSpecDB(32), defender=ImprovedDiffusionDDPM(unet, reverse_timestep=5), defense_type="spec").eval()
if len(sys.argv) > 2:
    unet.set_precision(sys.argv[2]); system.classifier.set_precision(sys.argv[2])
x = torch.rand(B, 1, 16000, device=dev) - 0.5
lib = N.lib()
with torch.no_grad():
    system(x, True); system(x, True)
    torch.cuda.synchronize()
    N.check(lib.ap_conv_profile_enable(1))
    system(x, True)
    torch.cuda.synchronize()
NAMES = ["big2<128,128>", "big2<64,128>", "big2<128,64>", "split", "big", "generic", "w3 (F(2,3))", "p1 (stream)"]
agg = {}
i = 0
ms, fl, sh = C.c_double(), C.c_double(), (C.c_int * 10)()
while lib.ap_conv_profile_launch(i, C.byref(ms), C.byref(fl), sh) == 0:
    key = tuple(sh)
    a = agg.setdefault(key, [0, 0.0, 0.0]); a[0] += 1; a[1] += ms.value; a[2] += fl.value
    i += 1
N.check(lib.ap_conv_profile_enable(0))
tot_ms = sum(a[1] for a in agg.values()); tot_fl = sum(a[2] for a in agg.values())
print(f"configs[4] B={B}: {i} conv launches per step, {tot_ms:.1f} ms, {tot_fl / 1e12:.2f} TFLOP -> {tot_fl / tot_ms / 1e9:.1f} TFLOP/s = {tot_fl / tot_ms / 1e9 / 157.3:.3f} of the fp32 MFMA peak")
print(f"{'B':>4} {'Cin':>5} {'H':>3} {'W':>3} {'Cout':>5} {'k':>3} {'s':>2} {'g':>2} {'kernel':>14} {'n':>4} {'ms':>8} {'share':>6} {'TFLOP/s':>8} {'frac':>6}")
for key, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    b, cin, h, w, cout, kh, kw, st, g, cls = key
    print(f"{b:4d} {cin:5d} {h:3d} {w:3d} {cout:5d} {kh}x{kw} {st:2d} {g:2d} {NAMES[cls]:>14} {a[0]:4d} {a[1]:8.3f} {a[1] / tot_ms:6.3f} {a[2] / a[1] / 1e9:8.1f} {a[2] / a[1] / 1e9 / 157.3:6.3f}")

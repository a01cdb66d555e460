import sys, os, types, torch, collections, ctypes
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from audiopure_amd import _native as N
from synth_convnets import CifarResNeXt, synth_init
from audiopure_amd.convnet import NativeConvNet
from audiopure_amd.acoustic_system import AcousticSystem
from audiopure_amd.transforms import MelSpecDB
from audiopure_amd.diffusion_models.improved_diffusion_unet import create_model, model_and_diffusion_defaults
from audiopure_amd.diffusion_models.improved_diffusion_sde import RevImprovedDiffusion
dev = torch.device("cuda:0")
B = 256
unet = synth_init(create_model(**model_and_diffusion_defaults()), 0).to(dev)
args = types.SimpleNamespace(t=1, rand_t=False, t_delta=0, use_bm=False, sample_step=1, score_type="guided_diffusion")
mode = sys.argv[1] if len(sys.argv) > 1 else 'f32'
unet.set_precision(mode)
defender = RevImprovedDiffusion.from_model(unet, args)
clf = NativeConvNet(synth_init(CifarResNeXt(10), 0)).eval().set_precision(mode)
system = AcousticSystem(classifier=clf, transform=MelSpecDB(32), defender=defender, defense_type="spec")
x = (torch.rand(B, 1, 16000, device=dev) - 0.5)
y = system(x, True); torch.cuda.synchronize()
lib = N.lib(); orig = lib.ap_conv2d_fwd
hist = collections.Counter(); times = collections.defaultdict(float)
def wrapped(*a):
    (x_, wT, bias, res, out, B_, Cin, H, W, Cout, kh, kw, stride, pad, groups, relu, xcs, xco, st) = a
    Ho = (H + 2 * pad - kh) // stride + 1; Wo = (W + 2 * pad - kw) // stride + 1
    Nn = B_ * Ho * Wo; Mg = Cout // groups; Cg = Cin // groups
    frag = (Cg % 16 == 0 and Mg >= 64)
    t128 = -(-Nn // 128) * -(-Mg // 128) * groups
    path = ("big2_m64" if Mg < 128 else "big2_128" if t128 >= 512 else "big2_64") if frag else ("big" if (Mg >= 128 and kh <= 3 and t128 >= 512) else "small")
    key = (path, B_, Cin, H, W, Cout, kh, stride, groups)
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); r = orig(*a); e1.record(); torch.cuda.synchronize()
    hist[key] += 1; times[key] += e0.elapsed_time(e1)
    return r
lib.ap_conv2d_fwd = wrapped
import audiopure_amd.convnet as cv, audiopure_amd.diffusion_models.improved_diffusion_unet as un
y = system(x, True); torch.cuda.synchronize()
tot = sum(times.values())
print("total conv ms", tot)
for k, v in sorted(times.items(), key=lambda kv: -kv[1])[:22]:
    path, B_, Cin, H, W, Cout, kh, stride, groups = k
    Ho = (H + 2 * (kh // 2) - kh) // stride + 1
    fl = 2.0 * B_ * Ho * Ho * Cout * (Cin // groups) * kh * kh
    print(f"{v:8.2f} ms x{hist[k]:3d} {path:9s} Cin{Cin:4d} H{H:3d} Cout{Cout:4d} k{kh} s{stride} g{groups}  {fl * hist[k] / v / 1e9:7.1f} TF/s")

"""Block launch time of the AP_PREC_F32_SPLIT forms at B = 512 (HIP events, same process): F(2,3) two-kernel form against the direct form.
    python tools/time_f32s_forms.py [B] [layer] [reps]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audiopure_amd import synth, _native as N
from audiopure_amd.diffusion_models.DiffWave_Unconditional.WaveNet import WaveNet_Speech_Commands
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
layer = int(sys.argv[2]) if len(sys.argv) > 2 else 5
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 8
L = 16000
net = WaveNet_Speech_Commands(**dict(synth.FULL_WAVENET_CONFIG))
net.load_state_dict({k: torch.from_numpy(v) for k, v in synth.wavenet_state_dict(dict(synth.FULL_WAVENET_CONFIG), 0).items()})
net = net.to(dev)
h = torch.rand(B, 256, L, device=dev) * 3 - 1.5
ho = torch.empty_like(h); sk = torch.zeros_like(h)
pt = torch.rand(256, device=dev)
for mode in ("f32sw", "f32s", "f32sw", "f32s"):
    net.set_precision(mode)
    eng = net.engine(); lib = eng.lib
    for _ in range(2):
        N.check(lib.ap_resblock_fwd(eng.ctx, layer, N.ptr(h), N.ptr(pt), N.ptr(ho), N.ptr(sk), 1, B, L, N.stream()))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        N.check(lib.ap_resblock_fwd(eng.ctx, layer, N.ptr(h), N.ptr(pt), N.ptr(ho), N.ptr(sk), 1, B, L, N.stream()))
    e1.record(); torch.cuda.synchronize()
    print(f"{mode:6s} layer {layer} B {B}: {e0.elapsed_time(e1) / reps:.3f} ms per block")

"""bf16 chain: operand images handed from layer to layer (UB experiment, tools bit 0x400000) against the product's layer-wise
form (every layer stages fp32 h itself).  Same arithmetic -> eps must be bit-identical; then both timed.
python tools/ab_bf16_ub.py [B] [reps]        (AP_CMP_L=<clip length>)"""
import os as _os, sys as _sys; _sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__))); import _toolslib  # noqa: E401,E702
import sys, ctypes as C, torch
from audiopure_amd import synth, _native as N
from audiopure_amd.diffusion_models.DiffWave_Unconditional.WaveNet import WaveNet_Speech_Commands
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
lib = C.CDLL(N.LIB_PATH)
lib.ap_debug_bf16_dbg.argtypes = [C.c_int]
net = WaveNet_Speech_Commands(**dict(synth.FULL_WAVENET_CONFIG))
sd = synth.wavenet_state_dict(dict(synth.FULL_WAVENET_CONFIG), 0)
net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
net = net.to(dev).set_precision("bf16")
bad = 0
for L in [int(v) for v in _os.environ.get("AP_CMP_L", "16000,4001,1002,643,130").split(",")]:
    b = min(B, 4)
    x = torch.from_numpy(synth.waveforms(b, L, seed=5)).to(dev).reshape(b, 1, L)
    with torch.no_grad():
        lib.ap_debug_bf16_dbg(0x400000); a = net.eps(x, 3.0); torch.cuda.synchronize()
        lib.ap_debug_bf16_dbg(0); r = net.eps(x, 3.0); torch.cuda.synchronize()
        lib.ap_debug_bf16_dbg(0)
    same = torch.equal(a, r)
    bad += not same
    print(f"L = {L}: eps {'bit-identical' if same else 'DIFFERENT, max ' + str(float((a - r).abs().max()))} (|eps| max {float(r.abs().max()):.3f}, finite {bool(torch.isfinite(a).all())})")
L = 16000
x = torch.from_numpy(synth.waveforms(B, L, seed=6)).to(dev).reshape(B, 1, L)
for name, bits in (("layer-wise", 0), ("operand images", 0x400000), ("layer-wise", 0), ("operand images", 0x400000)):
    lib.ap_debug_bf16_dbg(bits)
    with torch.no_grad():
        net.eps(x, 3.0); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): net.eps(x, 3.0)
        e1.record(); torch.cuda.synchronize()
    print(f"B = {B}: {name:15s} {e0.elapsed_time(e1) / reps:8.3f} ms per eps evaluation (36 layers)")
lib.ap_debug_bf16_dbg(0)
sys.exit(1 if bad else 0)

"""One layer's bf16 backward (ap_resblock_bwd_bf16: two launches) timed at the white-box shape: python tools/time_bwd_bf16.py [B] [L]
(under rocprofv3 --kernel-trace --stats for the split between the two kernels)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 3:
    import _toolslib  # noqa: F401  (the -DAP_TOOLS library)
from audiopure_amd import synth, _native as N
from audiopure_amd.diffusion_models.DiffWave_Unconditional.WaveNet import WaveNet_Speech_Commands
B = int(sys.argv[1]) if len(sys.argv) > 1 else 10
L = int(sys.argv[2]) if len(sys.argv) > 2 else 16000
dev = torch.device("cuda:0")
cfg = synth.mini_wavenet_config(256, 12, 12)
net = WaveNet_Speech_Commands(**cfg)
net.load_state_dict({k: torch.from_numpy(v) for k, v in synth.wavenet_state_dict(cfg, 3).items()})
net = net.to(dev).set_precision("bf16")
eng = net.engine(); lib = eng.lib
N.check(lib.ap_ctx_prepare_backward(eng.ctx, N.stream()))
h = torch.randn(B, 256, L, device=dev); gh = torch.randn_like(h); gs = torch.randn_like(h); out = torch.empty_like(h)
pt = torch.randn(256, device=dev)
dy = torch.empty((B, L, 512), device=dev, dtype=torch.bfloat16)
def run(layer):
    N.check(lib.ap_resblock_bwd_bf16(eng.ctx, layer, N.ptr(h), N.ptr(pt), N.ptr(gh), N.ptr(gs), dy.data_ptr(), N.ptr(out), B, L, N.stream()))
def timed(layers=(0, 3, 6, 9, 11), reps=5):
    for l in layers: run(l)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        for l in layers: run(l)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (reps * len(layers))
print(f"B={B} L={L}: {timed()*1e3:.1f} us per layer (both launches)")
if os.path.exists(os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libaudiopure_hip_tools.so")) and len(sys.argv) > 3:
    # same-process A/B of the XCD-contiguous tile order (tools build: python tools/time_bwd_bf16.py B L ab, with _toolslib on the path)
    import ctypes
    lib.ap_debug_bwdb_linear.argtypes = [ctypes.c_int]
    for rep in range(3):
        for name, v in (("tile k on workgroup k", 1), ("one contiguous run of tiles per XCD", 0)):
            lib.ap_debug_bwdb_linear(v)
            print(f"  {name}: {timed()*1e3:.1f} us per layer")

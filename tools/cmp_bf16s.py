"""ap_resblock_bf16s.hip (small-batch deferred-skip block, 64-sample tiles) against the persistent kernel (tools build): bit identity of
h' and the g image over dilations / clip lengths / batches, and the launch time of both: python tools/cmp_bf16s.py"""
import sys, os, ctypes, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _toolslib  # noqa: F401
from audiopure_amd import synth, _native as N
from audiopure_amd.diffusion_models.DiffWave_Unconditional.WaveNet import WaveNet_Speech_Commands
dev = torch.device("cuda:0")
cfg = synth.mini_wavenet_config(256, 12, 12)
net = WaveNet_Speech_Commands(**cfg)
net.load_state_dict({k: torch.from_numpy(v) for k, v in synth.wavenet_state_dict(cfg, 3).items()})
net = net.to(dev).set_precision("bf16")
eng = net.engine(); lib = eng.lib
lib.ap_debug_no_bf16s.argtypes = [ctypes.c_int]
bad = 0
def run(layer, h, pt, B, L, with_h):
    ho = torch.full_like(h, float("nan")) if with_h else None
    g = torch.full((B, L, 256), 7.0, device=dev, dtype=torch.bfloat16)
    N.check(lib.ap_resblock_fwd_gate(eng.ctx, layer, N.ptr(h), N.ptr(pt), N.ptr(ho) if with_h else None, g.data_ptr(), B, L, N.stream()))
    return ho, g
for B, L in ((1, 16000), (2, 16000), (1, 130), (3, 1001), (2, 64), (1, 1), (2, 4100), (1, 7777), (5, 3), (1, 16001)):
    assert lib.ap_workspace_bytes(eng.ctx, B, L) > 0
    h = torch.randn(B, 256, L, device=dev) * 1.5
    pt = torch.randn(256, device=dev) * 0.5
    for layer in range(12):
        for with_h in (True, False):
            lib.ap_debug_no_bf16s(1); a = run(layer, h, pt, B, L, with_h)
            lib.ap_debug_no_bf16s(0); b = run(layer, h, pt, B, L, with_h)
            ok = torch.equal(a[1].view(torch.int16), b[1].view(torch.int16)) and (not with_h or torch.equal(a[0], b[0]))
            if not ok:
                bad += 1
                dh = float((a[0] - b[0]).abs().max()) if with_h else 0.0
                dg = float((a[1].float() - b[1].float()).abs().max())
                print(f"MISMATCH B={B} L={L} layer={layer} h'={with_h}: max |dh'| {dh:.3e} max |dg| {dg:.3e}")
print("bit identity:", "OK" if bad == 0 else f"{bad} mismatches")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for B in (1, 2, 3, 4, 6, 8, 10):
    L = 16000
    h = torch.randn(B, 256, L, device=dev); pt = torch.randn(256, device=dev)
    for flag, name in ((1, "persistent kernel"), (2, "64-sample tiles")):
        lib.ap_debug_no_bf16s(flag)
        for _ in range(3): run(5, h, pt, B, L, True)
        ho = torch.empty_like(h); g = torch.empty((B, L, 256), device=dev, dtype=torch.bfloat16)
        torch.cuda.synchronize(); e0.record()
        for _ in range(20):
            for layer in (0, 3, 6, 9, 11):
                lib.ap_resblock_fwd_gate(eng.ctx, layer, N.ptr(h), N.ptr(pt), N.ptr(ho), g.data_ptr(), B, L, N.stream())
        e1.record(); torch.cuda.synchronize()
        print(f"B={B} {name}: {e0.elapsed_time(e1) / 100 * 1e3:.1f} us per launch (back to back)")
sys.exit(1 if bad else 0)

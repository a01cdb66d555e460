"""The skip GEMM on 256-sample tiles (two column tiles per weight pass) against the 128-sample kernel (tools build):
bit identity over (batch, length, group size, accumulate) cases, then the launch time of both at a full group of 36 layers:
python tools/cmp_skipgemm.py [B]"""
import sys, os, ctypes, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _toolslib  # noqa: F401
from audiopure_amd import synth, _native as N
from audiopure_amd.diffusion_models.DiffWave_Unconditional.WaveNet import WaveNet_Speech_Commands
dev = torch.device("cuda:0")
cfg = dict(synth.FULL_WAVENET_CONFIG)
net = WaveNet_Speech_Commands(**cfg)
net.load_state_dict({k: torch.from_numpy(v) for k, v in synth.wavenet_state_dict(cfg, 0).items()})
net = net.to(dev).set_precision("bf16")
eng = net.engine(); lib = eng.lib
lib.ap_debug_skipgemm_wide.argtypes = [ctypes.c_int]
bad = 0
for B, L, nl, l0 in ((2, 16000, 36, 0), (1, 4133, 5, 7), (3, 130, 1, 35), (2, 1000, 2, 3), (1, 257, 12, 12), (5, 64, 3, 0), (1, 16001, 4, 20), (2, 2048, 36, 0)):
    g = (torch.randn(nl, B, L, 256, device=dev) * 0.5).to(torch.bfloat16)
    s0 = torch.randn(B, 256, L, device=dev)
    for acc in (0, 1):
        outs = []
        for wide in (0, 1):
            lib.ap_debug_skipgemm_wide(wide)
            G = 1024
            buf = torch.full((B * 256 * L + 2 * G,), 7.25, device=dev)
            sk = buf[G:G + B * 256 * L].view(B, 256, L)
            sk.copy_(s0)
            N.check(lib.ap_skip_gemm(eng.ctx, l0, nl, g.data_ptr(), N.ptr(sk), acc, B, L, N.stream()))
            if not (bool((buf[:G] == 7.25).all()) and bool((buf[G + B * 256 * L:] == 7.25).all())):
                print(f"OUT-OF-BOUNDS WRITE wide={wide} B={B} L={L} nl={nl}"); bad += 1
            outs.append(sk.clone())
        if not torch.equal(outs[0], outs[1]):
            bad += 1
            print(f"MISMATCH B={B} L={L} nl={nl} layer0={l0} accumulate={acc}: max |d| {float((outs[0] - outs[1]).abs().max()):.3e}")
print("bit identity:", "OK" if bad == 0 else f"{bad} mismatches")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
L, nl = 16000, 36
g = (torch.randn(nl, B, L, 256, device=dev) * 0.5).to(torch.bfloat16)
sk = torch.zeros(B, 256, L, device=dev)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for rep in range(3):
    for wide, name in ((0, "128-sample tiles"), (1, "256-sample tiles, half-layer chunks")):
        lib.ap_debug_skipgemm_wide(wide)
        for _ in range(2): lib.ap_skip_gemm(eng.ctx, 0, nl, g.data_ptr(), N.ptr(sk), 0, B, L, N.stream())
        torch.cuda.synchronize(); e0.record()
        for _ in range(5): lib.ap_skip_gemm(eng.ctx, 0, nl, g.data_ptr(), N.ptr(sk), 0, B, L, N.stream())
        e1.record(); torch.cuda.synchronize()
        print(f"B={B} nl={nl} {name}: {e0.elapsed_time(e1) / 5:.3f} ms per launch")
lib.ap_debug_skipgemm_wide(-1)
sys.exit(1 if bad else 0)

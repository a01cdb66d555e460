"""Timing-only variants of the bf16 residual block (tools build): python tools/dbg_resblock_bf16.py [B] [dbg bits ...]
bit 1 no weight loads in GEMM1, 2 no X loads, 4 no pack, 8 no MFMA in GEMM1, 16 no B-fragment LDS reads in GEMM1;
1048576 (0x100000, layers with d >= 64): two v_mfma_f32_16x16x32_bf16 in place of every 32x32x16 (the shape at equal FLOPs).
Outputs of the variants are wrong by construction; only the times mean anything."""
import os as _os, sys as _sys; _sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__))); import _toolslib  # noqa: E401,E702
import sys, ctypes as C, torch
from audiopure_amd import synth, _native as N
from audiopure_amd.diffusion_models.DiffWave_Unconditional.WaveNet import WaveNet_Speech_Commands
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
variants = [int(a) for a in sys.argv[2:]] or [0, 1, 2, 3, 4, 7, 8, 16, 23, 31]
L = 16000
h = torch.randn(B, 256, L, device=dev); ho = torch.empty_like(h); sk = torch.zeros_like(h)
pt = torch.randn(256, device=dev)
net = WaveNet_Speech_Commands(**dict(synth.FULL_WAVENET_CONFIG)).to(dev)
net.set_precision("bf16")
eng = net.engine(); lib = C.CDLL(N.LIB_PATH)
lib.ap_debug_bf16_dbg.argtypes = [C.c_int]
def t(layer, reps=5):
    for _ in range(2):
        N.check(eng.lib.ap_resblock_fwd(eng.ctx, layer, N.ptr(h), N.ptr(pt), N.ptr(ho), N.ptr(sk), 1, B, L, N.stream()))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        eng.lib.ap_resblock_fwd(eng.ctx, layer, N.ptr(h), N.ptr(pt), N.ptr(ho), N.ptr(sk), 1, B, L, N.stream())
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for rnd in range(2):
    for v in variants:
        lib.ap_debug_bf16_dbg(v)
        print(f"round {rnd} dbg {v:2d}: layer 0 (d=1) {t(0):7.3f} ms   layer 1 (d=2) {t(1):7.3f} ms   layer 5 (d=32) {t(5):7.3f} ms   layer 9 (d=512) {t(9):7.3f} ms", flush=True)
lib.ap_debug_bf16_dbg(0)

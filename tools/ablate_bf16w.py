"""Timing-only variants of the one-wave-per-SIMD bf16 block (tools build; outputs wrong by construction), interleaved rounds in
one process.  python tools/ablate_bf16w.py [B] [layer] [rounds]"""
import os as _os, sys as _sys; _sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__))); import _toolslib  # noqa: E401,E702
import sys, ctypes as C, torch
from audiopure_amd import synth, _native as N
from audiopure_amd.diffusion_models.DiffWave_Unconditional.WaveNet import WaveNet_Speech_Commands
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
layer = int(sys.argv[2]) if len(sys.argv) > 2 else 9
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 3
L = 16000
torch.manual_seed(0)
h = torch.randn(B, 256, L, device=dev); sk = torch.randn(B, 256, L, device=dev); pt = torch.randn(256, device=dev); ho = torch.empty_like(h)
net = WaveNet_Speech_Commands(**dict(synth.FULL_WAVENET_CONFIG)).to(dev)
net.set_precision("bf16")
eng = net.engine(); lib = C.CDLL(N.LIB_PATH)
lib.ap_debug_bf16_dbg.argtypes = [C.c_int]
def timed(dbg, n=4):
    lib.ap_debug_bf16_dbg(dbg)
    N.check(eng.lib.ap_resblock_fwd(eng.ctx, layer, N.ptr(h), N.ptr(pt), N.ptr(ho), N.ptr(sk), 1, B, L, N.stream()))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        N.check(eng.lib.ap_resblock_fwd(eng.ctx, layer, N.ptr(h), N.ptr(pt), N.ptr(ho), N.ptr(sk), 1, B, L, N.stream()))
    e1.record(); torch.cuda.synchronize(); lib.ap_debug_bf16_dbg(0)
    return e0.elapsed_time(e1) / n
W = 0x20000
V = [(W, "as built"), (0, "eight-wave product kernel"), (W + 1, "no W1 loads"), (W + 2, "no X loads"), (W + 4, "no pack"), (W + 6, "no X loads, no pack"),
     (W + 7, "no W1/X loads, no pack"), (W + 8, "no GEMM1 MFMA"), (W + 15, "GEMM1 emptied"), (W + 32, "no gate math"), (W + 64, "no GEMM2 MFMA"),
     (W + 128, "no RMW loads"), (W + 256, "no stores"), (W + 384, "no RMW loads/stores"), (W + 384 + 15, "GEMM1 emptied, no RMW"),
     (W + 384 + 96, "no gate, no GEMM2 MFMA, no RMW")]
for _ in range(6): timed(W)
for r in range(rounds):
    print(f"round {r} layer {layer} (d={1 << (layer % 12)}) B={B}: " + "  ".join(f"[{n}] {timed(d):.3f}" for d, n in V), flush=True)

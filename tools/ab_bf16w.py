"""One-wave-per-SIMD bf16 block kernel (ap_resblock_bf16w.hip) vs the eight-wave persistent kernel (ap_resblock_bf16p.hip) in
one process (tools build): bit-identity of both outputs, then interleaved timing rounds.
python tools/ab_bf16w.py [B] [rounds] [layers ...]"""
import os as _os, sys as _sys; _sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__))); import _toolslib  # noqa: E401,E702
import sys, ctypes as C, torch
from audiopure_amd import synth, _native as N
from audiopure_amd.diffusion_models.DiffWave_Unconditional.WaveNet import WaveNet_Speech_Commands
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
layers = [int(a) for a in sys.argv[3:]] or [2, 5, 9, 11]
L = 16000
torch.manual_seed(0)
h = torch.randn(B, 256, L, device=dev); sk0 = torch.randn(B, 256, L, device=dev); pt = torch.randn(256, device=dev)
net = WaveNet_Speech_Commands(**dict(synth.FULL_WAVENET_CONFIG)).to(dev)
net.set_precision("bf16")
eng = net.engine(); lib = C.CDLL(N.LIB_PATH)
lib.ap_debug_bf16_dbg.argtypes = [C.c_int]
ho = torch.empty_like(h)
def run(layer, dbg, acc, sk):
    lib.ap_debug_bf16_dbg(dbg)
    N.check(eng.lib.ap_resblock_fwd(eng.ctx, layer, N.ptr(h), N.ptr(pt), N.ptr(ho), N.ptr(sk), acc, B, L, N.stream()))
    lib.ap_debug_bf16_dbg(0)
bad = 0
for layer in layers:
    for acc in (0, 1):
        sa, sb = sk0.clone(), sk0.clone()
        run(layer, 0x20000, acc, sa); a = ho.clone(); run(layer, 0, acc, sb); b = ho.clone()
        torch.cuda.synchronize()
        same = torch.equal(a, b) and torch.equal(sa, sb)
        bad += not same
        msg = "bit-identical" if same else f"DIFFERENT max {float((a - b).abs().max())} / {float((sa - sb).abs().max())} nan {int(torch.isnan(a).sum())}"
        print(f"layer {layer:2d} (d={1 << (layer % 12)}) accumulate={acc}: {msg}", flush=True)
        if not same:
            df = (a - b).abs()
            cols = (df.amax(dim=(0, 1)) > 0).nonzero().flatten(); rows = (df.amax(dim=(0, 2)) > 0).nonzero().flatten()
            print("   h' differs at", len(cols), "of", L, "columns, first", cols[:8].tolist(), "; rows", len(rows), rows[:8].tolist())
            df = (sa - sb).abs()
            cols = (df.amax(dim=(0, 1)) > 0).nonzero().flatten(); rows = (df.amax(dim=(0, 2)) > 0).nonzero().flatten()
            print("   skip differs at", len(cols), "of", L, "columns, first", cols[:8].tolist(), "; rows", len(rows), rows[:8].tolist())
sk = sk0.clone()
def timed(layer, dbg, n=4):
    run(layer, dbg, 1, sk)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): run(layer, dbg, 1, sk)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for _ in range(6):                                      # warm the chip into its steady clock
    timed(layers[0], 0)
for r in range(rounds):
    print(f"round {r}: " + "   ".join(f"layer {l} (d={1 << (l % 12)}) w {timed(l, 0x20000):.3f} p {timed(l, 0):.3f} ms" for l in layers), flush=True)
sys.exit(1 if bad else 0)

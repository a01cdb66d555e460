"""bf16 block with GEMM1 on v_mfma_f32_16x16x32_bf16 (tools bit 0x200000) vs the product kernel (32x32x16): closeness of both outputs
(same operands, a different grouping of the fp32 accumulation), then interleaved timing.  python tools/ab_bf16_m16.py [B] [rounds] [layers ...]"""
import os as _os, sys as _sys; _sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__))); import _toolslib  # noqa: E401,E702
import sys, ctypes as C, torch
from audiopure_amd import synth, _native as N
from audiopure_amd.diffusion_models.DiffWave_Unconditional.WaveNet import WaveNet_Speech_Commands
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
layers = [int(a) for a in sys.argv[3:]] or [9, 11]
L = 16000
torch.manual_seed(0)
h = torch.randn(B, 256, L, device=dev); sk0 = torch.randn(B, 256, L, device=dev); pt = torch.randn(256, device=dev)
net = WaveNet_Speech_Commands(**dict(synth.FULL_WAVENET_CONFIG)).to(dev)
net.set_precision("bf16")
eng = net.engine(); lib = C.CDLL(N.LIB_PATH)
lib.ap_debug_bf16_dbg.argtypes = [C.c_int]
ho = torch.empty_like(h)
M = 0x200000
def run(layer, dbg, acc, sk):
    lib.ap_debug_bf16_dbg(dbg)
    N.check(eng.lib.ap_resblock_fwd(eng.ctx, layer, N.ptr(h), N.ptr(pt), N.ptr(ho), N.ptr(sk), acc, B, L, N.stream()))
    lib.ap_debug_bf16_dbg(0)
for layer in layers:
    sa, sb = sk0.clone(), sk0.clone()
    run(layer, M, 1, sa); a = ho.clone(); run(layer, 0, 1, sb); b = ho.clone()
    torch.cuda.synchronize()
    eh = float((a - b).abs().max() / b.abs().max()); es = float((sa - sb).abs().max() / sb.abs().max())
    print(f"layer {layer:2d} (d={1 << (layer % 12)}): max |dh'| / max|h'| = {eh:.2e}   max |dskip| / max|skip| = {es:.2e}   nan {int(torch.isnan(a).sum())}", flush=True)
sk = sk0.clone()
def timed(layer, dbg, n=4):
    run(layer, dbg, 1, sk)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): run(layer, dbg, 1, sk)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for _ in range(6): timed(layers[0], 0)
for r in range(rounds):
    print(f"round {r}: " + "   ".join(f"layer {l} (d={1 << (l % 12)}) 16x16x32 {timed(l, M):.3f}  32x32x16 {timed(l, 0):.3f} ms" for l in layers), flush=True)

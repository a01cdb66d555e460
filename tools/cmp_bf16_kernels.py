"""Persistent vs per-tile bf16 residual block on the same inputs (tools build): the two kernels do the same arithmetic in the
same order, so outputs must be bit-identical.  python tools/cmp_bf16_kernels.py [B] [layers ...]
(AP_CMP_DBG=<bits> compares a tools variant of the persistent kernel that is meant to be exact.)"""
import os as _os, sys as _sys; _sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__))); import _toolslib  # noqa: E401,E702
import sys, ctypes as C, torch
from audiopure_amd import synth, _native as N
from audiopure_amd.diffusion_models.DiffWave_Unconditional.WaveNet import WaveNet_Speech_Commands
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
layers = [int(a) for a in sys.argv[2:]] or [0, 1, 2, 9, 11, 12, 13]
L = int(_os.environ.get("AP_CMP_L", "16000"))
torch.manual_seed(0)
h = torch.randn(B, 256, L, device=dev); sk0 = torch.randn(B, 256, L, device=dev); pt = torch.randn(256, device=dev)
net = WaveNet_Speech_Commands(**dict(synth.FULL_WAVENET_CONFIG)).to(dev)
net.set_precision("bf16")
eng = net.engine(); lib = C.CDLL(N.LIB_PATH)
lib.ap_debug_bf16_dbg.argtypes = [C.c_int]
def run(layer, dbg, acc):
    lib.ap_debug_bf16_dbg(dbg)
    ho = torch.empty_like(h); sk = sk0.clone()
    N.check(eng.lib.ap_resblock_fwd(eng.ctx, layer, N.ptr(h), N.ptr(pt), N.ptr(ho), N.ptr(sk), acc, B, L, N.stream()))
    torch.cuda.synchronize(); lib.ap_debug_bf16_dbg(0)
    return ho, sk
bad = 0
for layer in layers:
    for acc in (0, 1):
        a, b = run(layer, int(_os.environ.get('AP_CMP_DBG', '0')), acc), run(layer, 4096, acc)
        same = torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
        bad += not same
        if not same:
            df = (a[0] - b[0]).abs()
            cols = (df.amax(dim=(0, 1)) > 0).nonzero().flatten()
            rows = (df.amax(dim=(0, 2)) > 0).nonzero().flatten()
            print("   h' differs at", len(cols), "of", L, "time positions, first", cols[:12].tolist(), "last", cols[-4:].tolist(), "; rows", len(rows))
        print(f"layer {layer:2d} (d={1 << (layer % 12)}) accumulate={acc}: {'bit-identical' if same else 'DIFFERENT max ' + str(float((a[0] - b[0]).abs().max())) + ' / ' + str(float((a[1] - b[1]).abs().max()))}")
sys.exit(1 if bad else 0)

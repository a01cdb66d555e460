"""Where a wave of the one-wave-per-SIMD bf16 residual block spends a tile (tools build, DBG 2048): s_memtime stamps of the 7th
tile of every workgroup.  python tools/trace_resblock_bf16w.py [B] [layer]"""
import os as _os, sys as _sys; _sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__))); import _toolslib  # noqa: E401,E702
import sys, ctypes as C, torch, numpy as np
from audiopure_amd import synth, _native as N
from audiopure_amd.diffusion_models.DiffWave_Unconditional.WaveNet import WaveNet_Speech_Commands
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
layer = int(sys.argv[2]) if len(sys.argv) > 2 else 9
L = 16000
h = torch.randn(B, 256, L, device=dev); ho = torch.empty_like(h); sk = torch.zeros_like(h)
pt = torch.randn(256, device=dev)
net = WaveNet_Speech_Commands(**dict(synth.FULL_WAVENET_CONFIG)).to(dev)
net.set_precision("bf16")
eng = net.engine(); lib = C.CDLL(N.LIB_PATH)
lib.ap_debug_bf16_dbg.argtypes = [C.c_int]; lib.ap_debug_wtrace.argtypes = [C.c_void_p]
G, NW = 256, 4
tr = torch.zeros(G * NW * 64, dtype=torch.int64, device=dev)
assert lib.ap_debug_wtrace(C.c_void_p(tr.data_ptr())) == 0
lib.ap_debug_bf16_dbg(0x20000 + 2048)
for _ in range(3):
    N.check(eng.lib.ap_resblock_fwd(eng.ctx, layer, N.ptr(h), N.ptr(pt), N.ptr(ho), N.ptr(sk), 1, B, L, N.stream()))
torch.cuda.synchronize()
lib.ap_debug_bf16_dbg(0)
t = tr.cpu().numpy().reshape(G, NW, 64).astype(np.int64)
names = {1: "accumulator init", 2: "barrier", 18: "gate (+ h patch requests)", 19: "barrier", 20: "GEMM2 pass 0 (+ 2nd half of h patch requests)",
         21: "epilogue 0 (+ skip row requests)", 22: "next X chunk 0 request + GEMM2 pass 1", 23: "epilogue 1, first half",
         24: "next tile head: fragments, pack chunk 0, X chunk 1 request", 30: "epilogue 1, second half"}
for ch in range(8):
    names[3 + 2 * ch] = f"chunk {ch} MFMA"
    if ch < 7: names[4 + 2 * ch] = f"chunk {ch} barrier"
order = sorted(names)
tot = np.median(t[:, :, 30] - t[:, :, 0]); rt = np.median(t[:, :, 41] - t[:, :, 40])
print(f"layer {layer} B={B}: median tile = {tot:.0f} shader cycles = {rt:.0f} ticks of 100 MHz -> {tot / rt * 0.1:.3f} GHz")
prev = 0; sums = {}
for i in order:
    dseg = t[:, :, i] - t[:, :, prev]; prev = i
    key = names[i].split(" ", 2)[2] if names[i].startswith("chunk") else None
    if key: sums[key] = sums.get(key, 0) + np.median(dseg)
    print(f"  {i:2d} {names[i]:58s} median {np.median(dseg):8.0f}   p10 {np.percentile(dseg, 10):8.0f}  p90 {np.percentile(dseg, 90):8.0f}")
print("sums over the chunks:", {k: int(v) for k, v in sums.items()})

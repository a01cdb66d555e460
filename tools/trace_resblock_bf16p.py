"""Where a wave of the persistent bf16 residual block spends a tile (tools build, DBG 2048): s_memtime stamps of the 7th tile
of every workgroup.  python tools/trace_resblock_bf16p.py [B] [layer] [extra dbg bits]"""
import os as _os, sys as _sys; _sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__))); import _toolslib  # noqa: E401,E702
import sys, ctypes as C, torch, numpy as np
from audiopure_amd import synth, _native as N
from audiopure_amd.diffusion_models.DiffWave_Unconditional.WaveNet import WaveNet_Speech_Commands
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
layer = int(sys.argv[2]) if len(sys.argv) > 2 else 9
extra = int(sys.argv[3], 0) if len(sys.argv) > 3 else 0
L = 16000
h = torch.randn(B, 256, L, device=dev); ho = torch.empty_like(h); sk = torch.zeros_like(h)
pt = torch.randn(256, device=dev)
net = WaveNet_Speech_Commands(**dict(synth.FULL_WAVENET_CONFIG)).to(dev)
net.set_precision("bf16")
eng = net.engine(); lib = C.CDLL(N.LIB_PATH)
lib.ap_debug_bf16_dbg.argtypes = [C.c_int]; lib.ap_debug_ptrace.argtypes = [C.c_void_p]
G = 256
tr = torch.zeros(G * 8 * 64, dtype=torch.int64, device=dev)
assert lib.ap_debug_ptrace(C.c_void_p(tr.data_ptr())) == 0
lib.ap_debug_bf16_dbg(2048 + extra)
for _ in range(3):
    N.check(eng.lib.ap_resblock_fwd(eng.ctx, layer, N.ptr(h), N.ptr(pt), N.ptr(ho), N.ptr(sk), 1, B, L, N.stream()))
torch.cuda.synchronize()
t = tr.cpu().numpy().reshape(G, 8, 64).astype(np.int64)
names = {1: "pack chunk 0", 2: "barrier", 26: "gate (+ operand requests)", 27: "barrier", 28: "GEMM2 pass 0 k-loop", 29: "epilogue 0",
         30: "GEMM2 pass 1 k-loop", 31: "next tile: X chunk 0 + weight requests", 32: "epilogue 1"}
for ch in range(7):
    names[3 + ch * 3] = f"chunk {ch} heavy half"; names[4 + ch * 3] = f"chunk {ch} light half (+pack)"; names[5 + ch * 3] = f"chunk {ch} barrier"
names[24] = "chunk 7 heavy half"; names[25] = "chunk 7 light half"
tot = np.median(t[:, :, 32] - t[:, :, 0])
rt = np.median(t[:, :, 41] - t[:, :, 40])
print(f"layer {layer} B={B} dbg {2048 + extra}: median tile = {tot:.0f} shader cycles = {rt:.0f} ticks of 100 MHz -> {tot / rt * 0.1:.3f} GHz")
acc = {}
for i in range(1, 33):
    dseg = t[:, :, i] - t[:, :, i - 1]
    key = names[i] if not names[i].startswith("chunk") else names[i].split(" ", 2)[2]
    acc.setdefault(key, []).append(np.median(dseg))
    print(f"  {i:2d} {names[i]:34s} median {np.median(dseg):8.0f}   p10 {np.percentile(dseg, 10):8.0f}  p90 {np.percentile(dseg, 90):8.0f}")
print("sums over the 8 chunks:", {k: int(sum(v)) for k, v in acc.items() if len(v) > 1})
lib.ap_debug_bf16_dbg(0)
# who is late: per workgroup the wave that arrives last at a chunk barrier waits ~0; the others wait for it
bar = np.stack([t[:, :, 5 + ch * 3] - t[:, :, 4 + ch * 3] for ch in range(7)], 0)           # [chunk][wg][wave]
arr = np.stack([t[:, :, 4 + ch * 3] - t[:, :, 2 + ch * 3] for ch in range(7)], 0)           # heavy + light, per wave
print("chunk barriers: median over workgroups of the SHORTEST wait in the workgroup:", int(np.median(bar.min(2))))
print("last wave to arrive (share of barriers):", np.round(np.bincount(bar.argmin(2).ravel(), minlength=8) / bar[:, :, 0].size, 2))
print("median heavy+light per wave:", np.median(arr, (0, 1)).astype(int))
print("median barrier wait per wave:", np.median(bar, (0, 1)).astype(int))
rel = t[:, :, 4:25:3] - t[:, :1, 4:25:3]
print("arrival at the chunk barrier relative to wave 0 (median per wave):", np.median(rel, (0, 2)).astype(int))

if extra & 0x8000000:                                     # per-k-step stamps of chunk 3 (slots 42..47; the chunk's end is stamp 13)
    ks = np.stack([t[:, :, 43 + i] - t[:, :, 42 + i] for i in range(5)] + [t[:, :, 13] - t[:, :, 47]], 0)      # [k-step][wg][wave]
    print("chunk 3, cycles per k-step (8 MFMAs = 256 cycles of matrix pipe per wave; median over workgroups), per wave:")
    for i in range(6):
        print(f"  k-step {i}:", np.median(ks[i], 0).astype(int), " all waves:", int(np.median(ks[i])))

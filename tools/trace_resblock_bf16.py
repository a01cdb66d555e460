"""Phase timeline of the bf16 residual-block kernel: per-WG cycle stamps of waves 0 and 7 (ap_debug_trace)."""
import os as _os, sys as _sys; _sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__))); import _toolslib  # noqa: E401,E702  (-DAP_TOOLS library)
import sys, os, ctypes as C, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audiopure_amd import synth, _native as N
from audiopure_amd.diffusion_models.DiffWave_Unconditional.WaveNet import WaveNet_Speech_Commands
dev = torch.device("cuda:0")
net = WaveNet_Speech_Commands(**dict(synth.FULL_WAVENET_CONFIG)).to(dev)
net.set_precision("bf16")
eng = net.engine(); lib = eng.lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
layer = int(sys.argv[2]) if len(sys.argv) > 2 else 5
mask = int(sys.argv[3]) if len(sys.argv) > 3 else 0
lib.ap_debug_ablate.argtypes = [C.c_int]
lib.ap_debug_ablate(mask)
L = 16000
h = torch.randn(B, 256, L, device=dev); ho = torch.empty_like(h); sk = torch.zeros_like(h)
pt = torch.randn(256, device=dev)
nblk = B * 125
tr = torch.zeros(nblk * 8 * 16, dtype=torch.int64, device=dev)
lib.ap_debug_trace.argtypes = [C.c_void_p]
for _ in range(2):
    lib.ap_resblock_fwd(eng.ctx, layer, N.ptr(h), N.ptr(pt), N.ptr(ho), N.ptr(sk), 1, B, L, N.stream())
lib.ap_debug_trace(C.c_void_p(tr.data_ptr()))
lib.ap_resblock_fwd(eng.ctx, layer, N.ptr(h), N.ptr(pt), N.ptr(ho), N.ptr(sk), 1, B, L, N.stream())
torch.cuda.synchronize()
lib.ap_debug_trace(None)
t = tr.cpu().numpy().reshape(nblk, 8, 16).astype(np.int64)
names = ["start", "prologue", "chunk0", "chunk1-3", "chunk4-7", "gate", "g2p0 mma", "g2p0 store", "g2p1 mma", "g2p1 store"]
sel = slice(nblk // 4, 3 * nblk // 4) if nblk > 1024 else slice(0, nblk)
for w in range(8):
    d = np.diff(t[sel, w, :10], axis=1)
    tot = t[sel, w, 9] - t[sel, w, 0]
    print(f"wave {w}: total median {np.median(tot):.0f} cyc  mean {tot.mean():.0f}")
    for i in range(9):
        if w not in (0, 7): break
        print(f"   {names[i + 1]:12s} median {np.median(d[:, i]):8.0f}  mean {d[:, i].mean():8.0f}  p90 {np.percentile(d[:, i], 90):8.0f}")
    c4 = t[sel, w][:, [3, 10, 11, 12, 13]]
    dd = np.diff(c4, axis=1)
    print("   chunk 4: mma(a0) %.0f  mma(a1) %.0f  store_chunk %.0f  barrier %.0f" % tuple(np.median(dd, axis=0)))
    c5 = t[sel, w][:, [13, 14, 15, 10]]
    dd = np.diff(c5, axis=1)
    print("   chunk 5 head: mark %.0f  wait-a0 %.0f  first half after wait %.0f" % tuple(np.median(dd, axis=0)))
span = t[:, :, 9].max() - t[:, :, 0].min()
print("whole launch span (cycles):", span, " tiles/CU:", nblk / 256, " span per tile-slot:", span / (nblk / 256))

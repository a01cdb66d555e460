"""Phase timeline of the split-precision residual-block kernels (s_memtime stamps of all 8 waves, ap_debug_trace):
python tools/trace_resblock_f32s.py [B] [layer]"""
import os as _os, sys as _sys; _sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__))); import _toolslib  # noqa: E401,E702  (-DAP_TOOLS library)
import sys, os, ctypes as C, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audiopure_amd import synth, _native as N
from audiopure_amd.diffusion_models.DiffWave_Unconditional.WaveNet import WaveNet_Speech_Commands
dev = torch.device("cuda:0")
net = WaveNet_Speech_Commands(**dict(synth.FULL_WAVENET_CONFIG)).to(dev)
net.set_precision(sys.argv[3] if len(sys.argv) > 3 else "f32s")
eng = net.engine(); lib = eng.lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
layer = int(sys.argv[2]) if len(sys.argv) > 2 else 5
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
sel = slice(nblk // 4, 3 * nblk // 4) if nblk > 1024 else slice(0, nblk)
H = False                                                    # (the fp16-split block of rounds 3-4 is gone)
seq = [0, 1, 2, 3, 4, 5, 6, 14] if H else [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 14]
names = ["prologue", "iter0 (2 chunks)", "iters1-3", "iters4-7", "gate + barrier", "g2 pass0", "g2 pass1"] if H else ["prologue", "iter0 (2 chunks)", "iters1-3", "iters4-7", "gate h0", "g2 h0 pass0", "g2 h0 pass1", "gate h1", "g2 h1 pass0", "g2 h1 pass1"]
for w in (0, 4, 7):
    tt = t[sel, w][:, seq]
    d = np.diff(tt, axis=1)
    print(f"wave {w}: total median {np.median(tt[:, -1] - tt[:, 0]):.0f} cycles")
    for i, n in enumerate(names):
        print(f"   {n:18s} median {np.median(d[:, i]):8.0f}")
    c = t[sel, w][:, [3, 10, 11, 12, 13]]
    dd = np.median(np.diff(c, axis=1), axis=0)
    print("   iteration 4, first chunk: kstep0 %.0f  ksteps1-2 %.0f  pack+store %.0f  barrier %.0f" % tuple(dd))

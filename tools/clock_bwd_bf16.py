"""The in-kernel clock of the bf16 backward's conv kernel (tools build): python tools/clock_bwd_bf16.py [B] [seconds]
s_memtime / s_memrealtime stamps around its chunk loop, read after [seconds] of back-to-back launches on random data
(MI355X_MICROARCH.md, DVFS item 6): clock = d(s_memtime) / d(s_memrealtime) x 100 MHz, median over workgroups; and the loop's
cycles against its MFMA issue time (384 MFMAs x 32 cycles per wave, three workgroups per CU sharing the SIMDs)."""
import sys, os, time, ctypes, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _toolslib  # noqa: F401
from audiopure_amd import synth, _native as N
from audiopure_amd.diffusion_models.DiffWave_Unconditional.WaveNet import WaveNet_Speech_Commands
B = int(sys.argv[1]) if len(sys.argv) > 1 else 10
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 3.0
L = 16000
dev = torch.device("cuda:0")
cfg = synth.mini_wavenet_config(256, 12, 12)
net = WaveNet_Speech_Commands(**cfg)
net.load_state_dict({k: torch.from_numpy(v) for k, v in synth.wavenet_state_dict(cfg, 3).items()})
net = net.to(dev).set_precision("bf16")
eng = net.engine(); lib = eng.lib
N.check(lib.ap_ctx_prepare_backward(eng.ctx, N.stream()))
lib.ap_debug_bwdb_stamp.argtypes = [ctypes.c_void_p]
h = torch.randn(B, 256, L, device=dev); gh = torch.randn_like(h); gs = torch.randn_like(h); out = torch.empty_like(h)
pt = torch.randn(256, device=dev)
dy = torch.empty((B, L, 512), device=dev, dtype=torch.bfloat16)
nwg = B * ((L + 63) // 64)
stamps = torch.zeros((nwg, 2), device=dev, dtype=torch.int64)
run = lambda layer: N.check(lib.ap_resblock_bwd_bf16(eng.ctx, layer, N.ptr(h), N.ptr(pt), N.ptr(gh), N.ptr(gs), dy.data_ptr(), N.ptr(out), B, L, N.stream()))
run(5); torch.cuda.synchronize()
t0 = time.time()
while time.time() - t0 < secs:
    for _ in range(20): run(5)
    torch.cuda.synchronize()
assert lib.ap_debug_bwdb_stamp(stamps.data_ptr()) == 0
run(5); torch.cuda.synchronize()
lib.ap_debug_bwdb_stamp(None)
s = stamps.cpu().double()
clk = (s[:, 0] / s[:, 1] * 0.1).median().item()                  # GHz
cyc = s[:, 0].median().item()
print(f"B={B}: K2 chunk loop, median over {nwg} workgroups: {cyc:.0f} shader cycles, in-kernel clock {clk:.3f} GHz "
      f"({s[:, 1].median().item() / 100:.1f} us); MFMA issue time of the loop 12 288 cycles per wave, x 3 workgroups per CU = 36 864: "
      f"the loop runs at {36864 / cyc:.2f} of that bound")

"""What would fewer GEMM1 products buy the AP_PREC_BF16_STORE block under the board's power cap?  Timing-only A/B in one process:
the product kernel against instantiations that run 6 and 5 of GEMM1's 8 chunks (75 % / 62.5 % of the dilated conv's matrix work;
a minimal-filtering F(2,3) form would run 66.7 % of it) with NOTHING added for the saving -- no fourth tap, no input transform, no
second accumulator set, no wider weight stream.  An upper bound on the gain of VERDICT r5 item 2b.
    python tools/ab_bf16u_flops.py [B] [layer] [reps]"""
import ctypes as C
import sys

import _toolslib  # noqa: F401
import torch

from audiopure_amd import _native as N, synth
from audiopure_amd.diffusion_models.DiffWave_Unconditional.WaveNet import WaveNet_Speech_Commands

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
layer = int(sys.argv[2]) if len(sys.argv) > 2 else 5
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
L = 16000
dev = torch.device("cuda:0")
cfg = dict(synth.FULL_WAVENET_CONFIG)
net = WaveNet_Speech_Commands(**cfg)
net.load_state_dict({k: torch.from_numpy(v) for k, v in synth.wavenet_state_dict(cfg, 0).items()})
net = net.to(dev).set_precision("bf16s")
eng = net.engine()
lib = eng.lib
lib.ap_debug_bf16_dbg.argtypes = [C.c_int]
uin = (torch.rand((B, 8, L, 32), device=dev) * 3 - 1.5).to(torch.bfloat16)
uout = torch.empty_like(uin)
g = torch.empty((B, L, 256), dtype=torch.bfloat16, device=dev)
ptn = torch.rand(256, device=dev)


def run(bits, n):
    lib.ap_debug_bf16_dbg(bits)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3):
        N.check(lib.ap_resblock_fwd_u(eng.ctx, layer, uin.data_ptr(), N.ptr(ptn), uout.data_ptr(), g.data_ptr(), B, L, N.stream()))
    e0.record()
    for _ in range(n):
        N.check(lib.ap_resblock_fwd_u(eng.ctx, layer, uin.data_ptr(), N.ptr(ptn), uout.data_ptr(), g.data_ptr(), B, L, N.stream()))
    e1.record()
    torch.cuda.synchronize()
    lib.ap_debug_bf16_dbg(0)
    return e0.elapsed_time(e1) / n


print(f"AP_PREC_BF16_STORE block, B = {B}, layer {layer} (d = {2 ** (layer % 12)}), {reps} launches per variant, two rounds")
for rnd in range(2):
    t8, t6, t5 = run(0, reps), run(0x10000000, reps), run(0x20000000, reps)
    # GEMM1 is 6/7 of the block's matrix work (res_conv 1/7): chunks 8 -> 6 removes 21.4 % of the MFMA flops, 8 -> 5 32.1 %, F(2,3) 28.6 %
    t_f23 = t6 + (t5 - t6) * (28.6 - 21.4) / (32.1 - 21.4)
    print(f"  round {rnd}: 8 chunks {t8:.3f} ms | 6 chunks {t6:.3f} ms ({100 * (1 - t6 / t8):.1f} % less) | 5 chunks {t5:.3f} ms "
          f"({100 * (1 - t5 / t8):.1f} % less) | interpolated at F(2,3)'s flop count {t_f23:.3f} ms ({100 * (1 - t_f23 / t8):.1f} % less)")

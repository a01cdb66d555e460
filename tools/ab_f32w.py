"""Same-process A/B of the F(2,3) fp32 block's two epilogue forms (tools build): 16-byte stores through LDS patches (clip lengths
that are multiples of four) against the 4-byte form every other length takes.
   python tools/ab_f32w.py [B]"""
import sys

import _toolslib  # noqa: F401
import ctypes as C

import torch

from audiopure_amd import synth, _native as N
from audiopure_amd.diffusion_models.DiffWave_Unconditional.WaveNet import WaveNet_Speech_Commands


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    dev = torch.device("cuda:0")
    cfg = synth.mini_wavenet_config(256, 12, 12)
    net = WaveNet_Speech_Commands(**cfg)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in synth.wavenet_state_dict(cfg, 3).items()})
    net = net.to(dev)
    eng = net.engine()
    lib = eng.lib
    L = 16000
    hd = torch.rand(B, 256, L, device=dev) * 3 - 1.5
    hout = torch.empty_like(hd)
    sk = torch.zeros_like(hd)
    pt = torch.rand(256, device=dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def t(layer, n=6):
        for _ in range(2):
            N.check(lib.ap_resblock_fwd(eng.ctx, layer, N.ptr(hd), N.ptr(pt), N.ptr(hout), N.ptr(sk), 1, B, L, N.stream()))
        e0.record()
        for _ in range(n):
            N.check(lib.ap_resblock_fwd(eng.ctx, layer, N.ptr(hd), N.ptr(pt), N.ptr(hout), N.ptr(sk), 1, B, L, N.stream()))
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n

    lib.ap_debug_f32w_q16.argtypes = [C.c_int]
    for rep in range(2):
        for layer in (0, 5, 11):
            t16 = t(layer)
            lib.ap_debug_f32w_q16(0)
            t4 = t(layer)
            lib.ap_debug_f32w_q16(1)
            print(f"layer {layer:2d}  16-byte epilogue: {t16:7.3f} ms   4-byte epilogue: {t4:7.3f} ms", flush=True)


if __name__ == "__main__":
    main()

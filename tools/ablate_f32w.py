"""Where the F(2,3) fp32 block's time goes: timing-only ablations of the tools build (results wrong by construction).
   python tools/ablate_f32w.py [B]      -> ms per launch (layer 5 and layer 11) for each ablation mask"""
import sys

import _toolslib  # noqa: F401  (points audiopure_amd at the -DAP_TOOLS library)
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
    lib.ap_debug_ablate_f32w.argtypes = [C.c_int]
    L = 16000
    hd = torch.rand(B, 256, L, device=dev) * 3 - 1.5
    hout = torch.empty_like(hd)
    sk = torch.zeros_like(hd)
    pt = torch.rand(256, device=dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def t(layer, mask, n=4):
        lib.ap_debug_ablate_f32w(mask)
        for _ in range(2):
            N.check(lib.ap_resblock_fwd(eng.ctx, layer, N.ptr(hd), N.ptr(pt), N.ptr(hout), N.ptr(sk), 1, B, L, N.stream()))
        e0.record()
        for _ in range(n):
            N.check(lib.ap_resblock_fwd(eng.ctx, layer, N.ptr(hd), N.ptr(pt), N.ptr(hout), N.ptr(sk), 1, B, L, N.stream()))
        e1.record()
        torch.cuda.synchronize()
        lib.ap_debug_ablate_f32w(0)
        return e0.elapsed_time(e1) / n

    names = [(0, "product kernel"), (1, "no gate math"), (2, "no epilogue stores"), (4, "no residual loads"), (2 | 4, "no epilogue stores, no residual loads"),
             (8, "no X loads in the chunk loop"), (16, "no GEMM1 weight loads"), (32, "no staging transform / LDS writes"),
             (8 | 32, "no X loads, no staging"), (64, "no per-chunk barrier"), (512, "no GEMM2 weight loads"), (16 | 512, "no weight loads at all"),
             (128, "no GEMM1 MFMAs"), (256, "no GEMM2 MFMAs"), (128 | 256, "no MFMAs"),
             (1 | 2 | 4 | 8 | 16 | 32 | 64 | 512, "MFMAs + LDS fragment reads only"), (0, "product kernel (again)")]
    ideal = 3072 * 64 * 250 * B / 256 / 2.4e9 * 1e3 / 1   # MFMA cycles per CU at 2.4 GHz, ms per launch
    print(f"B = {B}: MFMA time at 2.4 GHz = {ideal:.2f} ms per launch")
    for layer in (5, 11):
        base = None
        for mask, name in names:
            ms = t(layer, mask)
            base = base or ms
            print(f"layer {layer:2d}  mask {mask:5d}  {ms:8.3f} ms  ({(ms - base) / base * 100:+6.1f} %)  {name}", flush=True)


if __name__ == "__main__":
    main()

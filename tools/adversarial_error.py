"""Per-dot-product error of the residual-block kernels against an fp64 evaluation on adversarial operands:
cancellation (channel pairs of opposite sign and nearly equal magnitude), a 2^-20 ... 2^20 dynamic range across channels, and plain
unit-scale data.   python tools/adversarial_error.py  -> error of h' and skip_n relative to max |fp64 result|, per mode and case."""
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from audiopure_amd import synth, _native as N          # noqa: E402
from oracle import diffwave_oracle as O                # noqa: E402


def cases(B, C, L):
    g = torch.Generator().manual_seed(7)
    base = torch.randn(B, C, L, generator=g)
    out = {"unit": base}
    x = base.clone()                                                # cancellation: channel 2k+1 = -(channel 2k) (1 + 2^-12 noise), scale 64
    x[:, 1::2] = -x[:, 0::2] * (1 + 2.0 ** -12 * torch.randn(B, C // 2, L, generator=g))
    out["cancel"] = 64.0 * x
    e = torch.linspace(-20, 20, C).view(1, C, 1)                     # dynamic range: channel c scaled by 2^e(c)
    out["range"] = base * torch.pow(2.0, e.round())
    e2 = torch.linspace(-10, 10, C).view(1, C, 1)
    out["range10"] = base * torch.pow(2.0, e2.round())
    return out


def main():
    from audiopure_amd.diffusion_models.DiffWave_Unconditional.WaveNet import WaveNet_Speech_Commands
    dev = torch.device("cuda:0")
    cfg = synth.mini_wavenet_config(256, 12, 12)
    sd = synth.wavenet_state_dict(cfg, 3)
    net = WaveNet_Speech_Commands(**cfg)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    net = net.to(dev)
    w = O.fold_state_dict(sd)
    w64 = {k: v.double() for k, v in w.items()}
    B, C, L = 2, 256, 2048
    modes = sys.argv[1:] or ["f32d", "f32", "f32s"]
    res = {}
    for layer in (2, 7):
        d = 2 ** layer
        p = f"residual_layer.residual_blocks.{layer}"
        pt = torch.zeros(C)
        for name, h in cases(B, C, L).items():
            with torch.no_grad():
                emb0 = torch.zeros(B, 512)
                # fp64 evaluation with part_t = fc_t bias only (emb = 0)
                h64, s64 = O.residual_block(w64, layer, d, h.double(), emb0.double())
                part = w[p + ".fc_t.bias"].clone()
            for mode in modes:
                net.set_precision(mode)
                eng = net.engine()
                hd, ptd = h.to(dev), part.to(dev).contiguous()
                ho, sk = torch.empty_like(hd), torch.zeros_like(hd)
                N.check(eng.lib.ap_resblock_fwd(eng.ctx, layer, N.ptr(hd), N.ptr(ptd), N.ptr(ho), N.ptr(sk), 0, B, L, N.stream()))
                torch.cuda.synchronize()
                eh = float((ho.cpu().double() - h64).abs().max() / h64.abs().max())
                es = float((sk.cpu().double() - s64).abs().max() / s64.abs().max())
                res[(layer, name, mode)] = (eh, es)
    print(f"{'layer':>5} {'case':>8} " + " ".join(f"{m + ' h/skip':>22}" for m in modes) + "   ratio to f32d (h', skip)")
    for layer in (2, 7):
        for name in ("unit", "cancel", "range10", "range"):
            row = [res[(layer, name, m)] for m in modes]
            ref = res[(layer, name, modes[0])]
            print(f"{layer:5d} {name:>8} " + " ".join(f"{a:10.3e} {b:10.3e} " for a, b in row) + "  " +
                  " ".join(f"{m}:{a / ref[0]:.2f}/{b / ref[1]:.2f}" for m, (a, b) in zip(modes, row)))


if __name__ == "__main__":
    main()

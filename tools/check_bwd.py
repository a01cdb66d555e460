"""Debug / check of ap_resblock_bwd's two kernels against torch pieces.  python tools/check_bwd.py"""
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from audiopure_amd import synth, _native as N          # noqa: E402
from oracle import diffwave_oracle as O                # noqa: E402


def rel(a, b):
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def main():
    from audiopure_amd.diffusion_models.DiffWave_Unconditional.WaveNet import WaveNet_Speech_Commands
    dev = torch.device("cuda:0")
    cfg = synth.mini_wavenet_config(256, 12, 12)
    sd = synth.wavenet_state_dict(cfg, 3)
    net = WaveNet_Speech_Commands(**cfg)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    net = net.to(dev)
    w = O.fold_state_dict(sd)
    eng = net.engine()
    lib = eng.lib
    N.check(lib.ap_ctx_prepare_backward(eng.ctx, N.stream()))
    for L, layer in [(1100, 0), (2048, 5), (1000, 11), (130, 3), (16000, 6)]:
        B, C, d = 2, 256, 2 ** layer
        p = f"residual_layer.residual_blocks.{layer}"
        g = torch.Generator().manual_seed(L)
        gh = torch.randn(B, C, L, generator=g)
        gs = torch.randn(B, C, L, generator=g)
        y = torch.randn(B, 2 * C, L, generator=g)
        W1, Wr, Ws = w[p + ".dilated_conv_layer.conv.weight"], w[p + ".res_conv.weight"][:, :, 0], w[p + ".skip_conv.weight"][:, :, 0]
        dg = torch.einsum("oc,bot->bct", Wr, gh * (0.5 ** 0.5)) + torch.einsum("sc,bst->bct", Ws, gs)
        th, sg = torch.tanh(y[:, :C]), torch.sigmoid(y[:, C:])
        dy_ref = torch.cat([dg * sg * (1 - th * th), dg * th * sg * (1 - sg)], 1)
        dh_ref = gh * (0.5 ** 0.5) + F.conv_transpose1d(dy_ref, W1, dilation=d, padding=d)
        dy = torch.full((B, 2 * C, L), 3.0, device=dev)
        dh = torch.full((B, C, L), 5.0, device=dev)
        ghd, gsd, yd = gh.to(dev), gs.to(dev), y.to(dev)          # (held: a temporary's block would be reused by the next .to())
        N.check(lib.ap_resblock_bwd(eng.ctx, layer, N.ptr(ghd), N.ptr(gsd), N.ptr(yd), N.ptr(dy), N.ptr(dh), B, L, N.stream()))
        torch.cuda.synchronize()
        print(f"L={L} layer={layer} d={d}: dy err {rel(dy.cpu(), dy_ref):.3e}  dh err {rel(dh.cpu(), dh_ref):.3e}", flush=True)
        e = (dy.cpu() - dy_ref).abs()
        if e.max() > 1e-3:
            bad = (e > 1e-3).nonzero()
            print("  dy bad count", len(bad), "first", bad[:5].tolist(), "rows bad", sorted(set(bad[:, 1].tolist()))[:20], "cols", sorted(set(bad[:, 2].tolist()))[:20])
        e = (dh.cpu() - dh_ref).abs()
        if e.max() > 1e-3:
            bad = (e > 1e-3).nonzero()
            print("  dh bad count", len(bad), "of", e.numel(), "rows bad", sorted(set(bad[:, 1].tolist()))[:20], "cols", sorted(set(bad[:, 2].tolist()))[:40])


if __name__ == "__main__":
    main()

"""GPU check of the minimal-filtering fp32 block (ap_resblock_f32w.hip) against the oracle and the direct-form kernel, and a
timing A/B of the two forms.  Run on the GPU box:  python tools/check_f32w.py [--time B]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from audiopure_amd import synth, _native as N          # noqa: E402
from conftest import rel_err                           # noqa: E402
from oracle import diffwave_oracle as O                # noqa: E402


def net_for(dev, seed=3):
    from audiopure_amd.diffusion_models.DiffWave_Unconditional.WaveNet import WaveNet_Speech_Commands
    cfg = synth.mini_wavenet_config(256, 12, 12)
    sd = synth.wavenet_state_dict(cfg, seed)
    net = WaveNet_Speech_Commands(**cfg)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    return net.to(dev), O.fold_state_dict(sd)


def main():
    dev = torch.device("cuda:0")
    net, w = net_for(dev)
    eng = net.engine()
    lib = eng.lib
    assert lib.ap_ctx_get_f32_form(eng.ctx) == 1
    worst = 0.0
    cases = [(1500, 2), (2048, 10), (16000, 0), (16000, 1), (16000, 4), (16000, 5), (16000, 6), (16000, 7), (16000, 11), (4133, 8),
             (4133, 11), (1000, 11), (130, 3), (130, 9), (77, 0), (5, 1), (1, 0), (2, 0), (3, 1), (63, 5), (64, 5), (65, 5), (16001, 9), (333, 6)]
    for L, layer in cases:
        B = 2
        h = torch.from_numpy(synth.uniform(f"h/256/{L}", (B, 256, L), 1, -1.5, 1.5))
        skip0 = torch.from_numpy(synth.uniform(f"s/256/{L}", (B, 256, L), 1, -1.0, 1.0))
        emb = torch.from_numpy(synth.uniform("emb", (B, 512), 1, -1.0, 1.0))
        emb[1] = emb[0]
        with torch.no_grad():
            p = f"residual_layer.residual_blocks.{layer}"
            part_t = torch.nn.functional.linear(emb[:1], w[p + ".fc_t.weight"], w[p + ".fc_t.bias"]).reshape(-1)
            h_ref, s_ref = O.residual_block(w, layer, 2 ** (layer % 12), h.clone(), emb)
        hd, pt = h.to(dev), part_t.to(dev).contiguous()
        res = {}
        for form in (1, 0):
            N.check(lib.ap_ctx_set_f32_form(eng.ctx, form))
            sk = skip0.to(dev).clone()
            hout = torch.full_like(hd, 3.0)
            N.check(lib.ap_resblock_fwd(eng.ctx, layer, N.ptr(hd), N.ptr(pt), N.ptr(hout), N.ptr(sk), 1, B, L, N.stream()))
            sk2 = torch.full_like(sk, 7.0)
            N.check(lib.ap_resblock_fwd(eng.ctx, layer, N.ptr(hd), N.ptr(pt), N.ptr(hout), N.ptr(sk2), 0, B, L, N.stream()))
            torch.cuda.synchronize()
            res[form] = (rel_err(hout.cpu().numpy(), h_ref.numpy()), rel_err(sk.cpu().numpy(), (skip0 + s_ref).numpy()),
                         rel_err(sk2.cpu().numpy(), s_ref.numpy()))
        N.check(lib.ap_ctx_set_f32_form(eng.ctx, 1))
        worst = max(worst, *res[1])
        print(f"L={L:6d} layer={layer:2d} d={2 ** layer:5d}  winograd h'/skip+/skip= {res[1][0]:.2e} {res[1][1]:.2e} {res[1][2]:.2e}   direct {res[0][0]:.2e} {res[0][1]:.2e} {res[0][2]:.2e}",
              flush=True)
    print("worst winograd error", worst, "OK" if worst < 5e-6 else "FAIL")
    if "--time" in sys.argv:
        B = int(sys.argv[sys.argv.index("--time") + 1])
        L = 16000
        hd = torch.rand(B, 256, L, device=dev) * 3 - 1.5
        hout = torch.empty_like(hd)
        sk = torch.zeros_like(hd)
        pt = torch.rand(256, device=dev)
        for form in (1, 0, 1, 0):
            N.check(lib.ap_ctx_set_f32_form(eng.ctx, form))
            for layer in (0, 3, 5, 6, 9, 11):
                for _ in range(2):
                    N.check(lib.ap_resblock_fwd(eng.ctx, layer, N.ptr(hd), N.ptr(pt), N.ptr(hout), N.ptr(sk), 1, B, L, N.stream()))
                torch.cuda.synchronize()
                t = time.time()
                n = 5
                for _ in range(n):
                    N.check(lib.ap_resblock_fwd(eng.ctx, layer, N.ptr(hd), N.ptr(pt), N.ptr(hout), N.ptr(sk), 1, B, L, N.stream()))
                torch.cuda.synchronize()
                ms = (time.time() - t) / n * 1e3
                fl = 16.777e9 * B
                print(f"form={form} layer={layer:2d} B={B}: {ms:8.3f} ms  {fl / ms / 1e9:7.1f} TFLOP/s algorithmic (direct-form flops) = {fl / ms / 1e9 / 157.3:.3f} of peak",
                      flush=True)
    return 0 if worst < 5e-6 else 1


if __name__ == "__main__":
    sys.exit(main())

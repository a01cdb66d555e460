"""Where the bf16 backward's deviation from autograd through the bf16-emulating oracle comes from: python tools/check_bwd_bf16.py
Per block (random operands) and through a 6-layer eps net: cosine, relative L2 and max deviation (of the largest entry) of
  - ap_resblock_bwd_bf16 (cotangents rounded to bf16 for the matrix pipe),
  - the composed fp32 backward of bf16 mode (fp32 GEMMs on the bf16 forward's layer inputs)
against the oracle, and of the oracle's own fp32 gradient against its bf16-emulating one (the distance between the two arithmetics)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audiopure_amd import synth, _native as N
from audiopure_amd.diffusion_models.DiffWave_Unconditional.WaveNet import WaveNet_Speech_Commands
from audiopure_amd.diffusion_models._grad import EpsGrad
from oracle import diffwave_oracle as O

dev = torch.device("cuda:0")
def stats(a, b):
    a, b = a.double().reshape(-1), b.double().reshape(-1)
    return f"cos {float(a @ b / (a.norm() * b.norm())):.6f}  L2 {float((a - b).norm() / b.norm()):.3e}  max {float((a - b).abs().max() / b.abs().max()):.3e}"

def net_(cfg, seed):
    sd = synth.wavenet_state_dict(cfg, seed)
    net = WaveNet_Speech_Commands(**cfg)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    return net.to(dev).set_precision("bf16"), O.fold_state_dict(sd)

cfg = synth.mini_wavenet_config(256, 12, 12)
net, w = net_(cfg, 3)
eng = net.engine(); lib = eng.lib
N.check(lib.ap_ctx_prepare_backward(eng.ctx, N.stream()))
B, C_ = 2, 256
for L, layer in ((1100, 0), (2048, 5), (16000, 6)):
    d = 2 ** layer
    h = torch.from_numpy(synth.uniform(f"gh/{L}", (B, C_, L), 1, -1.5, 1.5))
    gh = torch.from_numpy(synth.uniform(f"gg/{L}", (B, C_, L), 2, -1.0, 1.0))
    gs = torch.from_numpy(synth.uniform(f"gs/{L}", (B, C_, L), 3, -1.0, 1.0))
    emb = torch.from_numpy(synth.uniform("emb", (1, 512), 1, -1.0, 1.0)).repeat(B, 1)
    p = f"residual_layer.residual_blocks.{layer}"
    refs = {}
    for bf in (True, False):
        hr = h.clone().requires_grad_(True)
        a, s = O.residual_block(w, layer, d, hr, emb, bf16_operands=bf)
        (refs[bf],) = torch.autograd.grad([a, s], hr, [gh, gs])
    with torch.no_grad():
        part_t = torch.nn.functional.linear(emb[:1], w[p + ".fc_t.weight"], w[p + ".fc_t.bias"]).reshape(-1)
    hd, ptd, ghd, gsd = h.to(dev), part_t.to(dev).contiguous(), gh.to(dev), gs.to(dev)
    dy = torch.zeros((B, L, 2 * C_), device=dev, dtype=torch.bfloat16)
    out = torch.empty_like(hd)
    N.check(lib.ap_resblock_bwd_bf16(eng.ctx, layer, N.ptr(hd), N.ptr(ptd), N.ptr(ghd), N.ptr(gsd), dy.data_ptr(), N.ptr(out), B, L, N.stream()))
    print(f"block L={L} d={d}: kernel vs bf16 oracle: {stats(out.cpu(), refs[True])}")
    print(f"                   fp32 oracle vs bf16 oracle: {stats(refs[False], refs[True])}")
    # minus the identity path sqrt(1/2) dh' (it dominates the norm and is exact in every arithmetic)
    idp = gh * (0.5 ** 0.5)
    print(f"                   without the residual path: kernel {stats(out.cpu() - idp, refs[True] - idp)}; fp32 oracle {stats(refs[False] - idp, refs[True] - idp)}")

cfg = synth.mini_wavenet_config(256, 6, 12)
net, w = net_(cfg, 6)
B, L, step = 2, 1500, 3.0
x = torch.from_numpy(synth.waveforms(B, L, seed=11))
v = torch.from_numpy(synth.uniform(f"v{L}", (B, 1, L), 1, -1.0, 1.0))
refs = {}
for bf in (True, False):
    xr = x.clone().requires_grad_(True)
    e = O.eps_net(w, cfg, xr, torch.full((B, 1), step), bf16_operands=bf)
    (refs[bf],) = torch.autograd.grad(e, xr, v)
eg = EpsGrad(net)
for fused in (True, False):
    eg.fused_bf16 = fused
    eps, saved = eg.forward_save(x.to(dev), step)
    g = eg.backward(saved, v.to(dev)).cpu()
    print(f"eps net (6 layers) {'fused bf16' if fused else 'composed fp32'} backward vs bf16 oracle: {stats(g, refs[True])}")
print(f"eps net fp32 oracle vs bf16 oracle: {stats(refs[False], refs[True])}")

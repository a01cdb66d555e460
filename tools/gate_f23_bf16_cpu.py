"""CPU-only numerics gate for an F(2,3) form of the dilated conv on the bf16 matrix pipe (no kernel computes this):
the transformed weights W0, (W0+W1+W2)/2, (W0-W1+W2)/2, W2 rounded to bf16 once, the input differences d0-d2, d1+d2, d2-d1, d3-d1 formed in
fp32 and rounded to bf16, four bf16 x bf16 -> fp32 products per dilation pair, the output transform in fp32.  Against the direct bf16
arithmetic the AP_PREC_BF16 / AP_PREC_BF16_STORE kernels compute, on the quantities the GPU tests hold those modes to:
one block against fp64, eps of the shipped net against the reference's fp32 vector (no bar of its own: the GPU tests hold eps to the
emulating oracle), the 5-step DDPM chain against the reference's fp32 vector (bar 2e-3 for `bf16`, tests/test_gpu_dropin.py, and for
`bf16s`, tests/test_gpu_bf16_store.py).
    python tools/gate_f23_bf16_cpu.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(1, os.path.join(ROOT, "tests"))
from audiopure_amd import synth  # noqa: E402
from oracle import diffwave_oracle as O  # noqa: E402

q = O._bf16


def f23_bf16(u, W, bias, d):
    """oracle.winograd_dilated_conv with both operands of every product rounded to bf16 (products and sums fp32)."""
    B, C, L = u.shape
    W64 = W.double()
    G = [W64[:, :, 0], (W64[:, :, 0] + W64[:, :, 1] + W64[:, :, 2]) / 2, (W64[:, :, 0] - W64[:, :, 1] + W64[:, :, 2]) / 2, W64[:, :, 2]]
    G = [q(g.float()).to(u.dtype) for g in G]
    t = torch.arange(L)
    tf = t[(t // d) % 2 == 0]

    def tap(off):
        idx = tf + off
        ok = (idx >= 0) & (idx < L)
        return torch.where(ok, u[:, :, idx.clamp(0, L - 1)], torch.zeros((), dtype=u.dtype))

    d0, d1, d2, d3 = tap(-d), tap(0), tap(d), tap(2 * d)
    r = (lambda x: q(x.float()).to(u.dtype))
    m1 = torch.einsum("oc,bcp->bop", G[0], r(d0 - d2))
    m2 = torch.einsum("oc,bcp->bop", G[1], r(d1 + d2)) + bias.view(1, -1, 1)
    m3 = torch.einsum("oc,bcp->bop", G[2], r(d2 - d1))
    m4 = torch.einsum("oc,bcp->bop", G[3], r(d3 - d1))
    y = torch.empty(B, W.shape[0], L, dtype=u.dtype)
    y[:, :, tf] = (m1 + m2) + m3
    ts = tf + d
    ok = ts < L
    y[:, :, ts[ok]] = ((m2 - m3) + m4)[:, :, ok]
    return y


def rel(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max())


FORMS = (("bf16 direct", dict(bf16_operands=True)), ("bf16 F(2,3)", dict(bf16_operands=True, winograd=True)),
         ("bf16s direct", dict(bf16_store=True)), ("bf16s F(2,3)", dict(bf16_store=True, winograd=True)))
fp32_f23 = O.winograd_dilated_conv


def run(fn, kw):
    O.winograd_dilated_conv = f23_bf16 if kw.get("winograd") else fp32_f23   # (residual_block hands the F(2,3) routine the unrounded u and W)
    try:
        with torch.no_grad():
            return fn(**kw)
    finally:
        O.winograd_dilated_conv = fp32_f23


print("== one block against fp64 (mini net C = 256, seed 3: the GPU block tests' weights), h ~ U(-1.5, 1.5), L = 2048; h' / skip, of max ==")
mcfg = synth.mini_wavenet_config(256, 12, 12)
wm = O.fold_state_dict(synth.wavenet_state_dict(mcfg, 3))
w64 = {k: v.double() for k, v in wm.items()}
B, C_, L = 2, 256, 2048
emb = torch.from_numpy(synth.uniform("emb", (1, 512), 1, -1.0, 1.0)).repeat(B, 1)
ratios = []
for layer in (0, 2, 5, 7, 9):
    h = torch.from_numpy(synth.uniform(f"h/256/{L}", (B, C_, L), 1, -1.5, 1.5))
    with torch.no_grad():
        h64, s64 = O.residual_block(w64, layer, 2 ** layer, h.double(), emb.double())
    row = {}
    for name, kw in FORMS:
        ho, so = run(lambda **k: O.residual_block(wm, layer, 2 ** layer, h.clone(), emb, **k), kw)
        row[name] = (rel(ho, h64), rel(so, s64))
    ratios += [row["bf16 F(2,3)"][k] / row["bf16 direct"][k] for k in (0, 1)]
    print(f"  layer {layer} (d = {2 ** layer:3d}): " + "   ".join(f"{n} {a:.2e}/{b:.2e}" for n, (a, b) in row.items()))
print(f"  F(2,3) / direct, bf16: x{min(ratios):.2f} .. x{max(ratios):.2f}")

print("== the shipped net against the reference's fp32 vectors ==")
g = np.load(os.path.join(ROOT, "tests", "golden", "golden_v1.npz"))
cfg = dict(synth.FULL_WAVENET_CONFIG)
w = O.fold_state_dict(synth.wavenet_state_dict(cfg, 0))
dh = O.diffusion_hyperparams(**synth.DIFFUSION_CONFIG)
x0 = torch.from_numpy(synth.waveforms(2, 16000, seed=1234))
zs = [torch.from_numpy(synth.noise(k, 2, 16000, seed=1234)) for k in range(6)]
eps_ref, x5_ref = torch.from_numpy(g["full/eps_t4"]), torch.from_numpy(g["full/ddpm_n5/x"])
res = {}
for name, kw in FORMS:
    eps = run(lambda **k: O.eps_net(w, cfg, x0, 4.0 * torch.ones(2, 1), **k), kw)
    x5 = run(lambda **k: O.ddpm_purify(w, cfg, dh, x0, 5, zs, **k), kw)
    res[name] = (rel(eps, eps_ref), rel(x5, x5_ref))
    print(f"  {name:13s} eps(t = 4) {res[name][0]:.2e}   DDPM-5 chain {res[name][1]:.2e} (bar 2e-3)", flush=True)
for m in ("bf16", "bf16s"):
    a, b = res[f"{m} direct"], res[f"{m} F(2,3)"]
    print(f"  {m}: F(2,3) / direct  eps x{b[0] / a[0]:.2f}  chain x{b[1] / a[1]:.2f}   chain within its bar: {b[1] <= 2e-3}")
